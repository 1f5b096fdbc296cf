#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint8_t *src, uint32_t nbytes, uint4 *out) {
    __shared__ __attribute__((aligned(16))) uint8_t dump[64 * 16 * 4];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, (int)nbytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(dump + wave * 1024), 16,
                                         (int)(threadIdx.x * 16), 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[threadIdx.x] = *(uint4 *)(dump + wave * 1024 + lane * 16);
}
int main() {
    std::vector<uint8_t> h(8192);
    for (int i = 0; i < 8192; i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; uint4 *o;
    hipMalloc(&d, 8192); hipMalloc(&o, 256 * 16);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, 8192u, o);
    std::vector<uint8_t> r(4096);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4096; i++) bad += r[i] != h[i];
    printf("lds-direct b128: %d mismatches\n", bad);
    return bad != 0;
}
