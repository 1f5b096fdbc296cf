#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t* src, uint32_t bytes, uint32_t* out, uint32_t* out4, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 3u + 1u, soff, 0);   // unaligned dword
    u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 15u + 1u, soff, 0);     // unaligned 16 bytes
    out[threadIdx.x] = v;
    out4[threadIdx.x * 4 + 0] = q.x; out4[threadIdx.x * 4 + 1] = q.y; out4[threadIdx.x * 4 + 2] = q.z; out4[threadIdx.x * 4 + 3] = q.w;
}
int main() {
    const int N = 4096;
    std::vector<uint8_t> h(N);
    for (int i = 0; i < N; i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t* d; uint32_t *o, *o4;
    hipMalloc(&d, N); hipMalloc(&o, 64 * 4); hipMalloc(&o4, 64 * 16);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    int bad = 0;
    for (int soff : {0, 5, 1022}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (uint32_t)N, o, o4, soff);
        std::vector<uint32_t> ho(64), ho4(256);
        hipMemcpy(ho.data(), o, 256, hipMemcpyDeviceToHost);
        hipMemcpy(ho4.data(), o4, 1024, hipMemcpyDeviceToHost);
        for (int t = 0; t < 64; t++) {
            uint32_t w; memcpy(&w, &h[soff + t * 3 + 1], 4);
            if (w != ho[t]) bad++;
            uint32_t q[4]; memcpy(q, &h[soff + t * 15 + 1], 16);
            for (int e = 0; e < 4; e++) if (q[e] != ho4[t * 4 + e]) bad++;
        }
    }
    printf("unaligned buffer loads: %d mismatches\n", bad);
    return bad != 0;
}
