// How fast does the chip start workgroups?  Empty kernels (one global load so that they are not
// optimised away entirely) with k_me16b's launch shape: threads per workgroup, static LDS, VGPRs.
// hipcc --offload-arch=gfx950 -O3 -o ub_dispatch tools/ub_dispatch.hip && ./ub_dispatch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int LDS, int VG>
__global__ __launch_bounds__(256) void k(const uint32_t* in, uint32_t* out, int n) {
    __shared__ uint8_t s[LDS > 0 ? LDS : 4];
    uint32_t v[VG];
    if (n < 0) {  // never: keeps LDS and the registers allocated
#pragma unroll
        for (int i = 0; i < VG; i++) v[i] = in[threadIdx.x + i * 64];
        uint32_t a = 0;
#pragma unroll
        for (int i = 0; i < VG; i++) a += v[i] * (i + 1);
        s[threadIdx.x % (LDS > 0 ? LDS : 4)] = (uint8_t)a;
        __syncthreads();
        out[threadIdx.x] = a + s[(threadIdx.x + 1) % (LDS > 0 ? LDS : 4)];
    }
}

template <int LDS, int VG>
void run(const char* name, int grid, int threads) {
    uint32_t *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<LDS, VG>), dim3(grid), dim3(threads), 0, 0, in, out, 1);
        hipEventRecord(e1); hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s grid %6d x %3d threads  LDS %5d B  %7.1f us per launch  %6.1f ns per workgroup\n", name, grid, threads, LDS,
           ms / 20 * 1e3, ms / 20 * 1e6 / grid);
    hipFree(in); hipFree(out);
}

int main() {
    run<0, 8>("no LDS", 8040, 256);
    run<8192, 8>("8 KB LDS", 8040, 256);
    run<27136, 8>("27 KB LDS", 8040, 256);
    run<27136, 60>("27 KB LDS, 60+ VGPRs", 8040, 256);
    run<27136, 8>("27 KB LDS, 64 threads", 32160, 64);
    run<6784, 8>("6.8 KB LDS, 64 threads", 32160, 64);
    run<0, 8>("no LDS, 64 threads", 32160, 64);
    run<27136, 8>("27 KB LDS, 4K grid", 32400, 256);
    run<54272, 8>("54 KB LDS, 512 threads", 4020, 512);
    return 0;
}
