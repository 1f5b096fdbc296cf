// Which XCD does workgroup i of a 2,048-workgroup launch land on, and does it depend on
// the kernel that ran before it on the same stream?  (k_me16r's unit dealing used to take
// blockIdx.x % 8 for the XCD.)   hipcc --offload-arch=gfx950 -O2 -o tools/xcc_map tools/xcc_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_prev(unsigned *sink) {
    __shared__ unsigned s[2432];
    s[threadIdx.x] = blockIdx.x;
    __syncthreads();
    if (threadIdx.x == 0 && s[1] == 0xFFFFFFFFu) sink[0] = 1;
}
__global__ __launch_bounds__(64) void k_probe(unsigned *out, unsigned *cu) {
    __shared__ unsigned s[4992];  // 19.9 KB: eight workgroups fill a CU's LDS, as k_me16r's do
    unsigned x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    s[threadIdx.x] = x;
    __syncthreads();
    // stay resident for a while so that all 2,048 workgroups coexist, as in the real kernel
    unsigned long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
    if (threadIdx.x == 0) { out[blockIdx.x] = s[0] & 15u; cu[blockIdx.x] = h; }
}
int main() {
    unsigned *out, *cu, *sink;
    hipMalloc(&out, 2048 * 4); hipMalloc(&cu, 2048 * 4); hipMalloc(&sink, 4);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int prevs[] = {0, 2048, 4096, 8040, 8041, 117450, 1};
    for (int pg : prevs) {
        for (int rep = 0; rep < 2; rep++) {
            if (pg) hipLaunchKernelGGL(k_prev, dim3(pg), dim3(64), 0, st, sink);
            hipLaunchKernelGGL(k_probe, dim3(2048), dim3(64), 0, st, out, cu);
            hipStreamSynchronize(st);
            std::vector<unsigned> h(2048);
            hipMemcpy(h.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
            int match = 0, cnt[16] = {0};
            for (int i = 0; i < 2048; i++) { match += (h[i] == (unsigned)(i % 8)); cnt[h[i] & 15]++; }
            printf("prev grid %6d rep %d: xcc == wg%%8 for %4d / 2048; first 16 wgs:", pg, rep, match);
            for (int i = 0; i < 16; i++) printf(" %u", h[i]);
            printf("; per-xcc counts:");
            for (int i = 0; i < 8; i++) printf(" %d", cnt[i]);
            printf("\n");
            if (rep == 1 && (pg == 2048 || pg == 8040)) {
                std::vector<unsigned> c(2048);
                hipMemcpy(c.data(), cu, 2048 * 4, hipMemcpyDeviceToHost);
                printf("  wgs of one XCD in id order -> (se.cu simd):");
                int shown = 0;
                for (int i = 0; i < 2048 && shown < 72; i++)
                    if (h[i] == h[0]) {
                        printf(" %u.%u/%u", (c[i] >> 13) & 7u, (c[i] >> 8) & 15u, (c[i] >> 4) & 3u);
                        shown++;
                    }
                printf("\n");
            }
        }
    }
    return 0;
}
