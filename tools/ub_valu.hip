// Micro-benchmark: issue cost (shader cycles per wave-instruction) of the VALU
// operations k_me16's row loop is made of, on one and on several waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o ub_valu tools/ub_valu.hip && ./ub_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define REP 64
template <int OP>
__global__ __launch_bounds__(64) void k(uint32_t *out, unsigned long long *cyc, int iters) {
    uint32_t a[16], b = threadIdx.x * 2654435761u, c = 0x01010101u;
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 40503u + i * 7919u + blockIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 16; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_sad_u8 %0, %1, 0, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 6) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 7) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
                if (OP == 8) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 9) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 10) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 11) asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(a[i]) : "v"(b));
                if (OP == 12) asm volatile("v_lshl_or_b32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 14) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 15) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 5) {  // the row-loop mix, dependent chain per element: lerp -> and -> bcnt
                    uint32_t t;
                    asm volatile("v_lerp_u8 %0, %1, %2, %3" : "=v"(t) : "v"(a[i]), "v"(b), "v"(c));
                    asm volatile("v_and_b32 %0, 0x80808080, %0" : "+v"(t));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(t));
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, int blocks, int per_it) {
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double n = (double)iters * REP * per_it;
    printf("%-22s waves=%5d  memtime-ticks/inst (median wave) %.3f   wall %.3f ms -> %.1f G wave-inst/s chip\n",
           name, blocks, h[blocks / 2] / n, ms, n * blocks / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {2048}) {
        run<6>("v_and_or_b32", blocks, 1);
        run<7>("v_lshrrev_b32", blocks, 1);
        run<8>("v_add3_u32", blocks, 1);
        run<9>("v_bfi_b32", blocks, 1);
        run<10>("v_perm_b32", blocks, 1);
        run<11>("v_alignbit_b32", blocks, 1);
        run<12>("v_lshl_or_b32", blocks, 1);
        run<14>("v_mad_u32_u24", blocks, 1);
        run<15>("v_msad_u8", blocks, 1);
    }
    for (int blocks : {2048}) {
        run<0>("v_lerp_u8", blocks, 1);
        run<1>("v_and_b32", blocks, 1);
        run<2>("v_bcnt_u32_b32", blocks, 1);
        run<3>("v_sad_u8", blocks, 1);
        run<4>("v_add_u32", blocks, 1);
        run<5>("lerp->and->bcnt x16", blocks, 3);
    }
    return 0;
}
