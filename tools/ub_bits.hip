// Micro-benchmark for a bit-sliced [r<c] count: lane = candidate keeps 24 plane dwords of the window in
// registers, the current block's planes are wave-uniform (SGPRs through s_load), looped over NF frames.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#ifndef NF
#define NF 8
#endif
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }

__global__ __launch_bounds__(64) void kbody(const uint32_t* __restrict__ rch, const uint32_t* __restrict__ cpl,
                                            uint32_t* out, unsigned long long* cyc, int rounds) {
    const int lane = threadIdx.x;
    uint32_t acc[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) acc[f] = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rd = 0; rd < rounds; rd++) {
        uint32_t r[24];
        const uint4* rp = (const uint4*)(rch + ((size_t)(blockIdx.x & 63) * 16 + (rd & 15)) * 64 * 24 + lane * 24);
#pragma unroll
        for (int q = 0; q < 6; q++) { uint4 v = rp[q]; r[4*q]=v.x; r[4*q+1]=v.y; r[4*q+2]=v.z; r[4*q+3]=v.w; }
        const uint32_t* cb = cpl + (size_t)((blockIdx.x * 16 + (rd & 15)) & 1023) * NF * 24;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const uint32_t* c = cb + f * 24;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                uint32_t b = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    uint32_t cc = c[k * 3 + d];
                    uint32_t x = r[k * 3 + d] ^ cc;
                    b = bfi(x, cc, b);
                }
                acc[f] += __builtin_popcount(b);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
#pragma unroll
    for (int f = 0; f < NF; f++) s += acc[f] * (f + 1);
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
__global__ __launch_bounds__(64) void kop(uint32_t* out, unsigned long long* cyc, int iters, uint32_t sarg) {
    uint32_t a[16], b = threadIdx.x * 2654435761u, c = 0x01010101u;
    uint32_t s = __builtin_amdgcn_readfirstlane(sarg);
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 40503u + i * 7919u + blockIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (OP == 2) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 3) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(s));
                if (OP == 4) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 5) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 6) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 7) asm volatile("v_xnor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
                if (OP == 8) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(s));
                if (OP == 9) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 10) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 11) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n" :: "v"(a[i]), "v"(b) : "vcc");
                if (OP == 12) asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(a[i]) : "v"(b));
                if (OP == 13) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 14) asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void runop(const char* name, int blocks) {
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(kop<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, 10, 0x12345u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kop<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 0x12345u);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double n = (double)iters * 64;
    // SIMD-cycles per instruction from wall: blocks waves over 1024 SIMDs
    printf("%-20s waves=%5d ticks/inst/wave %.3f  wall %.3f ms  chip %.1f G wave-inst/s\n", name, blocks,
           h[blocks / 2] / n, ms, n * blocks / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1024, 4096, 8192}) {
        runop<0>("v_xor vv", blocks); runop<1>("v_xor sv", blocks);
        runop<2>("v_bfi vvv", blocks); runop<3>("v_bfi vsv", blocks);
        runop<4>("v_bcnt", blocks); runop<5>("v_and", blocks); runop<6>("v_lerp_u8", blocks);
        runop<7>("v_xnor sv", blocks); runop<8>("v_and_or vsv", blocks); runop<9>("v_add_u32", blocks);
        runop<10>("v_sad_u8", blocks); runop<11>("v_cmp_lt_u32", blocks); runop<12>("v_alignbit", blocks);
        runop<13>("v_perm", blocks); runop<14>("v_or3", blocks); runop<15>("v_mov", blocks);
    }
    // body
    size_t rn = 64 * 16 * 64 * 24, cn = 1024 * NF * 24;
    uint32_t *rch, *cpl, *out; unsigned long long* cyc;
    hipMalloc(&rch, rn * 4); hipMalloc(&cpl, cn * 4);
    std::vector<uint32_t> hr(rn), hc(cn);
    uint32_t x = 12345; for (auto& v : hr) { x = x * 1664525u + 1013904223u; v = x; }
    for (auto& v : hc) { x = x * 1664525u + 1013904223u; v = x; }
    hipMemcpy(rch, hr.data(), rn * 4, hipMemcpyHostToDevice); hipMemcpy(cpl, hc.data(), cn * 4, hipMemcpyHostToDevice);
    for (int blocks : {1024, 2048, 4096, 6144, 8192}) {
        hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
        int rounds = 400;
        hipLaunchKernelGGL(kbody, dim3(blocks), dim3(64), 0, 0, rch, cpl, out, cyc, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kbody, dim3(blocks), dim3(64), 0, 0, rch, cpl, out, cyc, rounds);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        double bodies = (double)rounds * NF;  // per wave: 3 dwords x 32 pairs = 96 pairs/lane
        double pairs = bodies * blocks * 64 * 96;
        printf("body NF=%d waves=%5d ticks/body/wave %.1f  wall %.3f ms  chip %.2f T pairs/s  (SIMD-cyc/body at 2.1GHz %.1f)\n", NF,
               blocks, h[blocks / 2] / bodies, ms, pairs / (ms * 1e-3) / 1e12,
               ms * 1e-3 * 2.1e9 * 1024 / (bodies * blocks));
        hipFree(out); hipFree(cyc);
    }
    return 0;
}
