// Which physical CUs does a stream created with hipExtStreamCreateWithCUMask use on MI355X
// (8 XCDs x 32 CUs)?  For masks with the first c bits set and for strided masks: the set of
// (xcc, se, cu) the workgroups of a 4,096-workgroup launch land on.
//   hipcc --offload-arch=gfx950 -O2 -o tools/cu_mask_probe tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ __launch_bounds__(64) void k_probe(unsigned *xcc, unsigned *hw) {
    unsigned x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    unsigned long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) { xcc[blockIdx.x] = x & 15u; hw[blockIdx.x] = h; }
}
static void run(const char *name, const std::vector<uint32_t> &mask) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    const int G = 4096;
    unsigned *xcc, *hw;
    hipMalloc(&xcc, G * 4); hipMalloc(&hw, G * 4);
    hipLaunchKernelGGL(k_probe, dim3(G), dim3(64), 0, st, xcc, hw);
    hipStreamSynchronize(st);
    std::vector<unsigned> hx(G), hh(G);
    hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hh.data(), hw, G * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    int per_xcc[8] = {0};
    std::set<unsigned> per_xcc_cus[8];
    for (int i = 0; i < G; i++) {
        unsigned se = (hh[i] >> 13) & 7u, cu = (hh[i] >> 8) & 15u;
        cus.insert((hx[i] << 8) | (se << 4) | cu);
        per_xcc_cus[hx[i] & 7].insert((se << 4) | cu);
    }
    printf("%-28s distinct CUs used: %3zu; per XCC:", name, cus.size());
    for (int x = 0; x < 8; x++) printf(" %zu", per_xcc_cus[x].size());
    printf("\n");
    hipFree(xcc); hipFree(hw);
    hipStreamDestroy(st);
}
int main() {
    std::vector<uint32_t> m(8, 0);
    auto first = [&](int c) { std::vector<uint32_t> v(8, 0); for (int i = 0; i < c; i++) v[i / 32] |= 1u << (i % 32); return v; };
    char nm[64];
    for (int c : {8, 16, 32, 64, 128, 192, 256}) { snprintf(nm, 64, "first %d bits", c); run(nm, first(c)); }
    { std::vector<uint32_t> v(8, 0); for (int i = 64; i < 256; i++) v[i / 32] |= 1u << (i % 32); run("bits 64..255", v); }
    { std::vector<uint32_t> v(8, 0); for (int i = 0; i < 256; i += 4) v[i / 32] |= 1u << (i % 32); run("every 4th bit", v); }
    { std::vector<uint32_t> v(8, 0); for (int i = 0; i < 256; i++) if (i % 4) v[i / 32] |= 1u << (i % 32); run("all but every 4th bit", v); }
    { std::vector<uint32_t> v(8, 0); for (int i = 0; i < 256; i++) if ((i / 8) % 4 == 0) v[i / 32] |= 1u << (i % 32); run("8 of every 32 bits", v); }
    return 0;
}
