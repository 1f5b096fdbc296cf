// vcs_common.hpp -- shared device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- experiment builds -------------------------------------------------------------------
// The VCS_EXP_* switches (timing probes, and variants that compute WRONG results on purpose to
// price a part of a kernel) exist only under the umbrella VCS_EXPERIMENT_BUILD, which
// `make EXPERIMENT="-DVCS_EXP_..."` defines and vcs_experiment_build() reports: a library built
// with any of them says so, and tests/test_host_cpu.py asserts the shipped one was not.  Every switch
// used anywhere under csrc/ must be named here (the same test greps for strays).
#if defined(VCS_EXP_PHASES) || defined(VCS_EXP_REF_NT) || defined(VCS_EXP_BITS_NOFOLD) || defined(VCS_EXP_NOCUR) || \
    defined(VCS_EXP_NOWIN) || defined(VCS_EXP_NOAND) || defined(VCS_EXP_NOBARRIER) || defined(VCS_EXP_NOREDUCE) ||   \
    defined(VCS_EXP_NOCORE) || defined(VCS_EXP_BITS_SAMEC) || defined(VCS_EXP_PLAIN_STORES) || defined(VCS_EXP_IDCT_WPE)
#ifndef VCS_EXPERIMENT_BUILD
#error "VCS_EXP_* switches need -DVCS_EXPERIMENT_BUILD (make EXPERIMENT=...): the library must be able to say what it is"
#endif
#endif

namespace vcs {

// Frame / block geometry shared by all kernels (filled on the host).
struct Geom {
    int H, W;        // frame size in pixels
    int pitch;       // bytes per frame row = 3 * W
    int bs;          // motion block size
    int nbx, nby, N; // full blocks per row / column / frame (motion.py:74-98)
    int win, step;   // search half-window and stride (motion.py:18, :132)
    uint32_t thresh; // static threshold (motion.py:8), clamped to u32
    int never_static;// thresh < 0: no block can be static
    int wpitch;      // LDS bytes per staged window row (generic ME kernel)
    // x / bs and strip / groups_x for the tail kernels without a division sequence (magic_u32 below)
    uint32_t bs_magic, gx_magic;
};

// x / d for x * d < 2^32 as one v_mul_hi_u32 / s_mul_hi_u32: magic = floor(2^32 / d) + 1 (d >= 2);
// magic 0 stands for d = 1.  (magic * d = 2^32 + e with 0 < e <= d, so the product's high word is
// floor(x / d + x e / (d 2^32)) and x e < 2^32 keeps the second term below 1 / d.)  The runtime
// divisors of the tail kernels -- motion block size, 64-pixel groups per row -- took ~25 vector
// instructions per division, a third of k_idct's and a quarter of k_fdct's instructions before the
// first load was issued.
inline uint32_t magic_u32(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((1ull << 32) / d + 1ull); }
__device__ __forceinline__ uint32_t div_magic(uint32_t x, uint32_t magic) {
    return magic ? __umulhi(x, magic) : x;
}

// ---- wave64 reductions ---------------------------------------------------
// DPP inside each 16-lane row (4 VALU ops, no LDS crossbar round trips), then
// four v_readlane + scalar ops across the rows.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += dpp_u32<0xB1>(v);   // quad_perm:[1,0,3,2]
    v += dpp_u32<0x4E>(v);   // quad_perm:[2,3,0,1]
    v += dpp_u32<0x141>(v);  // row_half_mirror
    v += dpp_u32<0x140>(v);  // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_u32<0xB1>(v));
    v = min(v, dpp_u32<0x4E>(v));
    v = min(v, dpp_u32<0x141>(v));
    v = min(v, dpp_u32<0x140>(v));
    return min(min((uint32_t)__builtin_amdgcn_readlane(v, 0), (uint32_t)__builtin_amdgcn_readlane(v, 16)),
               min((uint32_t)__builtin_amdgcn_readlane(v, 32), (uint32_t)__builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}

// ---- LDS hand-off between the lanes of ONE wave ---------------------------
// Every kernel here runs single-wave workgroups.  The DS operations of a wave
// execute in order, so lanes can exchange data through LDS without s_barrier;
// what is needed is only that the compiler keeps the write -> read order.
// __syncthreads() must NOT be used for this: its fence also drains vmcnt, i.e.
// it waits for every outstanding global load (the register prefetch of the next
// block in k_me16) and global store (the coefficient planes in k_fdct).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- buffer loads: SGPR resource + 32-bit lane offset + uniform offset ----------------
// The flat-address form of a load keeps every per-lane byte offset as a 64-bit register
// pair and spends a 64-bit vector add per load to form the address.  A raw buffer load
// takes the (wave-uniform) base as a 4-SGPR resource, a 32-bit per-lane offset and a
// uniform offset in an SGPR; reads past `bytes` return 0 instead of faulting.  Unaligned
// dword / 16-byte accesses are fine on gfx950 (tools/buf_test.hip).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t buf_ld32(__amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t uni_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)uni_off, 0);
}
__device__ __forceinline__ uint4 buf_ld128(__amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t uni_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)uni_off, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- packed-byte arithmetic ---------------------------------------------
// Per-byte (a - b) mod 256 on four packed bytes, no borrow across bytes.
__device__ __forceinline__ uint32_t sub_u8x4(uint32_t a, uint32_t b) {
    const uint32_t Hm = 0x80808080u;
    return ((a | Hm) - (b & ~Hm)) ^ ((a ^ ~b) & Hm);
}
// Per-byte (a + b) mod 256.
__device__ __forceinline__ uint32_t add_u8x4(uint32_t a, uint32_t b) {
    const uint32_t Hm = 0x80808080u;
    return ((a & ~Hm) + (b & ~Hm)) ^ ((a ^ b) & Hm);
}
// acc + sum of the four bytes of v  (v_sad_u8 against zero).
__device__ __forceinline__ uint32_t bytesum_acc(uint32_t v, uint32_t acc) {
    return __builtin_amdgcn_sad_u8(v, 0u, acc);
}

// ---- OpenCV 8-bit BGR <-> YCrCb (cv2.cvtColor at DCTcompressor.py:55/:92) --
// Restated from OpenCV 4.x imgproc/src/color_yuv.simd.hpp; third-party code,
// parity unpinned against a real OpenCV binary.
__device__ __forceinline__ int descale14(int v) { return (v + (1 << 13)) >> 14; }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ void bgr2ycc(int B, int G, int R, int &Y, int &Cr, int &Cb) {
    int y = descale14(R * 4899 + G * 9617 + B * 1868);
    Cr = sat8(descale14((R - y) * 11682 + (128 << 14)));
    Cb = sat8(descale14((B - y) * 9241 + (128 << 14)));
    Y = sat8(y);
}
__device__ __forceinline__ void ycc2bgr(int Y, int Cr, int Cb, int &B, int &G, int &R) {
    Cr -= 128;
    Cb -= 128;
    B = sat8(Y + descale14(Cb * 29049));
    G = sat8(Y + descale14(Cb * -5636 + Cr * -11698));
    R = sat8(Y + descale14(Cr * 22987));
}

}  // namespace vcs
