// me_fast.hpp -- k_me16: the motion-search kernel for the BASELINE operating
// point (16x16 blocks, +-32 px window, stride-5 grid: motion.py:18,132 for
// block_size = 16).  Same results as k_me_generic, bit for bit.
//
// Design (one wavefront per macroblock, lane = candidate):
//   * 5-pixel groups.  Candidates step by 5 px = 15 bytes, which never lines
//     up with a dword.  The reference window is therefore parked in LDS with
//     one pad byte after every 5 pixels (15 -> 16 bytes), counted from the
//     window's left edge j_min -- exactly the anchor of the candidate grid
//     (motion.py:126,138).  Candidate cj then starts at byte 16*cj of a row:
//     every candidate row is four aligned ds_read_b128, no byte shifting.
//   * The current block is wave-uniform: it is fetched with scalar loads and
//     re-packed into the same 5-pixel-group format on the scalar ALU, so the
//     vector ALU only sees  v_lerp_u8 / v_and / v_bcnt / v_sad_u8.
//   * Cost without a per-byte subtract.  (r - c) mod 256 = r - c + 256*[r < c],
//     so  cost = sum(r) - sum(c) + 256 * #{r < c}.  #{r >= c} comes from bit 7
//     of v_lerp_u8(r, ~c, 1) = (r - c + 256) >> 1 per byte (one v_and + one
//     v_bcnt), sum(r) from v_sad_u8(r, 0); pad bytes carry ~c = 0xFF (always
//     counted, subtracted as a constant) and r = 0.
//   * LDS row pitch 18 slots of 16 B: with lane = 10*ci + cj the slot index is
//     congruent to the lane index mod 16, which makes every ds_read_b128 lane
//     group hit 16 distinct 16-byte slots (conflict-free; MI355X_MICROARCH.md
//     LDS table).
//   * 100 candidates = two passes of 64 lanes; both passes share each
//     re-packed row of the current block.
#pragma once
#include <stdlib.h>

#include "vcs_common.hpp"

namespace vcs {

constexpr int kMe16Rows = 64;                 // 2 * win
constexpr int kMe16Groups = 13;               // ceil(64 px / 5)

__device__ __forceinline__ uint4 load16_any_align(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}

// Keeps a wave-uniform value in a scalar register and hides it from the
// optimiser, so that the shift/or re-packing below stays on the scalar ALU
// (hipcc would otherwise fuse it into v_alignbit on the vector ALU).
__device__ __forceinline__ uint32_t sgpr_opaque(uint32_t v) {
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ uint32_t funnel_s(uint32_t lo, uint32_t hi, int sh) {
    return sgpr_opaque(lo >> sh) | sgpr_opaque(hi << (32 - sh));
}

// One row (13 dwords) of one candidate against the re-packed current row.
__device__ __forceinline__ void me16_row(const uint8_t *lds_row, const uint32_t (&nc)[13],
                                         uint32_t &cnt, uint32_t &sum) {
    const uint4 *p = (const uint4 *)lds_row;
    uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    uint32_t r[13] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                      q2.x, q2.y, q2.z, q2.w, q3.x & 0x00FFFFFFu};
#pragma unroll
    for (int d = 0; d < 13; d++) {
        uint32_t t = __builtin_amdgcn_lerp(r[d], nc[d], 0x01010101u) & 0x80808080u;
        // v_bcnt_u32_b32 with its accumulate operand (hipcc splits popcount + add
        // into bcnt(x, 0) and extra v_add3 otherwise)
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt) : "v"(t));
        sum = __builtin_amdgcn_sad_u8(r[d], 0u, sum);
    }
}

// SLOTS: 16-byte slots per LDS window row (18 = conflict-free, 13/14 = denser).
template <int SLOTS>
__global__ __launch_bounds__(64) void k_me16(
    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ curs, size_t cur_stride,
    Geom g, int16_t *__restrict__ mv, uint32_t *__restrict__ n_static,
    unsigned long long *__restrict__ cost_out) {
    constexpr int kMe16RowB = SLOTS * 16;
    __shared__ __attribute__((aligned(16))) uint8_t s_win[kMe16Rows * kMe16RowB];
    constexpr int BS = 16, WIN = 32, STEP = 5;
    const int lane = threadIdx.x;
    const int blk = blockIdx.x, f = blockIdx.y;
    const uint8_t *__restrict__ cur = curs + (size_t)f * cur_stride;
    const int bxi = blk % g.nbx, byi = blk / g.nbx;
    const int x = bxi * BS, y = byi * BS;
    const size_t blk_off = (size_t)y * g.pitch + (size_t)(48 * bxi);

    // ---- issue every global load of this block up front -----------------------
    // static-test operands (3 dwords per lane of the co-located ref / cur block)
    uint32_t rv[3], cv[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int idx = lane + 64 * t, row = idx / 12, col = idx - row * 12;
        const size_t off = blk_off + (size_t)row * g.pitch + (size_t)(4 * col);
        rv[t] = *(const uint32_t *)(ref + off);
        cv[t] = *(const uint32_t *)(cur + off);
    }
    // the search window (motion.py:125-129), fetched speculatively: 13 unaligned
    // 16-byte loads per lane are in flight while the static test is evaluated
    const int i_min = max(y - WIN, 0), j_min = max(x - WIN, 0);
    const int i_max = min(y + WIN, g.H), j_max = min(x + WIN, g.W);
    const int wrows = i_max - i_min;
    const uint8_t *__restrict__ wsrc = ref + (size_t)i_min * g.pitch + (size_t)(3 * j_min);
    uint4 stg[kMe16Groups];
#pragma unroll
    for (int t = 0; t < kMe16Groups; t++) {
        const int e = lane + 64 * t, row = e / kMe16Groups, gi = e - row * kMe16Groups;
        // rows past the clipped window are clamped (never read back), so the
        // load is unconditional: a branch here would cost a vmcnt(0) per load
        const int rc = min(row, wrows - 1);
        stg[t] = load16_any_align(wsrc + (size_t)rc * g.pitch + (size_t)(15 * gi));
    }

    // ---- static test (motion.py:109-116): sum(max(r - c, 0)) -------------------
    //   max(r - c, 0) = (|r - c| + r - c) / 2 per byte -> three v_sad_u8 per dword
    uint32_t s_abs = 0, s_r = 0, s_c = 0;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        s_abs = __builtin_amdgcn_sad_u8(rv[t], cv[t], s_abs);
        s_r = __builtin_amdgcn_sad_u8(rv[t], 0u, s_r);
        s_c = __builtin_amdgcn_sad_u8(cv[t], 0u, s_c);
    }
    s_abs = wave_sum_u32(s_abs);
    s_r = wave_sum_u32(s_r);
    s_c = wave_sum_u32(s_c);  // sum of the current block's bytes
    const uint32_t sdiff = (s_abs + s_r - s_c) >> 1;
    const size_t oidx = (size_t)f * g.N + blk;
    if (!g.never_static && sdiff <= g.thresh) {
        if (lane == 0) {
            mv[2 * oidx + 0] = 0;
            mv[2 * oidx + 1] = 0;
            atomicAdd(&n_static[f], 1u);
            if (cost_out) cost_out[oidx] = sdiff;
        }
        return;
    }

    // ---- park the window in 5-pixel groups ---------------------------------------
#pragma unroll
    for (int t = 0; t < kMe16Groups; t++) {
        const int e = lane + 64 * t, row = e / kMe16Groups, gi = e - row * kMe16Groups;
        uint4 v = stg[t];
        v.w &= 0x00FFFFFFu;  // pad byte = 0 so that it adds nothing to sum(r)
        if (row < kMe16Rows) *(uint4 *)(s_win + row * kMe16RowB + gi * 16) = v;
    }
    __syncthreads();

    // ---- candidate grid (motion.py:132-140) ---------------------------------------
    const int ri = i_max - BS - i_min, rj = j_max - BS - j_min;
    const int nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
    const int ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
    const int ncand = nci * ncj;  // <= 100
    const int cand0 = lane, cand1 = lane + 64;
    const bool act0 = cand0 < ncand, act1 = cand1 < ncand;
    const bool two = ncand > 64;  // wave-uniform
    int ci0 = 0, cj0 = 0, ci1 = 0, cj1 = 0;
    if (act0) { ci0 = cand0 / ncj; cj0 = cand0 - ci0 * ncj; }
    if (act1) { ci1 = cand1 / ncj; cj1 = cand1 - ci1 * ncj; }
    const uint8_t *p0 = s_win + (STEP * ci0) * kMe16RowB + cj0 * 16;
    const uint8_t *p1 = s_win + (STEP * ci1) * kMe16RowB + cj1 * 16;

    uint32_t cnt0 = 0, sum0 = 0, cnt1 = 0, sum1 = 0;
    const uint32_t *__restrict__ crow = (const uint32_t *)(cur + blk_off);
    const int pitch_dw = g.pitch >> 2;
    uint32_t d[12];
#pragma unroll
    for (int i = 0; i < 12; i++) d[i] = crow[i];
#pragma unroll 4
    for (int k = 0; k < BS; k++) {
        // current row k: 48 wave-uniform bytes -> scalar loads (row k+1 is
        // requested before row k is consumed), re-packed into 5-pixel groups
        // (pad byte -> ~c = 0xFF) on the scalar ALU
        const uint32_t *cr = crow + (size_t)(k + 1 < BS ? k + 1 : k) * pitch_dw;
        uint32_t dn[12];
#pragma unroll
        for (int i = 0; i < 12; i++) dn[i] = cr[i];
        uint32_t nc[13];
        nc[0] = ~d[0];
        nc[1] = ~d[1];
        nc[2] = ~d[2];
        nc[3] = ~d[3] | 0xFF000000u;
        nc[4] = ~funnel_s(d[3], d[4], 24);
        nc[5] = ~funnel_s(d[4], d[5], 24);
        nc[6] = ~funnel_s(d[5], d[6], 24);
        nc[7] = ~funnel_s(d[6], d[7], 24) | 0xFF000000u;
        nc[8] = ~funnel_s(d[7], d[8], 16);
        nc[9] = ~funnel_s(d[8], d[9], 16);
        nc[10] = ~funnel_s(d[9], d[10], 16);
        nc[11] = ~funnel_s(d[10], d[11], 16) | 0xFF000000u;
        nc[12] = ~(d[11] >> 8) | 0xFF000000u;
        me16_row(p0 + k * kMe16RowB, nc, cnt0, sum0);
        if (two) me16_row(p1 + k * kMe16RowB, nc, cnt1, sum1);
#pragma unroll
        for (int i = 0; i < 12; i++) d[i] = dn[i];
    }
    // cost = sum(r) - sum(c) + 256 * #{r < c}; 64 pad bytes were always counted
    const uint32_t n_lt0 = 768u + 64u - cnt0, n_lt1 = 768u + 64u - cnt1;
    const uint32_t cost0 = sum0 + 256u * n_lt0 - s_c;
    const uint32_t cost1 = sum1 + 256u * n_lt1 - s_c;
    uint32_t key = 0xFFFFFFFFu;  // (cost << 7) | raster index: motion.py:149 order
    if (act0) key = (cost0 << 7) | (uint32_t)cand0;
    if (act1) key = min(key, (cost1 << 7) | (uint32_t)cand1);
    key = wave_min_u32(key);

    if (lane == 0) {
        int bx = 0, by = 0;                    // motion.py:102
        unsigned long long c = 9999999999ull;  // motion.py:120
        if (key != 0xFFFFFFFFu) {
            const int cand = (int)(key & 127u);
            by = i_min + (cand / ncj) * STEP;
            bx = j_min + (cand % ncj) * STEP;
            c = key >> 7;
        }
        mv[2 * oidx + 0] = (int16_t)(bx - x);
        mv[2 * oidx + 1] = (int16_t)(by - y);
        if (cost_out) cost_out[oidx] = c;
    }
}

// The fast kernel covers the reference's own parameters for block_size 16 on
// frames whose rows are dword-aligned; anything else takes k_me_generic.
inline bool me_fast_supported(const Geom &g) {
    return g.bs == 16 && g.win == 32 && g.step == 5 && (g.W % 4) == 0 && g.N > 0;
}

inline int launch_me_fast(const Geom &g, hipStream_t stream, const uint8_t *ref,
                          const uint8_t *curs, size_t stride, int n, int16_t *mv, uint32_t *ns,
                          unsigned long long *cost) {
    static const int slots = [] {
        const char *e = getenv("VCS_ME16_SLOTS");  // tuning knob: 13, 14 or 18
        int v = e ? atoi(e) : 18;
        return (v == 13 || v == 14) ? v : 18;
    }();
    dim3 grid(g.N, n), block(64);
    if (slots == 13)
        hipLaunchKernelGGL(k_me16<13>, grid, block, 0, stream, ref, curs, stride, g, mv, ns, cost);
    else if (slots == 14)
        hipLaunchKernelGGL(k_me16<14>, grid, block, 0, stream, ref, curs, stride, g, mv, ns, cost);
    else
        hipLaunchKernelGGL(k_me16<18>, grid, block, 0, stream, ref, curs, stride, g, mv, ns, cost);
    return 0;
}

}  // namespace vcs
