// me_fast.hpp -- k_me16: the motion-search kernel for the BASELINE operating
// point (16x16 blocks, +-32 px window, stride-5 grid: motion.py:18,132 for
// block_size = 16).  Same results as k_me_generic, bit for bit.
//
// Design (one wavefront per macroblock at a time, lane = candidate):
//   * 5-pixel groups.  Candidates step by 5 px = 15 bytes, which never lines
//     up with a dword.  The reference window is therefore parked in LDS with
//     one pad byte after every 5 pixels (15 -> 16 bytes), counted from the
//     window's left edge j_min -- exactly the anchor of the candidate grid
//     (motion.py:126,138).  Candidate cj then starts at byte 16*cj of a row:
//     every candidate row is aligned ds_read_b128s, no byte shifting.
//   * Cost without a per-byte subtract.  (r - c) mod 256 = r - c + 256*[r < c],
//     so  cost = sum(r) - sum(c) + 256 * #{r < c}.  #{r >= c} comes from bit 7
//     of v_lerp_u8(r, ~c, 1) = (r - c + 256) >> 1 per byte: one v_lerp_u8, one
//     v_and, one v_bcnt per dword.  Pad bytes carry ~c = 0xFF (always counted,
//     subtracted as a constant).
//   * sum(r) does not depend on the current frame.  Every P frame of a GOP is
//     searched against the same raw I frame (encoder.py:51-52), so the byte
//     sum of each (block, candidate) reference patch is computed ONCE per
//     reference frame (k_me16_sums, 100 x uint32 per block) and read back by
//     the search: the row loop holds no v_sad_u8 at all.
//   * The current block is wave-uniform.  Its 16 rows are re-packed once per
//     block into the same 5-pixel-group format (negated, pad byte 0xFF), lane-
//     parallel, parked in 1 KB of LDS and read back with wave-uniform
//     (broadcast) ds_read_b128: no scalar loads in the row loop (SMEM shares
//     lgkmcnt with LDS and would force full drains), no re-packing there.
//   * LDS row pitch 18 slots of 16 B: with lane = 10*ci + cj the slot index is
//     congruent to the lane index mod 16, which makes every ds_read_b128 lane
//     group hit 16 distinct 16-byte slots (conflict-free; MI355X_MICROARCH.md
//     LDS table).
//   * 100 candidates = two passes of 64 lanes (64 + 36) over the same rows.
//   * Persistent waves with register prefetch.  The kernel is bound by vector
//     ALU issue and LDS admits only two waves per SIMD, so nothing else may
//     stall: each wave walks a strided list of blocks and issues ALL global
//     loads of block i+1 (into registers, "uniform base + 32-bit lane offset"
//     form) before it searches block i.  Work is dealt so that the blocks of
//     one XCD are contiguous in the frame (reference-window reuse in that
//     XCD's L2); the result never depends on the mapping.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include <hip/hip_ext.h>

#include "vcs_common.hpp"

namespace vcs {

constexpr int kMe16Rows = 64;        // 2 * win
constexpr int kMe16Groups = 13;      // ceil(64 px / 5)
constexpr int kMe16SumStride = 128;  // uint32 per block in the sum(r) table (100 used)

__device__ __forceinline__ uint4 load16_any_align(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// 16 dwords of LDS (four ds_read_b128) starting at p.
struct Me16Q {
    uint4 q0, q1, q2, q3;
};
__device__ __forceinline__ void me16_ld(Me16Q &Q, const void *p) {
    const uint4 *a = (const uint4 *)p;
    Q.q0 = a[0]; Q.q1 = a[1]; Q.q2 = a[2]; Q.q3 = a[3];
}
// Only the first dword of the 4th read is used.  Pretending to use the rest AT
// THE POINT OF CONSUMPTION keeps it a conflict-free ds_read_b128 (as a
// ds_read_b32 the 16-byte lane stride would land 32 lanes on 8 banks) without
// forcing a wait right behind the load.
__device__ __forceinline__ void me16_keep4(const Me16Q &Q) {
    asm volatile("" ::"v"(Q.q3.y), "v"(Q.q3.z), "v"(Q.q3.w));
}

// #{r >= c} over the 13 dwords of one candidate row.  A dependent VALU
// instruction issues ~1.7x later than an independent one on gfx950, so the
// three steps are written as three passes over the 13 dwords (13 independent
// v_lerp_u8, 13 independent v_and, then v_bcnt into 4 rotating accumulators)
// and pinned in that order with a scheduling fence between the passes.
// v_bcnt_u32_b32 is written with its accumulate operand (hipcc splits
// popcount + add into bcnt(x, 0) and extra v_add3 otherwise).
__device__ __forceinline__ void me16_count13(const Me16Q &a, const Me16Q &c, uint32_t (&cnt)[4]) {
    const uint32_t r[13] = {a.q0.x, a.q0.y, a.q0.z, a.q0.w, a.q1.x, a.q1.y, a.q1.z,
                            a.q1.w, a.q2.x, a.q2.y, a.q2.z, a.q2.w, a.q3.x};
    const uint32_t nc[13] = {c.q0.x, c.q0.y, c.q0.z, c.q0.w, c.q1.x, c.q1.y, c.q1.z,
                             c.q1.w, c.q2.x, c.q2.y, c.q2.z, c.q2.w, c.q3.x};
    me16_keep4(a);
    uint32_t t[13];
#pragma unroll
    for (int d = 0; d < 13; d++) t[d] = __builtin_amdgcn_lerp(r[d], nc[d], 0x01010101u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < 13; d++) t[d] &= 0x80808080u;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < 13; d++) {
        // selects to v_bcnt_u32_b32 t, cnt; the empty asm pins the chain (as inline asm TEXT
        // the instruction drew a 4-cycle s_nop from the hazard recogniser every fourth use)
        cnt[d & 3] = (uint32_t)__builtin_popcount(t[d]) + cnt[d & 3];
        asm volatile("" : "+v"(cnt[d & 3]));
    }
    __builtin_amdgcn_sched_barrier(0);
}
// All 16 rows of two candidates per lane (p0: candidate `lane`, p1: candidate
// `lane + 64`) against the same re-packed current rows, fully unrolled and
// software-pipelined one row ahead: the LDS reads of row k+1 are issued before
// row k is evaluated (two waves per SIMD do not cover the LDS latency on their
// own).  Scheduling fences keep hipcc from hoisting more than that one row
// (which spills).
template <bool TWO, int ROWB>
__device__ __forceinline__ void me16_rows(const uint32_t *s_cur, const uint8_t *p0,
                                          const uint8_t *p1, uint32_t (&acc0)[4],
                                          uint32_t (&acc1)[4]) {
    Me16Q c[2], a[2], b[2];
    me16_ld(c[0], s_cur);
    me16_ld(a[0], p0);
    if (TWO) me16_ld(b[0], p1);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int cur = k & 1, nxt = cur ^ 1;
        if (k + 1 < 16) {
            me16_ld(c[nxt], s_cur + (k + 1) * 16);
            me16_ld(a[nxt], p0 + (k + 1) * ROWB);
            if (TWO) me16_ld(b[nxt], p1 + (k + 1) * ROWB);
        }
        __builtin_amdgcn_sched_barrier(0);
        me16_count13(a[cur], c[cur], acc0);
        if (TWO) me16_count13(b[cur], c[cur], acc1);
    }
}

// Diagnostic build only (-DVCS_ME16_STAMPS): s_memtime at phase boundaries,
// summed per wave into cost_out[wave * 8 + phase] (shares, never quote its run
// time).  No stamp executes in the product build.
#ifdef VCS_ME16_STAMPS
#define ME16_STAMP(i)                                                                        \
    do {                                                                                     \
        unsigned long long t_;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        st_acc[i] += t_ - st_prev;                                                           \
        st_prev = t_;                                                                        \
    } while (0)
#else
#define ME16_STAMP(i) do { } while (0)
#endif

// Everything one block needs from HBM, held in registers until it is used.
struct Me16Regs {
    uint32_t rv[3], cv[3];   // co-located ref / cur block, 3 dwords per lane
    uint4 stg[kMe16Groups];  // search window, 13 x 16 bytes per lane
    uint32_t cq[4];          // current block in 5-pixel-group order, 208 dwords / 64 lanes
    uint32_t sr[2];          // sum(r) of candidates lane and lane + 64 (from the table)
};
// Per-lane byte offsets of those loads relative to wave-uniform bases.  They do
// not depend on the block, so they are computed once per wave; every load is
// then "uniform 64-bit base + 32-bit lane offset" (saddr form): no per-block
// address arithmetic and no 64-bit address registers.
struct Me16Offs {
    uint32_t blk[3];
    uint32_t stg[kMe16Groups];
    uint32_t cq[4];
};
__device__ __forceinline__ void me16_offsets(const Geom &g, int lane, Me16Offs &O) {
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int idx = lane + 64 * t, row = idx / 12, col = idx - row * 12;
        O.blk[t] = (uint32_t)(row * g.pitch + 4 * col);
    }
#pragma unroll
    for (int t = 0; t < kMe16Groups; t++) {
        const int e = lane + 64 * t, row = e / kMe16Groups, gi = e - row * kMe16Groups;
        O.stg[t] = (uint32_t)(row * g.pitch + 15 * gi);
    }
    // current block, dword j of row `row` in 5-pixel-group order starts at byte
    // 15*(j/4) + 4*(j%4) of the row (j = 12: byte 45); unaligned dword loads
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int e = min(lane + 64 * t, 16 * 13 - 1), row = e / 13, j = e - row * 13;
        O.cq[t] = (uint32_t)(row * g.pitch + (j == 12 ? 45 : 15 * (j >> 2) + 4 * (j & 3)));
    }
}

// Issues the loads of work item `w` (= frame * N + block).  No branches (a
// conditional load would make hipcc wait for it on the spot) and no clamping:
// the window is always fetched 64 rows deep; for blocks near the bottom edge
// the extra rows lie in the slack behind the reference frame (the context
// allocates it) and are never used.
__device__ __forceinline__ void me16_issue(const uint8_t *__restrict__ ref,
                                           const uint8_t *__restrict__ curs, size_t cur_stride,
                                           const uint32_t *__restrict__ sums, const Geom &g, int w,
                                           int lane, const Me16Offs &O, Me16Regs &R) {
    constexpr int BS = 16, WIN = 32;
    const int f = w / g.N, blk = w - f * g.N;
    const int byi = blk / g.nbx, bxi = blk - byi * g.nbx;
    const int x = bxi * BS, y = byi * BS;
    const size_t blk_off = (size_t)y * g.pitch + (size_t)(48 * bxi);
    const int i_min = max(y - WIN, 0), j_min = max(x - WIN, 0);
    const uint8_t *__restrict__ wsrc = ref + (size_t)i_min * g.pitch + (size_t)(3 * j_min);
    const uint8_t *__restrict__ cblk = curs + (size_t)f * cur_stride + blk_off;  // uniform
    const uint8_t *__restrict__ rblk = ref + blk_off;                             // uniform
#pragma unroll
    for (int t = 0; t < 3; t++) {
        R.rv[t] = *(const uint32_t *)(rblk + O.blk[t]);
        R.cv[t] = *(const uint32_t *)(cblk + O.blk[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        uint32_t v;
        __builtin_memcpy(&v, cblk + O.cq[t], 4);
        R.cq[t] = v;
    }
    const uint32_t *__restrict__ sb = sums + (size_t)blk * kMe16SumStride;  // uniform
    R.sr[0] = sb[lane];
    R.sr[1] = sb[lane + 64];
#pragma unroll
    for (int t = 0; t < kMe16Groups; t++) R.stg[t] = load16_any_align(wsrc + O.stg[t]);
}

// Motion search of `total` work items (frame-major blocks); sums = the reference's
// sum(r) table (k_me16_sums).
// SLOTS: 16-byte slots per LDS window row; 18 is the conflict-free pitch (13 and 14 were
// measured: more waves per CU, but their 2-way bank conflicts cost more than that gains).
template <int SLOTS>
__global__ __launch_bounds__(64, 2) void k_me16(
    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ curs, size_t cur_stride,
    Geom g, int total, uint32_t *__restrict__ sums, int16_t *__restrict__ mv,
    uint32_t *__restrict__ n_static, unsigned long long *__restrict__ cost_out) {
    constexpr int kRowB = SLOTS * 16;
    // one array: [window 64 rows x kRowB][current block 16 rows x 64 B]
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[kMe16Rows * kRowB + 16 * 64];
    uint8_t *const s_win = s_lds;
    uint32_t *const s_cur = (uint32_t *)(s_lds + kMe16Rows * kRowB);
    constexpr int BS = 16, WIN = 32, STEP = 5;
    const int lane = threadIdx.x;

    // ---- work list of this wave: XCD-contiguous chunks, strided inside -----------
    const int nx = 8;  // workgroups are dealt round-robin over the 8 XCDs (speed only)
    const int xcd = blockIdx.x % nx, wi = blockIdx.x / nx, nw = gridDim.x / nx;
    const int chunk = (total + nx - 1) / nx;
    const int w_end = min((xcd + 1) * chunk, total);
    int w = xcd * chunk + wi;
    if (w >= w_end) return;

    Me16Offs O;
    me16_offsets(g, lane, O);
    Me16Regs A;
    me16_issue(ref, curs, cur_stride, sums, g, w, lane, O, A);
#ifdef VCS_ME16_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif

    for (;;) {
        // ---- prefetch the next block of this wave while this one is searched ------
        const int wn = w + nw;
        Me16Regs B;
        me16_issue(ref, curs, cur_stride, sums, g, min(wn, w_end - 1), lane, O, B);
        ME16_STAMP(0);  // issue of the next block's loads

        const int f = w / g.N, blk = w - f * g.N;
        const int byi = blk / g.nbx, bxi = blk - byi * g.nbx;
        const int x = bxi * BS, y = byi * BS;
        const size_t oidx = (size_t)w;

        // ---- static test (motion.py:109-116): sum(max(r - c, 0)) -------------------
        //   max(r - c, 0) = (|r - c| + r - c) / 2 per byte -> three v_sad_u8 per dword
        bool is_static = false;
        uint32_t s_c = 0;
        uint32_t s_d = 0;  // |r - c| + r - c = 2 * max(r - c, 0) per byte: never negative
#pragma unroll
        for (int t = 0; t < 3; t++) {
            s_d = __builtin_amdgcn_sad_u8(A.rv[t], A.cv[t], s_d);
            s_d = __builtin_amdgcn_sad_u8(A.rv[t], 0u, s_d);
            s_c = __builtin_amdgcn_sad_u8(A.cv[t], 0u, s_c);
        }
        // per lane: s_d holds sad(r,c) + sum(r); subtracting this lane's sum(c) keeps
        // it >= 0, so one reduction gives 2 * sum(max(r - c, 0)) and one gives sum(c)
        s_d = wave_sum_u32(s_d - s_c);
        s_c = wave_sum_u32(s_c);  // sum of the current block's bytes
        const uint32_t sdiff = s_d >> 1;
        is_static = !g.never_static && sdiff <= g.thresh;
        if (is_static && lane == 0) {
            mv[2 * oidx + 0] = 0;
            mv[2 * oidx + 1] = 0;
            atomicAdd(&n_static[f], 1u);
#ifndef VCS_ME16_STAMPS
            if (cost_out) cost_out[oidx] = sdiff;
#endif
        }
        ME16_STAMP(1);  // static test (waits for this block's co-located loads)
        if (!is_static) {
            // ---- park the window in 5-pixel groups (motion.py:125-129) -------------
#pragma unroll
            for (int t = 0; t < kMe16Groups; t++) {
                const int e = lane + 64 * t, row = e / kMe16Groups, gi = e - row * kMe16Groups;
                uint4 v = A.stg[t];
                v.w &= 0x00FFFFFFu;  // pad byte = 0
                *(uint4 *)(s_win + row * kRowB + gi * 16) = v;
            }
            // ... and the negated current block next to it (pad byte -> 0xFF)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int e = lane + 64 * t, row = e / 13, j = e - row * 13;
                const uint32_t padm = ((j & 3) == 3 || j == 12) ? 0xFF000000u : 0u;
                if (e < 16 * 13) s_cur[row * 16 + j] = ~A.cq[t] | padm;
            }
            wave_lds_fence();
            ME16_STAMP(2);  // window + current block parked (waits for this block's window loads)

            // ---- candidate grid (motion.py:132-140) ---------------------------------
            const int i_min = max(y - WIN, 0), j_min = max(x - WIN, 0);
            const int i_max = min(y + WIN, g.H), j_max = min(x + WIN, g.W);
            const int ri = i_max - BS - i_min, rj = j_max - BS - j_min;
            const int nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
            const int ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
            const int ncand = nci * ncj;  // <= 100
            const int cand0 = lane, cand1 = lane + 64;
            const bool act0 = cand0 < ncand, act1 = cand1 < ncand;
            int ci0 = 0, cj0 = 0, ci1 = 0, cj1 = 0;
            if (act0) { ci0 = cand0 / ncj; cj0 = cand0 - ci0 * ncj; }
            if (act1) { ci1 = cand1 / ncj; cj1 = cand1 - ci1 * ncj; }
            const uint8_t *p0 = s_win + (STEP * ci0) * kRowB + cj0 * 16;
            const uint8_t *p1 = s_win + (STEP * ci1) * kRowB + cj1 * 16;

            uint32_t acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
            if (ncand > 64)  // wave-uniform
                me16_rows<true, kRowB>(s_cur, p0, p1, acc0, acc1);
            else
                me16_rows<false, kRowB>(s_cur, p0, p1, acc0, acc1);

            ME16_STAMP(3);  // the row loop
            // cost = sum(r) - sum(c) + 256 * #{r < c}; 64 pad bytes always counted
            const uint32_t n_lt0 = 768u + 64u - (acc0[0] + acc0[1] + acc0[2] + acc0[3]);
            const uint32_t n_lt1 = 768u + 64u - (acc1[0] + acc1[1] + acc1[2] + acc1[3]);
            const uint32_t cost0 = A.sr[0] + 256u * n_lt0 - s_c;
            const uint32_t cost1 = A.sr[1] + 256u * n_lt1 - s_c;
            uint32_t key = 0xFFFFFFFFu;  // (cost << 7) | raster index: motion.py:149
            if (act0) key = (cost0 << 7) | (uint32_t)cand0;
            if (act1) key = min(key, (cost1 << 7) | (uint32_t)cand1);
            key = wave_min_u32(key);
            if (lane == 0) {
                int bx = 0, by = 0;                    // motion.py:102
                unsigned long long c = 9999999999ull;  // motion.py:120
                if (key != 0xFFFFFFFFu) {
                    const int cb = (int)(key & 127u);
                    by = i_min + (cb / ncj) * STEP;
                    bx = j_min + (cb % ncj) * STEP;
                    c = key >> 7;
                }
                mv[2 * oidx + 0] = (int16_t)(bx - x);
                mv[2 * oidx + 1] = (int16_t)(by - y);
#ifndef VCS_ME16_STAMPS
                if (cost_out) cost_out[oidx] = c;
#endif
            }
            wave_lds_fence();  // the window is rewritten by the next block
            ME16_STAMP(4);  // reduction + result store
        }
        if (wn >= w_end) break;  // wave-uniform; every wave leaves after its share
        w = wn;
        A = B;
        ME16_STAMP(5);  // register hand-over
    }
#ifdef VCS_ME16_STAMPS
    if (lane == 0 && cost_out)
        for (int i = 0; i < 8; i++) cost_out[(size_t)blockIdx.x * 8 + i] = st_acc[i];
#endif
}

// sum(r) table of one reference frame: sums[blk * 128 + cand] = byte sum of the
// 16 x 16 block at candidate `cand` of block `blk` (what k_me16 / k_me16r add to
// 256 * #{r < c} - sum(c)).  Separable instead of 100 block sums of 208 dwords:
//   1. byte sum of every 5-pixel group of the window (64 rows x 13 groups) and of
//      its first pixel -- four / one v_sad_u8 on the registers the loads return;
//   2. candidate row sums S[row][cj] = three whole groups + the first pixel of the
//      fourth (16 pixels from group cj on), 640 values;
//   3. candidate sums = 16 consecutive S rows.
// One wave per block, one block per workgroup (the latency of the 13 loads is
// covered by occupancy, 9 KB of LDS per wave).  16 us for a 1080p frame against
// 26 us for the row-loop form (k_me16 in sum mode, removed), once per GOP.
__global__ __launch_bounds__(64) void k_me16_sums(const uint8_t *__restrict__ ref, Geom g,
                                                  uint32_t *__restrict__ sums) {
    constexpr int BS = 16, WIN = 32, STEP = 5, NG = kMe16Groups;
    __shared__ uint32_t s_g[kMe16Rows * NG + 4];  // whole-group sums (+ slack for cj + 3 = 13)
    __shared__ uint32_t s_p[kMe16Rows * NG + 4];  // first-pixel sums
    __shared__ uint32_t s_S[kMe16Rows * 10];
    // workgroups go round-robin to the 8 XCDs: give each XCD a contiguous run of blocks so that
    // the 75 %-overlapping windows of neighbouring blocks meet in one L2 (grid = 8 * chunk)
    const int lane = threadIdx.x, chunk = gridDim.x >> 3;
    const int blk = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if (blk >= g.N) return;
    const int byi = blk / g.nbx, bxi = blk - byi * g.nbx;
    const int x = bxi * BS, y = byi * BS;
    const int i_min = max(y - WIN, 0), j_min = max(x - WIN, 0);
    const uint8_t *__restrict__ wsrc = ref + (size_t)i_min * g.pitch + (size_t)(3 * j_min);
    uint4 stg[NG];
#pragma unroll
    for (int t = 0; t < NG; t++) {
        const int e = lane + 64 * t, row = e / NG, gi = e - row * NG;
        stg[t] = load16_any_align(wsrc + (size_t)row * g.pitch + 15 * gi);
    }
#pragma unroll
    for (int t = 0; t < NG; t++) {
        const uint4 v = stg[t];
        uint32_t a = __builtin_amdgcn_sad_u8(v.x, 0u, 0u);
        a = __builtin_amdgcn_sad_u8(v.y, 0u, a);
        a = __builtin_amdgcn_sad_u8(v.z, 0u, a);
        a = __builtin_amdgcn_sad_u8(v.w & 0x00FFFFFFu, 0u, a);  // 15 bytes, not the pad
        s_g[lane + 64 * t] = a;
        s_p[lane + 64 * t] = __builtin_amdgcn_sad_u8(v.x & 0x00FFFFFFu, 0u, 0u);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 10; q++) {
        const int idx = lane + 64 * q, row = idx / 10, cj = idx - row * 10;
        const uint32_t *gr = s_g + row * NG + cj;
        s_S[idx] = gr[0] + gr[1] + gr[2] + s_p[row * NG + cj + 3];
    }
    __syncthreads();
    // candidate grid exactly as in k_me16 (motion.py:132-140)
    const int i_max = min(y + WIN, g.H), j_max = min(x + WIN, g.W);
    const int ri = i_max - BS - i_min, rj = j_max - BS - j_min;
    const int nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
    const int ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
    const int ncand = nci * ncj;
    uint32_t *__restrict__ sb = sums + (size_t)blk * kMe16SumStride;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int cand = lane + 64 * h;
        uint32_t acc = 0;
        if (cand < ncand) {
            const int ci = cand / ncj, cj = cand - ci * ncj;
            const uint32_t *col = s_S + (STEP * ci) * 10 + cj;
#pragma unroll
            for (int k = 0; k < BS; k++) acc += col[k * 10];
        }
        sb[cand] = acc;
    }
}

// The fast kernel covers the reference's own parameters for block_size 16 on
// frames whose rows are dword-aligned; anything else takes k_me_generic.
inline bool me_fast_supported(const Geom &g) {
    return g.bs == 16 && g.win == 32 && g.step == 5 && (g.W % 4) == 0 && g.N > 0;
}
// Bytes of slack the reference frame needs behind it (window rows fetched
// without clamping at the bottom edge) and of the per-reference sum(r) table.
inline size_t me_fast_ref_slack(const Geom &g) { return (size_t)(kMe16Rows + 2) * g.pitch; }
inline size_t me_fast_sums_bytes(const Geom &g) {
    return (size_t)(g.N > 0 ? g.N : 1) * kMe16SumStride * sizeof(uint32_t);
}

inline int me16_grid(int cus, long long total) {
    static const int per_cu = [] {
        const char *e = getenv("VCS_ME16_WAVES_PER_CU");  // tuning knob
        int v = e ? atoi(e) : 8;
        return v >= 1 && v <= 32 ? v : 8;
    }();
    long long grid = (long long)cus * per_cu;  // resident waves: LDS admits 8 per CU
    grid = (grid + 7) / 8 * 8;                 // a multiple of the XCD count
    const long long cap = (total + 7) / 8 * 8;
    return (int)(grid > cap ? cap : grid);
}

// sum(r) table of `ref` (once per reference frame).
inline int launch_me_fast_sums(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                               uint32_t *sums) {
    hipLaunchKernelGGL(k_me16_sums, dim3(8 * ((g.N + 7) / 8)), dim3(64), 0, stream, ref, g, sums);
    return 0;
}

// n frames of g.N blocks each; `cus` = compute units of the device.  mv / cost
// are indexed frame-major, i.e. by work item.  `sums` must hold ref's table.
inline int launch_me_fast(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                          const uint8_t *curs, size_t stride, int n, uint32_t *sums, int16_t *mv,
                          uint32_t *ns, unsigned long long *cost, hipEvent_t ev_start = nullptr,
                          hipEvent_t ev_stop = nullptr) {
    const long long total = (long long)g.N * n;
    if (total > 0x7FFFFFFF) return -1;
    dim3 gd(me16_grid(cus, total)), block(64);
#ifdef VCS_ME16_STAMPS
    static unsigned long long *dbg = nullptr;
    if (!dbg) hipMalloc(&dbg, sizeof(unsigned long long) * 8 * 65536);
    hipMemsetAsync(dbg, 0, sizeof(unsigned long long) * 8 * gd.x, stream);
    cost = dbg;
#endif
    hipExtLaunchKernelGGL((k_me16<18>), gd, block, 0, stream, ev_start, ev_stop, 0, ref, curs, stride, g,
                          (int)total, sums, mv, ns, cost);
#ifdef VCS_ME16_STAMPS
    {
        static int calls = 0;
        if (++calls == 3) {  // a warm launch
            hipStreamSynchronize(stream);
            std::vector<unsigned long long> h(8 * gd.x);
            hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
            double acc[8] = {0};
            for (unsigned w = 0; w < gd.x; w++)
                for (int i = 0; i < 8; i++) acc[i] += (double)h[w * 8 + i];
            double tot = 0;
            for (int i = 0; i < 6; i++) tot += acc[i];
            const char *nm[6] = {"issue next loads", "static test", "park window", "row loop",
                                 "reduce+store", "reg hand-over"};
            const double blocks = (double)total;
            for (int i = 0; i < 6; i++)
                fprintf(stderr, "[me16 stamps] %-18s %6.1f %%  %8.0f ticks/block\n", nm[i],
                        100.0 * acc[i] / tot, acc[i] / blocks);
            fprintf(stderr, "[me16 stamps] total %.0f ticks/block (s_memtime, 100 MHz?)\n", tot / blocks);
        }
    }
#endif
    return 0;
}

}  // namespace vcs
