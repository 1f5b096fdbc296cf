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
//   * The current block is wave-uniform.  Its 16 rows are re-packed ONCE per
//     block into the same 5-pixel-group format (negated, pad byte 0xFF), lane-
//     parallel, and parked in 1 KB of LDS; the row loop reads them back with
//     wave-uniform (broadcast) ds_read_b128.  The row loop therefore holds no
//     scalar loads (SMEM shares lgkmcnt with LDS and would force full drains)
//     and no re-packing: the vector ALU only sees v_lerp_u8 / v_and / v_bcnt /
//     v_sad_u8.
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
//   * Persistent waves with register prefetch.  The kernel is bound by vector
//     ALU issue (about 2,000 wave-instructions per block), and LDS limits it to
//     two waves per SIMD, so nothing else may stall: each wave walks a strided
//     list of blocks and issues ALL global loads of block i+1 (19 per lane,
//     into registers) before it searches block i.  Work is dealt so that the
//     blocks of one XCD are contiguous in the frame (reference-window reuse in
//     that XCD's L2); the result never depends on the mapping.
#pragma once
#include <stdlib.h>

#include "vcs_common.hpp"

namespace vcs {

constexpr int kMe16Rows = 64;    // 2 * win
constexpr int kMe16Groups = 13;  // ceil(64 px / 5)

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

// 16 dwords of LDS (four ds_read_b128) starting at p.  keep4: force the 4th
// read to stay a conflict-free ds_read_b128 -- only its first dword is used,
// but as a ds_read_b32 the 16-byte lane stride would land 32 lanes on 8 banks.
struct Me16Q {
    uint4 q0, q1, q2, q3;
};
__device__ __forceinline__ void me16_ld(Me16Q &Q, const void *p, bool keep4) {
    const uint4 *a = (const uint4 *)p;
    Q.q0 = a[0]; Q.q1 = a[1]; Q.q2 = a[2]; Q.q3 = a[3];
    if (keep4) asm volatile("" ::"v"(Q.q3.y), "v"(Q.q3.z), "v"(Q.q3.w));
}

// 13 dwords of one candidate row against the re-packed current row.  Two
// accumulator pairs keep the v_bcnt / v_sad_u8 chains short.
__device__ __forceinline__ void me16_cost13(const uint4 &q0, const uint4 &q1, const uint4 &q2,
                                            const uint4 &q3, const uint32_t (&nc)[13],
                                            uint32_t (&cnt)[2], uint32_t (&sum)[2]) {
    const uint32_t r[13] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                            q2.x, q2.y, q2.z, q2.w, q3.x & 0x00FFFFFFu};
#pragma unroll
    for (int d = 0; d < 13; d++) {
        uint32_t t = __builtin_amdgcn_lerp(r[d], nc[d], 0x01010101u) & 0x80808080u;
        // v_bcnt_u32_b32 with its accumulate operand (hipcc splits popcount + add
        // into bcnt(x, 0) and extra v_add3 otherwise)
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[d & 1]) : "v"(t));
        sum[d & 1] = __builtin_amdgcn_sad_u8(r[d], 0u, sum[d & 1]);
    }
}

// All 16 rows, software-pipelined one half-row ahead: while candidate set A of
// row k is evaluated the reads of set B are in flight, and while B is
// evaluated those of row k+1 (current row + set A) are -- with two waves per
// SIMD nothing else would cover the LDS latency.  Live: <= 4 operand groups.
template <bool TWO, int ROWB>
__device__ __forceinline__ void me16_search(const uint32_t *s_cur, const uint8_t *p0,
                                            const uint8_t *p1, uint32_t (&cnt0)[2],
                                            uint32_t (&sum0)[2], uint32_t (&cnt1)[2],
                                            uint32_t (&sum1)[2]) {
    Me16Q c, a, b, cn, an;
    me16_ld(c, s_cur, false);
    me16_ld(a, p0, true);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (TWO) me16_ld(b, p1 + k * ROWB, true);
        if (!TWO && k + 1 < 16) {
            me16_ld(cn, s_cur + (k + 1) * 16, false);
            me16_ld(an, p0 + (k + 1) * ROWB, true);
        }
        const uint32_t nc[13] = {c.q0.x, c.q0.y, c.q0.z, c.q0.w, c.q1.x, c.q1.y, c.q1.z,
                                 c.q1.w, c.q2.x, c.q2.y, c.q2.z, c.q2.w, c.q3.x};
        me16_cost13(a.q0, a.q1, a.q2, a.q3, nc, cnt0, sum0);
        if (TWO) {
            if (k + 1 < 16) {
                me16_ld(cn, s_cur + (k + 1) * 16, false);
                me16_ld(an, p0 + (k + 1) * ROWB, true);
            }
            me16_cost13(b.q0, b.q1, b.q2, b.q3, nc, cnt1, sum1);
        }
        if (k + 1 < 16) {
            c = cn;
            a = an;
        }
    }
}

// Everything one block needs from HBM, held in registers until it is used.
struct Me16Regs {
    uint32_t rv[3], cv[3];   // co-located ref / cur block, 3 dwords per lane
    uint4 stg[kMe16Groups];  // search window, 13 x 16 bytes per lane
    uint32_t cq[4];          // current block in 5-pixel-group order, 208 dwords / 64 lanes
};

// Issues the loads of work item `w` (= frame * N + block).  No branches: a
// conditional load would make hipcc wait for it on the spot.
__device__ __forceinline__ void me16_issue(const uint8_t *__restrict__ ref,
                                           const uint8_t *__restrict__ curs, size_t cur_stride,
                                           const Geom &g, int w, int lane, Me16Regs &R) {
    constexpr int BS = 16, WIN = 32;
    const int f = w / g.N, blk = w - f * g.N;
    const int byi = blk / g.nbx, bxi = blk - byi * g.nbx;
    const int x = bxi * BS, y = byi * BS;
    const uint8_t *__restrict__ cur = curs + (size_t)f * cur_stride;
    const size_t blk_off = (size_t)y * g.pitch + (size_t)(48 * bxi);
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int idx = lane + 64 * t, row = idx / 12, col = idx - row * 12;
        const size_t off = blk_off + (size_t)row * g.pitch + (size_t)(4 * col);
        R.rv[t] = *(const uint32_t *)(ref + off);
        R.cv[t] = *(const uint32_t *)(cur + off);
    }
    const int i_min = max(y - WIN, 0), j_min = max(x - WIN, 0);
    const int wrows = min(y + WIN, g.H) - i_min;
    const uint8_t *__restrict__ wsrc = ref + (size_t)i_min * g.pitch + (size_t)(3 * j_min);
#pragma unroll
    for (int t = 0; t < kMe16Groups; t++) {
        const int e = lane + 64 * t, row = e / kMe16Groups, gi = e - row * kMe16Groups;
        const int rc = min(row, wrows - 1);  // rows past a clipped window: clamped, never read
        R.stg[t] = load16_any_align(wsrc + (size_t)rc * g.pitch + (size_t)(15 * gi));
    }
    // current block, dword j of row `row` in 5-pixel-group order starts at byte
    // 15*(j/4) + 4*(j%4) of the row (j = 12: byte 45); unaligned dword loads
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int e = min(lane + 64 * t, 16 * 13 - 1), row = e / 13, j = e - row * 13;
        const int boff = j == 12 ? 45 : 15 * (j >> 2) + 4 * (j & 3);
        uint32_t v;
        __builtin_memcpy(&v, cur + blk_off + (size_t)row * g.pitch + (size_t)boff, 4);
        R.cq[t] = v;
    }
}

// SLOTS: 16-byte slots per LDS window row (18 = conflict-free, 13/14 = denser).
template <int SLOTS>
__global__ __launch_bounds__(64, 2) void k_me16(
    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ curs, size_t cur_stride,
    Geom g, int total, int16_t *__restrict__ mv, uint32_t *__restrict__ n_static,
    unsigned long long *__restrict__ cost_out) {
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

    Me16Regs A;
    me16_issue(ref, curs, cur_stride, g, w, lane, A);

    for (;;) {
        // ---- prefetch the next block of this wave while this one is searched ------
        const int wn = w + nw;
        Me16Regs B;
        me16_issue(ref, curs, cur_stride, g, min(wn, w_end - 1), lane, B);

        const int f = w / g.N, blk = w - f * g.N;
        const int byi = blk / g.nbx, bxi = blk - byi * g.nbx;
        const int x = bxi * BS, y = byi * BS;
        const size_t oidx = (size_t)w;

        // ---- static test (motion.py:109-116): sum(max(r - c, 0)) -------------------
        //   max(r - c, 0) = (|r - c| + r - c) / 2 per byte -> three v_sad_u8 per dword
        uint32_t s_abs = 0, s_r = 0, s_c = 0;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            s_abs = __builtin_amdgcn_sad_u8(A.rv[t], A.cv[t], s_abs);
            s_r = __builtin_amdgcn_sad_u8(A.rv[t], 0u, s_r);
            s_c = __builtin_amdgcn_sad_u8(A.cv[t], 0u, s_c);
        }
        s_abs = wave_sum_u32(s_abs);
        s_r = wave_sum_u32(s_r);
        s_c = wave_sum_u32(s_c);  // sum of the current block's bytes
        const uint32_t sdiff = (s_abs + s_r - s_c) >> 1;
        if (!g.never_static && sdiff <= g.thresh) {
            if (lane == 0) {
                mv[2 * oidx + 0] = 0;
                mv[2 * oidx + 1] = 0;
                atomicAdd(&n_static[f], 1u);
                if (cost_out) cost_out[oidx] = sdiff;
            }
        } else {
            // ---- park the window in 5-pixel groups (motion.py:125-129) -------------
#pragma unroll
            for (int t = 0; t < kMe16Groups; t++) {
                const int e = lane + 64 * t, row = e / kMe16Groups, gi = e - row * kMe16Groups;
                uint4 v = A.stg[t];
                v.w &= 0x00FFFFFFu;  // pad byte = 0 so that it adds nothing to sum(r)
                *(uint4 *)(s_win + row * kRowB + gi * 16) = v;
            }
            // ... and the negated current block next to it (pad byte -> 0xFF)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int e = lane + 64 * t, row = e / 13, j = e - row * 13;
                const uint32_t padm = ((j & 3) == 3 || j == 12) ? 0xFF000000u : 0u;
                if (e < 16 * 13) s_cur[row * 16 + j] = ~A.cq[t] | padm;
            }
            __syncthreads();

            // ---- candidate grid (motion.py:132-140) ---------------------------------
            const int i_min = max(y - WIN, 0), j_min = max(x - WIN, 0);
            const int i_max = min(y + WIN, g.H), j_max = min(x + WIN, g.W);
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
            const uint8_t *p0 = s_win + (STEP * ci0) * kRowB + cj0 * 16;
            const uint8_t *p1 = s_win + (STEP * ci1) * kRowB + cj1 * 16;

            uint32_t cnt0[2] = {0, 0}, sum0[2] = {0, 0}, cnt1[2] = {0, 0}, sum1[2] = {0, 0};
            if (two)  // wave-uniform
                me16_search<true, kRowB>(s_cur, p0, p1, cnt0, sum0, cnt1, sum1);
            else
                me16_search<false, kRowB>(s_cur, p0, p1, cnt0, sum0, cnt1, sum1);
            const uint32_t cnt0s = cnt0[0] + cnt0[1], cnt1s = cnt1[0] + cnt1[1];
            const uint32_t sum0s = sum0[0] + sum0[1], sum1s = sum1[0] + sum1[1];
            // cost = sum(r) - sum(c) + 256 * #{r < c}; 64 pad bytes were always counted
            const uint32_t n_lt0 = 768u + 64u - cnt0s, n_lt1 = 768u + 64u - cnt1s;
            const uint32_t cost0 = sum0s + 256u * n_lt0 - s_c;
            const uint32_t cost1 = sum1s + 256u * n_lt1 - s_c;
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
            __syncthreads();  // the window is rewritten by the next block
        }
        if (wn >= w_end) break;  // wave-uniform; every wave leaves after <= its share
        w = wn;
        A = B;
    }
}

// The fast kernel covers the reference's own parameters for block_size 16 on
// frames whose rows are dword-aligned; anything else takes k_me_generic.
inline bool me_fast_supported(const Geom &g) {
    return g.bs == 16 && g.win == 32 && g.step == 5 && (g.W % 4) == 0 && g.N > 0;
}

// n frames of g.N blocks each; `cus` = compute units of the device.  mv / cost
// are indexed frame-major, i.e. by work item.
inline int launch_me_fast(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                          const uint8_t *curs, size_t stride, int n, int16_t *mv, uint32_t *ns,
                          unsigned long long *cost) {
    static const int slots = [] {
        const char *e = getenv("VCS_ME16_SLOTS");  // tuning knob: 14 or 18
        int v = e ? atoi(e) : 18;
        return v == 14 ? v : 18;
    }();
    static const int per_cu = [] {
        const char *e = getenv("VCS_ME16_WAVES_PER_CU");  // tuning knob
        int v = e ? atoi(e) : 8;
        return v >= 1 && v <= 32 ? v : 8;
    }();
    const long long total = (long long)g.N * n;
    if (total > 0x7FFFFFFF) return -1;
    int grid = cus * per_cu;            // resident waves: LDS admits 8 (18 slots) per CU
    grid = (grid + 7) / 8 * 8;          // a multiple of the XCD count
    if ((long long)grid > (total + 7) / 8 * 8) grid = (int)((total + 7) / 8 * 8);
    dim3 gd(grid), block(64);
    const int tot = (int)total;
    if (slots == 14)
        hipLaunchKernelGGL(k_me16<14>, gd, block, 0, stream, ref, curs, stride, g, tot, mv, ns, cost);
    else
        hipLaunchKernelGGL(k_me16<18>, gd, block, 0, stream, ref, curs, stride, g, tot, mv, ns, cost);
    return 0;
}

}  // namespace vcs
