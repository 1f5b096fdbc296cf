// me_ext.hpp -- k_me_ext: the motion searches that are NOT in the reference.
//
// The reference has one search (the stride-round(bs/3) grid of motion.py:132-140) and one
// cost (the uint8 wrap-around sum of motion.py:146).  BASELINE config 3 is worded "+-32px
// three-step search"; this kernel adds, behind vcs_set_search (default: off),
//   search = TSS   the classic three-step (logarithmic) search: from the co-located position,
//                  steps S = S0, S0/2, ..., 1 with S0 the largest power of two such that
//                  2*S0 - 1 <= win (16, 8, 4, 2, 1 for +-32); every stage evaluates the 8
//                  neighbours at distance S of the current centre that keep the block inside the
//                  frame and moves to the best one if it is STRICTLY cheaper than the centre
//                  (ties between neighbours: raster order dy, dx);
//   cost   = SAD   sum |r - c| (v_sad_u8) instead of sum (r - c) mod 256.
// The static early-out (motion.py:109-116) is unchanged.  No reference output exists for any
// of this: the oracle is oracle/vcs_oracle.c: vcso_me_ext, a plain restatement of the
// definition above ("parity unpinned").
//
// One wavefront per motion block; window and current block parked in LDS as in k_me_generic.
//   grid search : lane = candidate (k_me_generic's loop, cost selectable);
//   TSS         : lane = (neighbour k = lane / 8, row class p = lane % 8): 8 lanes share one
//                 neighbour's rows (r = p, p + 8, ...), partial costs meet by three xor-shuffles;
//                 the centre's cost is known from the stage before.
#pragma once
#include "vcs_common.hpp"

namespace vcs {

enum { kSearchGrid = 0, kSearchTss = 1 };
enum { kCostWrap = 0, kCostSad = 1 };

template <int COST>
__device__ __forceinline__ uint32_t ext_cost_dword(uint32_t rv, uint32_t cv, uint32_t acc) {
    if (COST == kCostSad) return __builtin_amdgcn_sad_u8(rv, cv, acc);
    return bytesum_acc(sub_u8x4(rv, cv), acc);  // motion.py:146
}

// cost of the block at window position (wi, wj) over rows r0, r0 + rstep, ... (< bs)
template <int COST>
__device__ __forceinline__ uint32_t ext_block_cost(const uint8_t *s_win, const uint8_t *s_cur,
                                                   const Geom &g, int bs, int cpitch, int cdw,
                                                   uint32_t tailmask, int i_min, int j_min, int wi,
                                                   int wj, int r0, int rstep) {
    uint32_t cost = 0;
    for (int r = r0; r < bs; r += rstep) {
        const int wr = wi + r;
        const int lead = (int)(((size_t)(i_min + wr) * g.pitch + (size_t)(3 * j_min)) & 3);
        const int A = wr * g.wpitch + lead + 3 * wj;
        const uint32_t *wp = (const uint32_t *)(s_win + (A & ~3));
        const uint32_t *cp = (const uint32_t *)(s_cur + r * cpitch);
        const uint32_t sh = (uint32_t)(A & 3);
        uint32_t lo = wp[0];
        for (int d = 0; d < cdw; d++) {
            const uint32_t hi = wp[d + 1];
            uint32_t rv = __builtin_amdgcn_alignbyte(hi, lo, sh);
            if (d == cdw - 1) rv &= tailmask;
            cost = ext_cost_dword<COST>(rv, cp[d], cost);
            lo = hi;
        }
    }
    return cost;
}

template <int SEARCH, int COST>
__global__ __launch_bounds__(64) void k_me_ext(
    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ curs, size_t cur_stride,
    Geom g, int16_t *__restrict__ mv, uint32_t *__restrict__ n_static,
    unsigned long long *__restrict__ cost_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int bs = g.bs;
    const int lane = threadIdx.x;
    const int blk = blockIdx.x, f = blockIdx.y;
    const uint8_t *__restrict__ cur = curs + (size_t)f * cur_stride;
    const int x = (blk % g.nbx) * bs, y = (blk / g.nbx) * bs;
    const int rowb = 3 * bs;
    const int cpitch = (rowb + 3) & ~3;
    const int cdw = cpitch >> 2;
    uint8_t *s_cur = smem;
    uint8_t *s_win = smem + ((bs * cpitch + 15) & ~15);

    // ---- static test (motion.py:109-116) + park the current block ----------
    uint32_t sdiff = 0;
    for (int e = lane; e < bs * cpitch; e += 64) {
        int r = e / cpitch, b = e - r * cpitch;
        uint8_t c = 0;
        if (b < rowb) {
            size_t off = (size_t)(y + r) * g.pitch + (size_t)(3 * x + b);
            c = cur[off];
            uint8_t rr = ref[off];
            sdiff += rr > c ? (uint32_t)(rr - c) : 0u;
        }
        s_cur[e] = c;
    }
    sdiff = wave_sum_u32(sdiff);
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

    // ---- stage the search window.  Grid: [y - win, y + win) as motion.py:125-129.  TSS:
    // rows / columns up to y + win + bs, so that displacements up to +(win - 1) fit.
    const int ext = SEARCH == kSearchTss ? bs : 0;
    const int i_min = max(y - g.win, 0), j_min = max(x - g.win, 0);
    const int i_max = min(y + g.win + ext, g.H), j_max = min(x + g.win + ext, g.W);
    const int wrows = i_max - i_min, wbytes = 3 * (j_max - j_min);
    const int wdw = g.wpitch >> 2;
    for (int e = lane; e < wrows * wdw; e += 64) {
        int r = e / wdw, d = e - r * wdw;
        size_t a0 = (size_t)(i_min + r) * g.pitch + (size_t)(3 * j_min);
        int lead = (int)(a0 & 3);
        if (4 * d < lead + wbytes)
            *(uint32_t *)(s_win + r * g.wpitch + 4 * d) =
                *(const uint32_t *)(ref + (a0 & ~(size_t)3) + 4 * d);
    }
    __syncthreads();
    const uint32_t tailmask = 0xFFFFFFFFu >> ((8 * (4 - (rowb & 3))) & 31);

    int bx = 0, by = 0;                    // motion.py:102
    unsigned long long c = 9999999999ull;  // motion.py:120
    if (SEARCH == kSearchGrid) {
        // ---- the reference's candidate grid (motion.py:132-140), cost selectable -----
        const int ri = i_max - bs - i_min, rj = j_max - bs - j_min;
        const int nci = ri >= 1 ? (ri - 1) / g.step + 1 : 0;
        const int ncj = rj >= 1 ? (rj - 1) / g.step + 1 : 0;
        const int ncand = nci * ncj;
        unsigned long long best = ~0ull;
        for (int cand = lane; cand < ncand; cand += 64) {
            const int ci = cand / ncj, cj = cand - ci * ncj;
            const uint32_t cost = ext_block_cost<COST>(s_win, s_cur, g, bs, cpitch, cdw, tailmask, i_min,
                                                       j_min, ci * g.step, cj * g.step, 0, 1);
            const unsigned long long key = ((unsigned long long)cost << 32) | (uint32_t)cand;
            best = key < best ? key : best;  // motion.py:149 strict <, raster order
        }
        best = wave_min_u64(best);
        if (best != ~0ull) {
            const int cand = (int)(uint32_t)best;
            by = i_min + (cand / ncj) * g.step;
            bx = j_min + (cand % ncj) * g.step;
            c = best >> 32;
        }
    } else {
        // ---- three-step search around the co-located position ---------------------------
        const int k = lane >> 3, p = lane & 7;       // neighbour, row class
        const int kk = k < 4 ? k : k + 1;            // 0..8 without the centre (4)
        const int ndy = kk / 3 - 1, ndx = kk % 3 - 1;
        int ci = y - i_min, cj = x - j_min;          // centre, window coordinates
        const int max_i = wrows - bs, max_j = (j_max - j_min) - bs;
        // centre cost: lanes 0..7 share its rows
        uint32_t cc = ext_block_cost<COST>(s_win, s_cur, g, bs, cpitch, cdw, tailmask, i_min, j_min, ci, cj, p, 8);
        cc += __shfl_xor(cc, 1, 64); cc += __shfl_xor(cc, 2, 64); cc += __shfl_xor(cc, 4, 64);
        unsigned long long best = (unsigned long long)__builtin_amdgcn_readfirstlane(cc);
        int S = 1;
        while (2 * (2 * S) - 1 <= g.win) S *= 2;     // largest power of two with 2S - 1 <= win
        for (; S >= 1; S >>= 1) {
            const int wi = ci + ndy * S, wj = cj + ndx * S;
            const bool valid = wi >= 0 && wi <= max_i && wj >= 0 && wj <= max_j;
            uint32_t pc = 0;
            if (valid)
                pc = ext_block_cost<COST>(s_win, s_cur, g, bs, cpitch, cdw, tailmask, i_min, j_min, wi, wj, p, 8);
            pc += __shfl_xor(pc, 1, 64); pc += __shfl_xor(pc, 2, 64); pc += __shfl_xor(pc, 4, 64);
            unsigned long long key = valid ? (((unsigned long long)pc << 8) | (unsigned)k) : ~0ull;
            key = wave_min_u64(key);
            if (key != ~0ull && (key >> 8) < best) {  // strictly cheaper than the centre
                const int kb = (int)(key & 7u), kkb = kb < 4 ? kb : kb + 1;
                ci += (kkb / 3 - 1) * S;
                cj += (kkb % 3 - 1) * S;
                best = key >> 8;
            }
        }
        by = i_min + ci;
        bx = j_min + cj;
        c = best;
    }
    if (lane == 0) {
        mv[2 * oidx + 0] = (int16_t)(bx - x);
        mv[2 * oidx + 1] = (int16_t)(by - y);
        if (cost_out) cost_out[oidx] = c;
    }
}

}  // namespace vcs
