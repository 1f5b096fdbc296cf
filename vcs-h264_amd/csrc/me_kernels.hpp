// me_kernels.hpp -- block-matching motion search (motion.py:100-154) on gfx950.
//
// Semantics reproduced exactly:
//   static test   sum(max(ref - cur, 0)) over the co-located block <= thresh
//                 (cv2.subtract saturates, motion.py:111) -> MV (0,0);
//   search        candidates on the stride-`step` grid anchored at
//                 (i_min, j_min), kept while i + bs < i_max (motion.py:133-140);
//   cost          sum((ref - cur) mod 256)  -- uint8 wrap-around, NOT SAD
//                 (motion.py:146);
//   selection     strict <, raster scan order: the winner minimises
//                 (cost, candidate raster index) lexicographically (:149);
//   no candidate  best_coord stays [0,0] -> MV = (-x, -y) (motion.py:102).
#pragma once
#include "vcs_common.hpp"

namespace vcs {

// --------------------------------------------------------------------------
// k_me_generic: any block size / window / stride.  One wavefront per motion
// block; the (<= 2win x 2win) reference window and the current block are
// parked in LDS; each LANE owns one candidate at a time and walks its rows as
// aligned dwords (v_alignbyte to undo the 3-byte-pixel misalignment, SWAR
// byte subtract, v_sad_u8 as a 4-byte horizontal add); the wave then reduces
// (cost << 32 | raster index) with a 64-bit min.
//
// BS_T > 0 fixes the block size at compile time (full unrolling); 0 = runtime.
// LDS: [cur block: bs rows x cpitch][window: (i_max - i_min) rows x wpitch].
// --------------------------------------------------------------------------
template <int BS_T>
__global__ __launch_bounds__(64) void k_me_generic(
    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ curs, size_t cur_stride,
    Geom g, int16_t *__restrict__ mv, uint32_t *__restrict__ n_static,
    unsigned long long *__restrict__ cost_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int bs = BS_T ? BS_T : g.bs;
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

    // ---- stage the search window (motion.py:125-129) -----------------------
    const int i_min = max(y - g.win, 0), j_min = max(x - g.win, 0);
    const int i_max = min(y + g.win, g.H), j_max = min(x + g.win, g.W);
    const int wrows = i_max - i_min, wbytes = 3 * (j_max - j_min);
    const int wdw = g.wpitch >> 2;
    for (int e = lane; e < wrows * wdw; e += 64) {
        int r = e / wdw, d = e - r * wdw;
        size_t a0 = (size_t)(i_min + r) * g.pitch + (size_t)(3 * j_min);
        int lead = (int)(a0 & 3);
        if (4 * d < lead + wbytes)  // aligned dwords covering the row's bytes
            *(uint32_t *)(s_win + r * g.wpitch + 4 * d) =
                *(const uint32_t *)(ref + (a0 & ~(size_t)3) + 4 * d);
    }
    __syncthreads();

    // ---- candidate grid (motion.py:132-140) ---------------------------------
    const int ri = i_max - bs - i_min, rj = j_max - bs - j_min;
    const int nci = ri >= 1 ? (ri - 1) / g.step + 1 : 0;
    const int ncj = rj >= 1 ? (rj - 1) / g.step + 1 : 0;
    const int ncand = nci * ncj;
    const uint32_t tailmask = 0xFFFFFFFFu >> ((8 * (4 - (rowb & 3))) & 31);  // rowb % 4 == 0: all

    constexpr int UR_ROWS = BS_T ? BS_T : 1, UR_DW = BS_T ? (3 * BS_T + 3) / 4 : 1;
    unsigned long long best = ~0ull;
    for (int cand = lane; cand < ncand; cand += 64) {
        const int ci = cand / ncj, cj = cand - ci * ncj;
        uint32_t cost = 0;
#pragma unroll UR_ROWS
        for (int r = 0; r < bs; r++) {
            const int wr = ci * g.step + r;
            const int lead = (int)(((size_t)(i_min + wr) * g.pitch + (size_t)(3 * j_min)) & 3);
            const int A = wr * g.wpitch + lead + 3 * cj * g.step;
            const uint32_t *wp = (const uint32_t *)(s_win + (A & ~3));
            const uint32_t *cp = (const uint32_t *)(s_cur + r * cpitch);
            const uint32_t sh = (uint32_t)(A & 3);
            uint32_t lo = wp[0];
#pragma unroll UR_DW
            for (int d = 0; d < cdw; d++) {
                uint32_t hi = wp[d + 1];
                uint32_t rv = __builtin_amdgcn_alignbyte(hi, lo, sh);
                if (d == cdw - 1) rv &= tailmask;
                cost = bytesum_acc(sub_u8x4(rv, cp[d]), cost);  // motion.py:146
                lo = hi;
            }
        }
        unsigned long long key = ((unsigned long long)cost << 32) | (uint32_t)cand;
        best = key < best ? key : best;  // motion.py:149 strict <, raster order
    }
    best = wave_min_u64(best);

    if (lane == 0) {
        int bx = 0, by = 0;  // motion.py:102
        unsigned long long c = 9999999999ull;  // motion.py:120
        if (best != ~0ull) {
            int cand = (int)(uint32_t)best;
            by = i_min + (cand / ncj) * g.step;
            bx = j_min + (cand % ncj) * g.step;
            c = best >> 32;
        }
        mv[2 * oidx + 0] = (int16_t)(bx - x);  // motion.py:158
        mv[2 * oidx + 1] = (int16_t)(by - y);  // motion.py:159
        if (cost_out) cost_out[oidx] = c;
    }
}

}  // namespace vcs
