// me_fast8.hpp -- k_me8: motion search at the reference's OWN default operating
// point (main.py:15 BLOCK_SIZE = 8): 8x8 blocks, +-16 px window, stride-3 grid
// (motion.py:18,132 for block_size = 8).  Same results as k_me_generic, bit for
// bit; same ideas as k_me16 (me_fast.hpp), re-dimensioned:
//   * 3-pixel groups: candidates step 3 px = 9 bytes; the window is parked in
//     LDS with each 9-byte group padded to a 16-byte slot, anchored at j_min, so
//     candidate cj starts at slot cj: three aligned ds_read_b128 per row, of
//     which 8 dwords carry the 24 bytes of the row.
//   * cost = sum(r) - sum(c) + 256 * #{r < c}: #{r >= c} from bit 7 of
//     v_lerp_u8(r, ~c, 1), pad / foreign bytes neutralised by ~c = 0xFF;
//     sum(r) from a per-reference table (k_me8<kMe8Sums>).
//   * lane = candidate: an interior block has exactly 8 x 8 = 64 candidates,
//     one full wave, one pass.  LDS row pitch 24 slots: slot = 24*(3ci + k) + cj
//     == 8ci + cj + 8k (mod 16) = lane + 8k: conflict-free ds_read_b128.
//   * persistent waves, register prefetch of the next block's loads.
#pragma once
#include "me_fast.hpp"

namespace vcs {

enum { kMe8Search = 0, kMe8Sums = 1 };  // k_me8<MODE>: motion search / the reference's sum(r) table

constexpr int kMe8Rows = 32;     // 2 * win
constexpr int kMe8Groups = 11;   // ceil(32 px / 3)
#ifndef VCS_ME8_SLOTS
#define VCS_ME8_SLOTS 24
#endif
constexpr int kMe8Slots = VCS_ME8_SLOTS;    // 16-byte slots per LDS window row (11 hold pixels; the rest spreads rows over banks)
constexpr int kMe8RowB = kMe8Slots * 16;
constexpr int kMe8StgN = (kMe8Rows * kMe8Groups + 63) / 64;  // 6 loads per lane
constexpr int kMe8SumStride = 64;

struct Me8Regs {
    uint32_t rv, cv;          // co-located ref / cur block: 48 dwords, lanes 0..47
    uint4 stg[kMe8StgN];      // search window groups
    uint32_t cq;              // re-packed current block: 64 dwords, one per lane
    uint32_t sr;              // sum(r) of candidate `lane`
};
struct Me8Offs {
    uint32_t blk, cq;
    uint32_t stg[kMe8StgN];
};

__device__ __forceinline__ void me8_offsets(const Geom &g, int lane, Me8Offs &O) {
    const int e0 = min(lane, 47), row0 = e0 / 6, col0 = e0 - row0 * 6;
    O.blk = (uint32_t)(row0 * g.pitch + 4 * col0);
#pragma unroll
    for (int t = 0; t < kMe8StgN; t++) {
        const int e = min(lane + 64 * t, kMe8Rows * kMe8Groups - 1);
        const int row = e / kMe8Groups, gi = e - row * kMe8Groups;
        O.stg[t] = (uint32_t)(row * g.pitch + 9 * gi);
    }
    // dword q of row `row` of the re-packed block: slot s = q / 3 (q < 6) or 2, dword d
    const int row = lane >> 3, q = lane & 7;
    const int sl = q < 3 ? 0 : q < 6 ? 1 : 2, d = q - 3 * sl;
    O.cq = (uint32_t)(row * g.pitch + 9 * sl + 4 * d);
}

template <int MODE>
__device__ __forceinline__ void me8_issue(const uint8_t *__restrict__ ref,
                                          const uint8_t *__restrict__ curs, size_t cur_stride,
                                          const uint32_t *__restrict__ sums, const Geom &g, int w,
                                          int lane, const Me8Offs &O, Me8Regs &R) {
    constexpr int BS = 8, WIN = 16;
    const int f = w / g.N, blk = w - f * g.N;
    const int byi = blk / g.nbx, bxi = blk - byi * g.nbx;
    const int x = bxi * BS, y = byi * BS;
    const size_t blk_off = (size_t)y * g.pitch + (size_t)(24 * bxi);
    const int i_min = max(y - WIN, 0), j_min = max(x - WIN, 0);
    const uint8_t *__restrict__ wsrc = ref + (size_t)i_min * g.pitch + (size_t)(3 * j_min);
    if (MODE == kMe8Search) {
        const uint8_t *__restrict__ cblk = curs + (size_t)f * cur_stride + blk_off;
        const uint8_t *__restrict__ rblk = ref + blk_off;
        R.rv = *(const uint32_t *)(rblk + O.blk);
        R.cv = *(const uint32_t *)(cblk + O.blk);
        uint32_t v;
        __builtin_memcpy(&v, cblk + O.cq, 4);
        R.cq = v;
        R.sr = sums[(size_t)blk * kMe8SumStride + lane];
    }
#pragma unroll
    for (int t = 0; t < kMe8StgN; t++) R.stg[t] = load16_any_align(wsrc + O.stg[t]);
}

// 8 dwords of one candidate row (slots cj, cj+1, cj+2) against the current row.
__device__ __forceinline__ uint32_t me8_count8(const uint4 &a0, const uint4 &a1, const uint4 &a2,
                                               const uint4 &c0, const uint4 &c1, const uint4 &c2) {
    const uint32_t r[8] = {a0.x, a0.y, a0.z, a1.x, a1.y, a1.z, a2.x, a2.y};
    const uint32_t nc[8] = {c0.x, c0.y, c0.z, c1.x, c1.y, c1.z, c2.x, c2.y};
    asm volatile("" ::"v"(a0.w), "v"(a1.w), "v"(a2.z), "v"(a2.w));  // keep whole ds_read_b128s
    uint32_t t[8], cnt[2] = {0, 0};
#pragma unroll
    for (int d = 0; d < 8; d++) t[d] = __builtin_amdgcn_lerp(r[d], nc[d], 0x01010101u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < 8; d++) t[d] &= 0x80808080u;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < 8; d++) {
        cnt[d & 1] = (uint32_t)__builtin_popcount(t[d]) + cnt[d & 1];  // v_bcnt_u32_b32 t, cnt
        asm volatile("" : "+v"(cnt[d & 1]));
    }
    __builtin_amdgcn_sched_barrier(0);
    return cnt[0] + cnt[1];
}
__device__ __forceinline__ uint32_t me8_sum8(const uint4 &a0, const uint4 &a1, const uint4 &a2) {
    asm volatile("" ::"v"(a0.w), "v"(a1.w), "v"(a2.z), "v"(a2.w));
    const uint32_t r[8] = {a0.x, a0.y, a0.z & 0xFFu, a1.x, a1.y, a1.z & 0xFFu, a2.x, a2.y & 0xFFFFu};
    uint32_t s = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) s = __builtin_amdgcn_sad_u8(r[d], 0u, s);
    return s;
}

template <int MODE>
__global__ __launch_bounds__(64, 3) void k_me8(
    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ curs, size_t cur_stride,
    Geom g, int total, uint32_t *__restrict__ sums, int16_t *__restrict__ mv,
    uint32_t *__restrict__ n_static, unsigned long long *__restrict__ cost_out) {
    // [window 32 rows x 384 B][current block 8 rows x 48 B]
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[kMe8Rows * kMe8RowB + 8 * 48];
    uint8_t *const s_win = s_lds;
    uint32_t *const s_cur = (uint32_t *)(s_lds + kMe8Rows * kMe8RowB);
    constexpr int BS = 8, WIN = 16, STEP = 3;
    const int lane = threadIdx.x;

    const int nx = 8;
    const int xcd = blockIdx.x % nx, wi = blockIdx.x / nx, nw = gridDim.x / nx;
    const int chunk = (total + nx - 1) / nx;
    const int w_end = min((xcd + 1) * chunk, total);
    int w = xcd * chunk + wi;
    if (w >= w_end) return;

    Me8Offs O;
    me8_offsets(g, lane, O);
    Me8Regs A;
    me8_issue<MODE>(ref, curs, cur_stride, sums, g, w, lane, O, A);

    for (;;) {
        const int wn = w + nw;
        Me8Regs B;
        me8_issue<MODE>(ref, curs, cur_stride, sums, g, min(wn, w_end - 1), lane, O, B);

        const int f = w / g.N, blk = w - f * g.N;
        const int byi = blk / g.nbx, bxi = blk - byi * g.nbx;
        const int x = bxi * BS, y = byi * BS;
        const size_t oidx = (size_t)w;

        // ---- static test (motion.py:109-116) over the 48 dwords of the block ------
        bool is_static = false;
        uint32_t s_c = 0;
        if (MODE == kMe8Search) {
            uint32_t s_d = 0;
            if (lane < 48) {
                s_d = __builtin_amdgcn_sad_u8(A.rv, A.cv, 0u);
                s_d = __builtin_amdgcn_sad_u8(A.rv, 0u, s_d);
                s_c = __builtin_amdgcn_sad_u8(A.cv, 0u, 0u);
            }
            s_d = wave_sum_u32(s_d - s_c);  // 2 * sum(max(r - c, 0)), see k_me16
            s_c = wave_sum_u32(s_c);
            const uint32_t sdiff = s_d >> 1;
            is_static = !g.never_static && sdiff <= g.thresh;
            if (is_static && lane == 0) {
                mv[2 * oidx + 0] = 0;
                mv[2 * oidx + 1] = 0;
                atomicAdd(&n_static[f], 1u);
                if (cost_out) cost_out[oidx] = sdiff;
            }
        }
        if (!is_static) {
            // ---- park the window in 3-pixel groups (motion.py:125-129) -------------
#pragma unroll
            for (int t = 0; t < kMe8StgN; t++) {
                const int e = lane + 64 * t, row = e / kMe8Groups, gi = e - row * kMe8Groups;
                uint4 v = A.stg[t];
                v.z &= 0xFFu;  // bytes 9..15 of the slot: pad
                v.w = 0u;
                if (e < kMe8Rows * kMe8Groups) *(uint4 *)(s_win + row * kMe8RowB + gi * 16) = v;
            }
            if (MODE == kMe8Search) {  // negated current block, pad / foreign bytes -> 0xFF
                const int row = lane >> 3, q = lane & 7;
                const int sl = q < 3 ? 0 : q < 6 ? 1 : 2, d = q - 3 * sl;
                const uint32_t padm = d == 2 ? 0xFFFFFF00u : (sl == 2 && d == 1) ? 0xFFFF0000u : 0u;
                s_cur[row * 12 + sl * 4 + d] = ~A.cq | padm;
            }
            wave_lds_fence();

            // ---- candidate grid (motion.py:132-140) ---------------------------------
            const int i_min = max(y - WIN, 0), j_min = max(x - WIN, 0);
            const int i_max = min(y + WIN, g.H), j_max = min(x + WIN, g.W);
            const int ri = i_max - BS - i_min, rj = j_max - BS - j_min;
            const int nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
            const int ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
            const int ncand = nci * ncj;  // <= 64
            const bool act = lane < ncand;
            int ci = 0, cj = 0;
            if (act) { ci = lane / ncj; cj = lane - ci * ncj; }
            const uint8_t *p = s_win + (STEP * ci) * kMe8RowB + cj * 16;

            uint32_t acc = 0;
            uint4 a[2][3], c[2][3];
#pragma unroll
            for (int m = 0; m < 3; m++) {
                a[0][m] = ((const uint4 *)p)[m];
                if (MODE == kMe8Search) c[0][m] = ((const uint4 *)s_cur)[m];
            }
#pragma unroll
            for (int k = 0; k < BS; k++) {
                const int cu = k & 1, nx2 = cu ^ 1;
                if (k + 1 < BS) {
#pragma unroll
                    for (int m = 0; m < 3; m++) {
                        a[nx2][m] = ((const uint4 *)(p + (k + 1) * kMe8RowB))[m];
                        if (MODE == kMe8Search) c[nx2][m] = ((const uint4 *)(s_cur + (k + 1) * 12))[m];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == kMe8Search)
                    acc += me8_count8(a[cu][0], a[cu][1], a[cu][2], c[cu][0], c[cu][1], c[cu][2]);
                else
                    acc += me8_sum8(a[cu][0], a[cu][1], a[cu][2]);
            }

            if (MODE == kMe8Sums) {
                sums[(size_t)blk * kMe8SumStride + lane] = act ? acc : 0u;
            } else {
                // cost = sum(r) - sum(c) + 256 * #{r < c}; 64 pad bytes always counted
                const uint32_t n_lt = 192u + 64u - acc;
                const uint32_t cost = A.sr + 256u * n_lt - s_c;
                uint32_t key = act ? ((cost << 7) | (uint32_t)lane) : 0xFFFFFFFFu;  // motion.py:149
                key = wave_min_u32(key);
                if (lane == 0) {
                    int bx = 0, by = 0;                    // motion.py:102
                    unsigned long long cc = 9999999999ull; // motion.py:120
                    if (key != 0xFFFFFFFFu) {
                        const int cb = (int)(key & 127u);
                        by = i_min + (cb / ncj) * STEP;
                        bx = j_min + (cb % ncj) * STEP;
                        cc = key >> 7;
                    }
                    mv[2 * oidx + 0] = (int16_t)(bx - x);
                    mv[2 * oidx + 1] = (int16_t)(by - y);
                    if (cost_out) cost_out[oidx] = cc;
                }
            }
            wave_lds_fence();  // the window is rewritten by the next block
        }
        if (wn >= w_end) break;
        w = wn;
        A = B;
    }
}

inline bool me_fast8_supported(const Geom &g) {
    return g.bs == 8 && g.win == 16 && g.step == 3 && (g.W % 4) == 0 && g.N > 0;
}
inline size_t me_fast8_sums_bytes(const Geom &g) {
    return (size_t)(g.N > 0 ? g.N : 1) * kMe8SumStride * sizeof(uint32_t);
}
inline int me8_grid(int cus, long long total) {
    long long grid = (long long)cus * 12;  // LDS (12.4 KB) and VGPRs admit 12 waves per CU
    grid = (grid + 7) / 8 * 8;
    const long long cap = (total + 7) / 8 * 8;
    return (int)(grid > cap ? cap : grid);
}
inline int launch_me_fast8_sums(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                                uint32_t *sums) {
    hipLaunchKernelGGL((k_me8<kMe8Sums>), dim3(me8_grid(cus, g.N)), dim3(64), 0, stream, ref, ref,
                       (size_t)0, g, g.N, sums, (int16_t *)nullptr, (uint32_t *)nullptr,
                       (unsigned long long *)nullptr);
    return 0;
}
inline int launch_me_fast8(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                           const uint8_t *curs, size_t stride, int n, uint32_t *sums, int16_t *mv,
                           uint32_t *ns, unsigned long long *cost) {
    const long long total = (long long)g.N * n;
    if (total > 0x7FFFFFFF) return -1;
    hipLaunchKernelGGL((k_me8<kMe8Search>), dim3(me8_grid(cus, total)), dim3(64), 0, stream, ref,
                       curs, stride, g, (int)total, sums, mv, ns, cost);
    return 0;
}

}  // namespace vcs
