// me_roll.hpp -- k_me16r: k_me16 (me_fast.hpp) with a ROLLING search window.
//
// k_me16 re-fetches the whole 64-row window (13.3 KB through the texture
// addresser, 13 ds_write_b128 per lane) for every macroblock, although the
// block below shares 48 of those rows.  Here a wave walks DOWN a column of
// macroblocks, one segment (about 34 blocks) at a time: the window lives in LDS
// as a circular buffer of 64 rows (frame row R -> slot R mod 64, same
// 5-pixel-group row format, same pitch); it is loaded in full once per segment
// and every further block brings in only its 16 new rows.  Vertically adjacent blocks
// share x, hence j_min and the group phase, so the parked rows stay valid.
// A candidate whose 16 rows cross the wrap point switches base once (one
// v_cmp / v_cndmask / v_add per row); the bank-conflict-free slot congruence of
// k_me16 is unaffected (64 rows * 18 slots = 0 mod 16).
// Results are identical to k_me16 / k_me_generic.
#pragma once
#include <hip/hip_ext.h>

#include "me_fast.hpp"

namespace vcs {

constexpr int kMe16NewN = 4;  // ceil(16 rows * 13 groups / 64 lanes)
constexpr int kMe16CurPitch = 20;  // dwords per re-packed current row in LDS (me16r_slabs)
constexpr int kMe16Pass2Max = 36;  // at most 100 - 64 candidates in the second pass

struct Me16rRegs {
    uint32_t rv[3], cv[3];
    uint32_t cq[4];
    uint32_t sr[2];
    uint4 nw[kMe16NewN];  // the next block's 16 new window rows
};
struct Me16rOffs {
    uint32_t blk[3], cq[4];
    uint32_t full[kMe16Groups];  // source offsets of the full 64-row load
    uint32_t nw[kMe16NewN];      // source offsets of the 16 new rows (relative to row B + 64)
    uint32_t nwdst[kMe16NewN];   // LDS offsets of the 16 new rows (relative to slot B mod 64)
};

template <int ROWB>
__device__ __forceinline__ void me16r_offsets(const Geom &g, int lane, Me16rOffs &O) {
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int idx = lane + 64 * t, row = idx / 12, col = idx - row * 12;
        O.blk[t] = (uint32_t)(row * g.pitch + 4 * col);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int e = min(lane + 64 * t, 16 * 13 - 1), row = e / 13, j = e - row * 13;
        O.cq[t] = (uint32_t)(row * g.pitch + (j == 12 ? 45 : 15 * (j >> 2) + 4 * (j & 3)));
    }
#pragma unroll
    for (int t = 0; t < kMe16Groups; t++) {
        const int e = lane + 64 * t, row = e / kMe16Groups, gi = e - row * kMe16Groups;
        O.full[t] = (uint32_t)(row * g.pitch + 15 * gi);
    }
#pragma unroll
    for (int t = 0; t < kMe16NewN; t++) {
        const int e = min(lane + 64 * t, 16 * kMe16Groups - 1);
        const int row = e / kMe16Groups, gi = e - row * kMe16Groups;
        O.nw[t] = (uint32_t)(row * g.pitch + 15 * gi);
        O.nwdst[t] = (uint32_t)(row * ROWB + gi * 16);
    }
}

// 16 rows of one candidate per lane out of the circular window.  base0: LDS byte offset
// of the candidate's first row; wrap0: first row index that lies past slot 63 (>= 16:
// never).  Otherwise as me16_rows.
template <int ROWB>
__device__ __forceinline__ void me16r_rows(const uint8_t *s_win, const uint32_t *s_cur, int base0,
                                           int wrap0, uint32_t (&acc0)[4]) {
    Me16Q c[2], a[2];
    me16_ld(c[0], s_cur);
    me16_ld(a[0], s_win + base0);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int cur = k & 1, nxt = cur ^ 1;
        if (k + 1 < 16) {
            me16_ld(c[nxt], s_cur + (k + 1) * kMe16CurPitch);
            const int o0 = base0 + (k + 1) * ROWB - ((k + 1) >= wrap0 ? 64 * ROWB : 0);
            me16_ld(a[nxt], s_win + o0);
        }
        __builtin_amdgcn_sched_barrier(0);
        me16_count13(a[cur], c[cur], acc0);
    }
}

// ---- second pass as 2-row slabs ------------------------------------------------
// A block has up to 100 candidates: 64 fill the first pass, and the n2 = ncand - 64
// others (36 for an interior block) used to leave 28+ of the 64 lanes idle for 16
// more rows.  Instead each of them is cut into eight 2-row slabs: 8 * n2 (candidate,
// slab) items = ceil(n2 / 8) wave passes of 2 rows -- 10 row evaluations instead of
// 16 for n2 = 36.  Item q = 64 * sp + lane -> slab q & 7 (the lane's slab never
// changes), candidate 64 + (q >> 3).  Every item adds its #{r >= c} to its
// candidate's LDS counter (ds_add_u32).  The re-packed current rows are kept at a
// pitch of 20 dwords so that the eight slabs of a lane group read eight different
// bank groups.  Row loads run one row ahead of the evaluation, as in me16r_rows.
// NSP: wave passes when known at compile time (5 = interior block: straight-line code, so
// that hipcc keeps counted lgkmcnt waits and the one-row-ahead pipelining), 0 = ceil(n2 / 8).
template <int ROWB, int NSP>
__device__ __forceinline__ void me16r_slabs(const uint8_t *s_win, const uint32_t *s_cur,
                                            uint32_t *s_part, int B, int n2, int ncj, int lane) {
    const int nsp = NSP ? NSP : (n2 + 7) >> 3;       // wave passes (uniform, 1..5)
    const uint32_t inv = 65536u / (uint32_t)ncj + 1u;  // cand / ncj for cand < 128, ncj <= 10
    const int slab = lane & 7, l3 = lane >> 3;
    const uint32_t *cptr =
        (const uint32_t *)__builtin_assume_aligned(s_cur + slab * 2 * kMe16CurPitch, 16);
    Me16Q a[2], c[2];
    int c2 = 0, o1 = 0;
    // issues the loads of row (t & 1) of sub-pass t >> 1
    auto ld = [&](int t, Me16Q &qa, Me16Q &qc) {
        if ((t & 1) == 0) {
            c2 = 8 * (t >> 1) + l3;
            const int cand = 64 + c2;
            const int ci = (int)(((uint32_t)cand * inv) >> 16), cj = cand - ci * ncj;
            const int slot = (B + 5 * ci + 2 * slab) & 63;
            const int o0 = slot * ROWB + cj * 16;
            o1 = o0 + (slot == 63 ? -63 * ROWB : ROWB);
            me16_ld(qc, cptr);
            me16_ld(qa, s_win + o0);
        } else {
            me16_ld(qc, cptr + kMe16CurPitch);
            me16_ld(qa, s_win + o1);
        }
    };
    ld(0, a[0], c[0]);
    uint32_t cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 2 * (NSP ? NSP : (kMe16Pass2Max + 7) / 8); t++) {
        const int cur = t & 1, nxt = cur ^ 1;
        if (NSP || t < 2 * nsp) {  // uniform
            int c2_this = c2;  // ld(t + 1) of an even t + 1 moves c2 on to the next sub-pass
            if (t + 1 < 2 * nsp) ld(t + 1, a[nxt], c[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            me16_keep4(c[cur]);  // per-lane addresses: keep the 4th read a ds_read_b128 too
            me16_count13(a[cur], c[cur], cnt);
            if (t & 1) {
                if (c2_this < n2) atomicAdd(&s_part[c2_this], cnt[0] + cnt[1] + cnt[2] + cnt[3]);
                cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
            }
        }
    }
}

// Empty kernel launched with one workgroup per CU right before k_me16r (launch_me_roll).
// k_me16r deals its column segments statically, and at 29 x 1080p a wave gets 3 or 4 of
// them; how long the kernel takes then depends on which two waves the dispatcher puts on
// the same SIMD (a "4" beside a "3" is fine, two "4"s are not), and that placement depends
// on where the previous dispatch left the workgroup dispatcher's per-CU state.  Measured
// on MI355X, same binary: k_me16r takes 0.82-0.83 ms after a kernel of 8,040 short
// workgroups, after a 1-workgroup kernel or after a memset, and 0.72-0.74 ms after any
// dispatch that visits every CU (256 to 4,096 workgroups, empty or not).  Schemes that
// do not depend on placement were measured too: equal-length runs in column-major order
// 0.77-0.79 ms (the waves of an XCD no longer sweep adjacent columns in step), per-XCD
// work queues 0.87 ms, the last round cut into shorter pieces 0.84 ms.
// It also zeroes the per-frame static-block counters k_me16r adds to (one memset less).
__global__ void k_me16r_primer(uint32_t *__restrict__ n_static, int n_frames) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_frames; i += gridDim.x * blockDim.x)
        n_static[i] = 0;
}

__global__ __launch_bounds__(64, 2) void k_me16r(
    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ curs, size_t cur_stride,
    Geom g, int n_frames, int seg_len, uint32_t *__restrict__ sums, int16_t *__restrict__ mv,
    uint32_t *__restrict__ n_static, unsigned long long *__restrict__ cost_out) {
    constexpr int ROWB = 18 * 16;
    __shared__ __attribute__((aligned(16)))
    uint8_t s_lds[kMe16Rows * ROWB + 16 * 4 * kMe16CurPitch + 4 * kMe16Pass2Max];
    uint8_t *const s_win = s_lds;
    uint32_t *const s_cur = (uint32_t *)(s_lds + kMe16Rows * ROWB);
    uint32_t *const s_part = s_cur + 16 * kMe16CurPitch;  // second-pass counters (me16r_slabs)
    constexpr int BS = 16, WIN = 32, STEP = 5;
    const int lane = threadIdx.x;

    // ---- work units: (frame, column segment, block column); the waves of one XCD
    // sweep neighbouring columns of the same segment at the same time, which is what
    // keeps the 75 %-overlapping window rows of adjacent columns in that XCD's L2
    // (measured: contiguous per-wave runs or dynamic queues are 5-6 % slower).
    const int nseg = (g.nby + seg_len - 1) / seg_len;
    const int per_frame = nseg * g.nbx, total = n_frames * per_frame;
    const int nx = 8;
    const int xcd = blockIdx.x % nx, wi = blockIdx.x / nx, nw = gridDim.x / nx;
    const int chunk = (total + nx - 1) / nx;
    const int u_end = min((xcd + 1) * chunk, total);

    Me16rOffs O;
    me16r_offsets<ROWB>(g, lane, O);

    for (int u = xcd * chunk + wi; u < u_end; u += nw) {
      const int f = u / per_frame, r = u - f * per_frame;
      const int seg = r / g.nbx, bxi = r - seg * g.nbx;
      const int by0 = seg * seg_len, by1 = min(by0 + seg_len, g.nby);
      const int x = bxi * BS, j_min = max(x - WIN, 0), j_max = min(x + WIN, g.W);
      const uint8_t *__restrict__ colbase = ref + (size_t)(3 * j_min);  // uniform

      // ---- the segment's first window: 64 rows from B (once per segment) ------------
      int B = max(by0 * BS - WIN, 0);
      {
          const uint8_t *__restrict__ wsrc = colbase + (size_t)B * g.pitch;
          uint4 stg[kMe16Groups];
#pragma unroll
          for (int q = 0; q < kMe16Groups; q++) stg[q] = load16_any_align(wsrc + O.full[q]);
#pragma unroll
          for (int q = 0; q < kMe16Groups; q++) {
              const int e = lane + 64 * q, row = e / kMe16Groups, gi = e - row * kMe16Groups;
              uint4 v = stg[q];
              v.w &= 0x00FFFFFFu;
              *(uint4 *)(s_win + ((B + row) & 63) * ROWB + gi * 16) = v;
          }
      }

      auto issue = [&](int byy, int Bwin, Me16rRegs &R) {
          const size_t blk_off = (size_t)(byy * BS) * g.pitch + (size_t)(48 * bxi);
          const uint8_t *__restrict__ cblk = curs + (size_t)f * cur_stride + blk_off;
          const uint8_t *__restrict__ rblk = ref + blk_off;
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
          const uint32_t *__restrict__ sb = sums + (size_t)(byy * g.nbx + bxi) * kMe16SumStride;
          R.sr[0] = sb[lane];
          R.sr[1] = sb[lane + 64];
          // rows Bwin + 64 .. Bwin + 79: what the next block down adds
          const uint8_t *__restrict__ nsrc = colbase + (size_t)(Bwin + 64) * g.pitch;
#pragma unroll
          for (int t = 0; t < kMe16NewN; t++) R.nw[t] = load16_any_align(nsrc + O.nw[t]);
      };

      Me16rRegs A;
      issue(by0, B, A);

      for (int by = by0; by < by1; by++) {
        const int y = by * BS;
        const bool last = by + 1 == by1;
        const int byn = last ? by : by + 1;
        const int Bn = max(byn * BS - WIN, 0);
        const bool roll = !last && Bn != B;  // wave-uniform
        Me16rRegs Bq;
        issue(byn, B, Bq);

        const size_t oidx = (size_t)f * g.N + (size_t)(by * g.nbx + bxi);

        // ---- static test (motion.py:109-116), as in k_me16 ---------------------------
        bool is_static = false;
        uint32_t s_c = 0;
        uint32_t s_d = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            s_d = __builtin_amdgcn_sad_u8(A.rv[q], A.cv[q], s_d);
            s_d = __builtin_amdgcn_sad_u8(A.rv[q], 0u, s_d);
            s_c = __builtin_amdgcn_sad_u8(A.cv[q], 0u, s_c);
        }
        s_d = wave_sum_u32(s_d - s_c);
        s_c = wave_sum_u32(s_c);
        const uint32_t sdiff = s_d >> 1;
        is_static = !g.never_static && sdiff <= g.thresh;
        if (is_static && lane == 0) {
            mv[2 * oidx + 0] = 0;
            mv[2 * oidx + 1] = 0;
            atomicAdd(&n_static[f], 1u);
            if (cost_out) cost_out[oidx] = sdiff;
        }
        if (!is_static) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int e = lane + 64 * q, row = e / 13, j = e - row * 13;
                const uint32_t padm = ((j & 3) == 3 || j == 12) ? 0xFF000000u : 0u;
                if (e < 16 * 13) s_cur[row * kMe16CurPitch + j] = ~A.cq[q] | padm;
            }
            if (lane < kMe16Pass2Max) s_part[lane] = 0;
            wave_lds_fence();

            // ---- candidate grid (motion.py:132-140); i_min = B ---------------------
            const int i_min = B, i_max = min(y + WIN, g.H);
            const int ri = i_max - BS - i_min, rj = j_max - BS - j_min;
            const int nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
            const int ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
            const int ncand = nci * ncj;
            const int cand0 = lane, cand1 = lane + 64;
            const bool act0 = cand0 < ncand, act1 = cand1 < ncand;
            int ci0 = 0, cj0 = 0;
            if (act0) { ci0 = cand0 / ncj; cj0 = cand0 - ci0 * ncj; }
            const int s0 = (B + STEP * ci0) & 63;
            const int base0 = s0 * ROWB + cj0 * 16;

            uint32_t acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0};
            me16r_rows<ROWB>(s_win, s_cur, base0, 64 - s0, acc0);
            if (ncand > 64) {  // candidates 64..99: 2-row slabs
                if (ncand == 100)
                    me16r_slabs<ROWB, 5>(s_win, s_cur, s_part, B, 36, ncj, lane);
                else
                    me16r_slabs<ROWB, 0>(s_win, s_cur, s_part, B, ncand - 64, ncj, lane);
                wave_lds_fence();
                acc1[0] = s_part[min(lane, kMe16Pass2Max - 1)];
            }

            const uint32_t n_lt0 = 768u + 64u - (acc0[0] + acc0[1] + acc0[2] + acc0[3]);
            const uint32_t n_lt1 = 768u + 64u - (acc1[0] + acc1[1] + acc1[2] + acc1[3]);
            const uint32_t cost0 = A.sr[0] + 256u * n_lt0 - s_c;
            const uint32_t cost1 = A.sr[1] + 256u * n_lt1 - s_c;
            uint32_t key = 0xFFFFFFFFu;
            if (act0) key = (cost0 << 7) | (uint32_t)cand0;
            if (act1) key = min(key, (cost1 << 7) | (uint32_t)cand1);
            key = wave_min_u32(key);
            if (lane == 0) {
                int bx = 0, byy = 0;
                unsigned long long c = 9999999999ull;
                if (key != 0xFFFFFFFFu) {
                    const int cb = (int)(key & 127u);
                    byy = i_min + (cb / ncj) * STEP;
                    bx = j_min + (cb % ncj) * STEP;
                    c = key >> 7;
                }
                mv[2 * oidx + 0] = (int16_t)(bx - x);
                mv[2 * oidx + 1] = (int16_t)(byy - y);
                if (cost_out) cost_out[oidx] = c;
            }
            wave_lds_fence();  // all reads of this block's window are done
        }
        // ---- roll: the 16 new rows replace the 16 oldest (slots B .. B+15 mod 64) -------
        if (roll) {
#pragma unroll
            for (int q = 0; q < kMe16NewN; q++) {
                uint4 v = Bq.nw[q];
                v.w &= 0x00FFFFFFu;
                if (lane + 64 * q < 16 * kMe16Groups)
                    *(uint4 *)(s_win + (B & 63) * ROWB + O.nwdst[q]) = v;
            }
        }
        B = Bn;
        A = Bq;
      }
      wave_lds_fence();  // the next unit reloads the whole window
    }
}

// VCS_ME16_ROLL: 0 = never (k_me16 only), 1 = when worthwhile (default), 2 = always
// (tests force it so that small frames exercise the rolling kernel too).
inline int me16r_mode() {
    static const int v = [] {
        const char *e = getenv("VCS_ME16_ROLL");
        return e ? (e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1) : 1;
    }();
    return v;
}
// Blocks per column segment: about 34, the column split evenly (1080p: 67 block
// rows -> 2 segments; 4K: 135 -> 4).  VCS_ME16_SEG overrides.
inline int me16r_seg_len(int nby) {
    static const int forced = [] {
        const char *e = getenv("VCS_ME16_SEG");
        const int s = e ? atoi(e) : 0;
        return s >= 1 && s <= 4096 ? s : 0;
    }();
    if (forced) return forced;
    const int nseg = nby <= 34 ? 1 : (nby + 17) / 34;
    return (nby + nseg - 1) / nseg;
}
// The rolling kernel needs enough column segments to occupy the chip: a single
// 1080p frame has only 240 of them for 2,048 resident waves, so small batches
// stay on k_me16 (one block per work item).
inline bool me16r_worthwhile(const Geom &g, int cus, int n) {
    if (me16r_mode() != 1) return me16r_mode() == 2;
    const int seg = me16r_seg_len(g.nby);
    const long long units = (long long)n * ((g.nby + seg - 1) / seg) * g.nbx;
    return units * 2 >= (long long)cus * 8 * 3;
}
inline int launch_me_roll(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                          const uint8_t *curs, size_t stride, int n, uint32_t *sums, int16_t *mv,
                          uint32_t *ns, unsigned long long *cost, hipEvent_t ev_start = nullptr,
                          hipEvent_t ev_stop = nullptr) {
    if ((long long)g.N * n > 0x7FFFFFFF) return -1;
    const int seg = me16r_seg_len(g.nby);
    const long long units = (long long)n * ((g.nby + seg - 1) / seg) * g.nbx;
    dim3 gd(me16_grid(cus, units)), block(64);
    // the events (optional) take this dispatch's own start / end timestamps
    hipExtLaunchKernelGGL(k_me16r, gd, block, 0, stream, ev_start, ev_stop, 0, ref, curs, stride, g, n,
                          seg, sums, mv, ns, cost);
    return 0;
}
// Must be the last kernel dispatched before launch_me_roll (see k_me16r_primer); also
// zeroes the n per-frame static-block counters.
inline void launch_me_roll_primer(hipStream_t stream, int cus, uint32_t *ns, int n) {
    static const bool primer = [] {
        const char *e = getenv("VCS_ME16_PRIMER");  // 0 = a single workgroup (A/B measurements)
        return !(e && e[0] == '0');
    }();
    hipLaunchKernelGGL(k_me16r_primer, dim3(primer ? cus : 1), dim3(64), 0, stream, ns, n);
}

}  // namespace vcs
