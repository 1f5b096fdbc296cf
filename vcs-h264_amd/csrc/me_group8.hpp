// me_group8.hpp -- k_me8g<W>: the shared rolling window of k_me16g (me_group.hpp) at the
// reference's OWN default operating point (main.py:15 BLOCK_SIZE = 8: +-16 px window,
// stride-3 grid; motion.py:18,132).  Same arithmetic as k_me8 (me_fast8.hpp: 3-pixel groups
// padded to 16-byte LDS slots, cost = sum(r) - sum(c) + 256 * #{r < c}, lane = candidate,
// 64 candidates = one full wave), same results bit for bit.
//
// k_me8 re-fetched and re-parked its whole 32-row window (352 16-byte items, 5.6 KB) for every
// 8x8 block, although the block below shares 24 of those rows and the same block of 28 other
// P frames needs exactly the same window: per block it spent 2.3 k cycles of a SIMD on 640
// cycles of comparisons.  Here a workgroup is W waves = W consecutive P frames walking down
// one block column (a segment of it) together:
//   * one window per workgroup, a ring of 40 rows (32 live + the 8 the next block adds) + 7
//     mirror rows (ring slots 0..6 repeated behind slot 39: a candidate's 8 rows never wrap);
//   * a block brings in 8 new rows x 11 groups = 88 items for the WHOLE workgroup;
//   * one LDS-only s_barrier per block (new rows go where the previous block's rows were);
//   * single-buffered row pipeline (me16g_step's scheme): next row's loads behind the lerps.
#pragma once
#include <hip/hip_ext.h>

#include "me_fast8.hpp"
#include "me_group.hpp"

namespace vcs {

// blocks per roll of the window: the workgroup meets (one s_barrier) and parks new reference rows
// once per kMe8gRB blocks, 8 * kMe8gRB rows at a time; between two meetings its waves drift.
#ifndef VCS_ME8G_RB
#define VCS_ME8G_RB 2
#endif
constexpr int kMe8gRB = VCS_ME8G_RB;
constexpr int kMe8gLive = kMe8Rows + 8 * (kMe8gRB - 1);   // rows the blocks between two meetings read
constexpr int kMe8gRing = kMe8gLive + 8 * kMe8gRB;        // + the rows parked for the blocks after the next meeting
constexpr int kMe8gMirror = 7;
constexpr int kMe8gWinBytes = (kMe8gRing + kMe8gMirror) * kMe8RowB;
constexpr int kMe8gWaveBytes = 8 * 48;  // the wave's re-packed current block

struct Me8Q {
    uint4 q0, q1, q2;
};
__device__ __forceinline__ void me8_ld(Me8Q &Q, const void *p) {
    const uint4 *a = (const uint4 *)p;
    Q.q0 = a[0]; Q.q1 = a[1]; Q.q2 = a[2];
}

// one row step: 8 v_lerp_u8, the next row's loads, 8 v_and, 8 v_bcnt (see me16g_step)
template <bool NEXT>
__device__ __forceinline__ void me8g_step(Me8Q &a, Me8Q &c, const void *na, const void *nc_,
                                          uint32_t (&cnt)[2]) {
    const uint32_t r[8] = {a.q0.x, a.q0.y, a.q0.z, a.q1.x, a.q1.y, a.q1.z, a.q2.x, a.q2.y};
    const uint32_t nc[8] = {c.q0.x, c.q0.y, c.q0.z, c.q1.x, c.q1.y, c.q1.z, c.q2.x, c.q2.y};
    asm volatile("" ::"v"(a.q0.w), "v"(a.q1.w), "v"(a.q2.z), "v"(a.q2.w));  // keep whole ds_read_b128s
    uint32_t t[8];
#pragma unroll
    for (int d = 0; d < 8; d++) t[d] = __builtin_amdgcn_lerp(r[d], nc[d], 0x01010101u);
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT) {
        me8_ld(c, nc_);
        me8_ld(a, na);
    }
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
}

template <int W>
struct Me8gCfg {
    static constexpr int kLanes = 64 * W;
    static constexpr int kNewN = (8 * kMe8gRB * kMe8Groups + kLanes - 1) / kLanes;  // items per lane per roll
    static constexpr int kFillN = (kMe8gLive * kMe8Groups + kLanes - 1) / kLanes;   // items per lane per fill
};

template <int W>
struct Me8gRegs {
    uint32_t rv, cv, cq, sr;
};

#ifdef VCS_EXP_PHASES   // experiment builds only: shader cycles per phase of the block loop
__device__ unsigned long long g_me8_phases[16];
#endif

template <int W, int WPS>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
void k_me8g(const uint8_t *__restrict__ ref, const uint8_t *__restrict__ curs, size_t cur_stride,
            Geom g, int n_frames, int seg_len, const uint32_t *__restrict__ sums,
            int16_t *__restrict__ mv, uint32_t *__restrict__ n_static,
            unsigned long long *__restrict__ cost_out, int pf_on) {
    using Cfg = Me8gCfg<W>;
#ifdef VCS_EXP_PHASES
    unsigned long long ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
#define VCS_TICK(k)                                                        \
    do {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
        ph[k] += t_ - ph_t;                                                \
        ph_t = t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                 \
    } while (0)
#else
#define VCS_TICK(k) do { } while (0)
#endif
    constexpr int ROWB = kMe8RowB, RING = kMe8gRing, NG = kMe8Groups, RB = kMe8gRB, LIVE = kMe8gLive;
    constexpr int BS = 8, WIN = 16, STEP = 3;
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[kMe8gWinBytes + W * kMe8gWaveBytes + kMe16gDumpBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t *const s_win = s_lds;
    uint32_t *const s_cur = (uint32_t *)(s_lds + kMe8gWinBytes + wave * kMe8gWaveBytes);
    // Infinity Cache prefetch of the launch's frames for the tail kernel (see k_me16g)
    uint8_t *const s_dump = s_lds + kMe8gWinBytes + W * kMe8gWaveBytes;   // one area for all waves: nobody reads it
    const uint32_t pf_total = pf_on ? (uint32_t)(((size_t)n_frames * cur_stride) >> 4) : 0u;
    const uint32_t pf_per = (pf_total + gridDim.x - 1) / gridDim.x;
    uint32_t pf_pos = min(blockIdx.x * pf_per, pf_total);
    const uint32_t pf_end = min(pf_pos + pf_per, pf_total);
    const __amdgpu_buffer_rsrc_t r_pf = buf_rsrc(curs, pf_total << 4);

    const int nseg = (g.nby + seg_len - 1) / seg_len;
    const int ngroups = (n_frames + W - 1) / W;
    const int per_group = nseg * g.nbx, total = ngroups * per_group;
    const int nx = 8;
    const int xcd = blockIdx.x % nx, wi = blockIdx.x / nx, nwg = gridDim.x / nx;
    const int chunk = (total + nx - 1) / nx;
    const int u_end = min((xcd + 1) * chunk, total);

    const __amdgpu_buffer_rsrc_t r_ref = buf_rsrc(ref, (uint32_t)((g.H + kMe16Rows + 2) * g.pitch));
    const __amdgpu_buffer_rsrc_t r_sums = buf_rsrc(sums, (uint32_t)(g.N * kMe8SumStride * 4));

    // ---- per-lane offsets, once ---------------------------------------------------
    const int e0 = min(lane, 47), row0 = e0 / 6, col0 = e0 - row0 * 6;
    const uint32_t o_blk = (uint32_t)(row0 * g.pitch + 4 * col0);
    const int crow = lane >> 3, cqq = lane & 7;
    const int csl = cqq < 3 ? 0 : cqq < 6 ? 1 : 2, cd = cqq - 3 * csl;
    const uint32_t o_cq = (uint32_t)(crow * g.pitch + 9 * csl + 4 * cd);
    const uint32_t cur_padm = cd == 2 ? 0xFFFFFF00u : (csl == 2 && cd == 1) ? 0xFFFF0000u : 0u;
    const int cur_dst = crow * 12 + csl * 4 + cd;
    uint32_t o_new_src[Cfg::kNewN], o_new_dst[Cfg::kNewN];
    int o_new_row[Cfg::kNewN];
#pragma unroll
    for (int t = 0; t < Cfg::kNewN; t++) {
        const int e = tid + Cfg::kLanes * t;
        const int ec = min(e, 8 * RB * NG - 1), row = ec / NG, gi = ec - row * NG;
        o_new_src[t] = (uint32_t)(row * g.pitch + 9 * gi);
        o_new_dst[t] = (uint32_t)(gi * 16);
        o_new_row[t] = e < 8 * RB * NG ? row : 999;
    }

    for (int u = xcd * chunk + wi; u < u_end; u += nwg) {
      const int gi_ = u / per_group, r = u - gi_ * per_group;
      const int seg = r / g.nbx, bxi = r - seg * g.nbx;
      const int f = gi_ * W + wave;
      const bool active = f < n_frames;
      const int fc = active ? f : n_frames - 1;
      const int by0 = seg * seg_len, by1 = min(by0 + seg_len, g.nby);
      const int x = bxi * BS, j_min = max(x - WIN, 0), j_max = min(x + WIN, g.W);
      const uint32_t col_off = (uint32_t)(3 * j_min);
      const __amdgpu_buffer_rsrc_t r_cur = buf_rsrc(curs + (size_t)fc * cur_stride, (uint32_t)(g.H * g.pitch + 64));  // + the slot's slack: the last row's unaligned dwords reach past it
      // candidate columns of this block column (motion.py:137-140); lane / ncj by multiplication
      const int rj = j_max - BS - j_min;
      const int ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
      const uint32_t inv_ncj = ncj ? 65536u / (uint32_t)ncj + 1u : 0u;
      const int lci = (int)(((uint32_t)lane * inv_ncj) >> 16), lcj = lane - lci * ncj;  // this lane's candidate

      const int B0 = max(by0 * BS - WIN, 0);
      uint32_t n_static_unit = 0;  // static blocks of this wave's frame in this unit: ONE atomic per unit
      if (W > 1) wg_lds_barrier();
      {
          const uint32_t wsrc = (uint32_t)(B0 * g.pitch) + col_off;
          const int Bslot = B0 % RING;
          uint4 stg[Cfg::kFillN];
#pragma unroll
          for (int q = 0; q < Cfg::kFillN; q++) {
              const int e = min(tid + Cfg::kLanes * q, LIVE * NG - 1), row = e / NG, gg = e - row * NG;
              stg[q] = buf_ld128(r_ref, (uint32_t)(row * g.pitch + 9 * gg), wsrc);
          }
#pragma unroll
          for (int q = 0; q < Cfg::kFillN; q++) {
              const int e = tid + Cfg::kLanes * q, row = e / NG, gg = e - row * NG;
              uint4 v = stg[q];
              v.z &= 0xFFu;  // bytes 9..15 of the slot: pad
              v.w = 0u;
              int slot = Bslot + row;
              slot -= slot >= RING ? RING : 0;
              if (e < LIVE * NG) {
                  *(uint4 *)(s_win + slot * ROWB + gg * 16) = v;
                  if (slot < kMe8gMirror) *(uint4 *)(s_win + (slot + RING) * ROWB + gg * 16) = v;
              }
          }
      }

      auto issue = [&](int byy, Me8gRegs<W> &R) {
          const uint32_t blk_off = (uint32_t)((byy * BS) * g.pitch + 24 * bxi);
          R.rv = buf_ld32(r_ref, o_blk, blk_off);
          R.cv = buf_ld32(r_cur, o_blk, blk_off);
          R.cq = buf_ld32(r_cur, o_cq, blk_off);  // unaligned dword
          R.sr = buf_ld32(r_sums, (uint32_t)(4 * lane), (uint32_t)((byy * g.nbx + bxi) * (kMe8SumStride * 4)));
      };
      // rows Rg + LIVE .. Rg + LIVE + 8 RB - 1 of the meeting whose first block has window base Rg
      uint4 nwq[Cfg::kNewN];
      auto issue_rows = [&](int Rg) {
          const uint32_t nsrc = (uint32_t)((Rg + LIVE) * g.pitch) + col_off;
#pragma unroll
          for (int t = 0; t < Cfg::kNewN; t++) nwq[t] = buf_ld128(r_ref, o_new_src[t], nsrc);
      };

      Me8gRegs<W> A;
      issue(by0, A);
      issue_rows(B0);
      uint32_t st_mv = 0, st_clo = 0, st_chi = 0;  // lane b: result of block by0 + 64 k + b
      VCS_TICK(8);

      for (int by = by0; by < by1; by++) {
        VCS_TICK(7);
        const int y = by * BS;
        const bool last = by + 1 == by1;
        const int B = max(y - WIN, 0), Bslot = B % RING;   // this block's window base (motion.py:132-135)
        const int kk = (by - by0) % RB;                     // position between two meetings
        const bool meet = kk == 0;
        const bool park = meet && by + RB < by1;            // somebody comes after the next meeting
        Me8gRegs<W> Bq;
        issue(last ? by : by + 1, Bq);
        if (meet && pf_pos < pf_end) {   // uniform; once per meeting is plenty at 8x8
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_pf, (__attribute__((address_space(3))) void *)s_dump, 16,
                                                     (int)((pf_pos + (uint32_t)tid) << 4), 0, 0, 0);
            pf_pos += 64 * W;
        }

        VCS_TICK(0);
        // Meeting: everybody has finished the blocks before this one (their oldest rows may be
        // replaced) and the rows parked at the previous meeting (or the fill) are visible.
#ifdef VCS_EXP_NOBARRIER   // experiment builds only (results are wrong): what the lock-step costs
        wave_lds_fence();
#else
        if (meet) { if (W > 1) wg_lds_barrier(); else wave_lds_fence(); }
#endif
        VCS_TICK(1);
        if (park) {
            // rows B + LIVE .. B + LIVE + 8 RB - 1 (read after the NEXT meeting) go where the rows
            // below B were: the ring is LIVE + 8 RB rows, slot = row % RING
            const int ns = (B + LIVE) % RING;
#pragma unroll
            for (int q = 0; q < Cfg::kNewN; q++) {
                uint4 v = nwq[q];
                v.z &= 0xFFu;
                v.w = 0u;
                int slot = ns + o_new_row[q];
                slot -= slot >= RING ? RING : 0;
                if (o_new_row[q] < 8 * RB) {
                    *(uint4 *)(s_win + slot * ROWB + o_new_dst[q]) = v;
                    if (slot < kMe8gMirror) *(uint4 *)(s_win + (slot + RING) * ROWB + o_new_dst[q]) = v;
                }
            }
        }
        // the rows the meeting after the next one parks: loaded while the last block before it runs
        if (kk == RB - 1) {
            const int byg = by + 1;                       // first block after the next meeting
            if (byg + RB < by1) issue_rows(max(byg * BS - WIN, 0));
        }

        VCS_TICK(2);
        if (active) {
        // ---- static test (motion.py:109-116) over the 48 dwords of the block ------
        uint32_t s_c = 0, s_d = 0;
        if (lane < 48) {
            s_d = __builtin_amdgcn_sad_u8(A.rv, A.cv, 0u);
            s_d = __builtin_amdgcn_sad_u8(A.rv, 0u, s_d);
            s_c = __builtin_amdgcn_sad_u8(A.cv, 0u, 0u);
        }
        s_d = wave_sum_u32(s_d - s_c);
        s_c = wave_sum_u32(s_c);
        const uint32_t sdiff = s_d >> 1;
        const bool is_static = !g.never_static && sdiff <= g.thresh;
        n_static_unit += is_static ? 1u : 0u;
        uint32_t mvw = 0u;                 // static block: [0, 0], cost = the difference
        unsigned long long cw = sdiff;
        VCS_TICK(3);
        if (!is_static) {
            s_cur[cur_dst] = ~A.cq | cur_padm;  // negated current block, pad / foreign bytes -> 0xFF
            wave_lds_fence();

            // ---- candidate grid (motion.py:132-140); i_min = B ---------------------
            const int i_min = B, i_max = min(y + WIN, g.H);
            const int ri = i_max - BS - i_min;
            const int nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
            const int ncand = nci * ncj;  // <= 64
            const bool act = lane < ncand;
            const int ci = act ? lci : 0, cj = act ? lcj : 0;
            int s0 = Bslot + STEP * ci;
            s0 -= s0 >= RING ? RING : 0;
            const uint8_t *p = s_win + s0 * ROWB + cj * 16;

            uint32_t cnt[2] = {0, 0};
            Me8Q a, c;
            me8_ld(c, s_cur);
            me8_ld(a, p);
            VCS_TICK(4);
#ifndef VCS_EXP_NOCORE   // experiment builds only (results are wrong): everything but the comparisons
#pragma unroll
            for (int k = 0; k < BS - 1; k++)
                me8g_step<true>(a, c, p + (k + 1) * ROWB, s_cur + (k + 1) * 12, cnt);
            me8g_step<false>(a, c, nullptr, nullptr, cnt);
#else
            cnt[0] = a.q0.x ^ c.q0.x;
#endif

            VCS_TICK(5);
            // cost = sum(r) - sum(c) + 256 * #{r < c}; 64 pad bytes always counted
            const uint32_t n_lt = 192u + 64u - (cnt[0] + cnt[1]);
            const uint32_t cost = A.sr + 256u * n_lt - s_c;
            uint32_t key = act ? ((cost << 7) | (uint32_t)lane) : 0xFFFFFFFFu;  // motion.py:149
            key = wave_min_u32(key);
            {
                int bx = 0, byy = 0;   // motion.py:102
                cw = 9999999999ull;    // motion.py:120
                if (key != 0xFFFFFFFFu) {
                    const int cb = (int)(key & 127u);
                    const int cbi = (int)(((uint32_t)cb * inv_ncj) >> 16);
                    byy = i_min + cbi * STEP;
                    bx = j_min + (cb - cbi * ncj) * STEP;
                    cw = key >> 7;
                }
                mvw = (uint32_t)(uint16_t)(bx - x) | ((uint32_t)(uint16_t)(byy - y) << 16);
            }
            wave_lds_fence();  // s_cur is rewritten by the next block
        }
        VCS_TICK(6);
        // results leave 64 blocks at a time (see k_me16g: no store between prefetch and its wait)
        const int slot = (by - by0) & 63;
        st_mv = lane == slot ? mvw : st_mv;
        if (cost_out) {
            st_clo = lane == slot ? (uint32_t)cw : st_clo;
            st_chi = lane == slot ? (uint32_t)(cw >> 32) : st_chi;
        }
        if (slot == 63 || last) {
            if (lane <= slot) {
                const size_t o = (size_t)f * g.N + (size_t)((by - slot + lane) * g.nbx + bxi);
                *(uint32_t *)(mv + 2 * o) = st_mv;
                if (cost_out) cost_out[o] = (unsigned long long)st_clo | ((unsigned long long)st_chi << 32);
            }
        }
        }  // active
        A = Bq;
      }
      // (real video is mostly static blocks: one atomic per block on 29 counters serialises in L2)
      if (active && n_static_unit && lane == 0) atomicAdd(&n_static[f], n_static_unit);
    }
#ifdef VCS_EXP_PHASES
    if (lane == 0)
        for (int k = 0; k < 9; k++) atomicAdd(&g_me8_phases[1 + k], ph[k]);
#endif
#undef VCS_TICK
}

// VCS_ME8_GROUP: frames per workgroup (1, 2, 4, 8; default 4; 0 = k_me8 only).
inline int me8g_group() {
    static const int v = [] {
        const char *e = getenv("VCS_ME8_GROUP");
        const int w = e ? atoi(e) : 4;
        return (w == 0 || w == 1 || w == 2 || w == 4 || w == 8) ? w : 4;
    }();
    return v;
}
inline int me8g_wps() {
    static const int v = [] {
        const char *e = getenv("VCS_ME8_WPS");
        const int w = e ? atoi(e) : 8;   // 59 VGPRs, 12.6 KB of LDS per workgroup: eight waves per SIMD fit
        return (w >= 2 && w <= 8) ? w : 8;
    }();
    return v;
}
inline int me8g_seg_len(int nby) {
    static const int forced = [] {
        const char *e = getenv("VCS_ME8G_SEG");
        const int s = e ? atoi(e) : 0;
        return s >= 1 && s <= 4096 ? s : 0;
    }();
    if (forced) return forced;
    const int nseg = nby <= 34 ? 1 : (nby + 17) / 34;
    return (nby + nseg - 1) / nseg;
}
inline bool me8g_worthwhile(const Geom &g, int cus, int n) {
    const int w = me8g_group();
    if (w == 0) return false;
    if (me16r_mode() == 2) return true;  // VCS_ME16_ROLL=2: tests force the rolling kernels
    if (me16r_mode() == 0) return false;
    const int seg = me8g_seg_len(g.nby);
    const long long units = (long long)((n + w - 1) / w) * ((g.nby + seg - 1) / seg) * g.nbx;
    const int per_cu = 4 * me8g_wps() / w > 0 ? 4 * me8g_wps() / w : 1;
    return units * 2 >= (long long)cus * per_cu * 3 && n >= w;
}

template <int W, int WPS>
inline int launch_me_group8_t(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                              const uint8_t *curs, size_t stride, int n, uint32_t *sums, int16_t *mv,
                              uint32_t *ns, unsigned long long *cost, bool *prefetch,
                              hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int seg = me8g_seg_len(g.nby);
    const long long units = (long long)((n + W - 1) / W) * ((g.nby + seg - 1) / seg) * g.nbx;
    static const int per_cu_env = [] {
        const char *e = getenv("VCS_ME8G_WG_PER_CU");
        return e ? atoi(e) : 0;
    }();
    const int per_cu = per_cu_env > 0 ? per_cu_env : (4 * WPS / W > 0 ? 4 * WPS / W : 1);
    long long grid = (long long)cus * per_cu;
    grid = (grid + 7) / 8 * 8;
    const long long cap = (units + 7) / 8 * 8;
    dim3 gd((unsigned)(grid > cap ? cap : grid)), block(64 * W);
    int pf_on = 0;   // as in launch_me_group_t: only if every workgroup meets often enough for its slice
    if (prefetch && *prefetch) {
        const long long total16 = (long long)n * (long long)stride / 16;
        const long long per = (total16 + gd.x - 1) / gd.x, need = (per + 64 * W - 1) / (64 * W);
        const int last_seg = g.nby - (g.nby - 1) / seg * seg;
        const long long have = units / gd.x * (long long)((last_seg < seg ? last_seg : seg) / kMe8gRB);
        pf_on = total16 * 16 < 0xFFFFFFFFll && have >= need;
        *prefetch = pf_on != 0;
    }
    hipExtLaunchKernelGGL((k_me8g<W, WPS>), gd, block, 0, stream, ev_start, ev_stop, 0, ref, curs,
                          stride, g, n, seg, (const uint32_t *)sums, mv, ns, cost, pf_on);
    return 0;
}

inline int launch_me_group8(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                            const uint8_t *curs, size_t stride, int n, uint32_t *sums, int16_t *mv,
                            uint32_t *ns, unsigned long long *cost, bool *prefetch = nullptr,
                            hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    if ((long long)g.N * n > 0x7FFFFFFF) return -1;
#define VCS_ME8G(Wv, Pv)                                                                           \
    return launch_me_group8_t<Wv, Pv>(g, stream, cus, ref, curs, stride, n, sums, mv, ns, cost,    \
                                      prefetch, ev_start, ev_stop)
    const int w = me8g_group(), p = me8g_wps();
    if (w == 1) { if (p <= 3) VCS_ME8G(1, 3); if (p >= 6) VCS_ME8G(1, 6); VCS_ME8G(1, 4); }
    if (w == 2) { if (p <= 3) VCS_ME8G(2, 3); if (p >= 6) VCS_ME8G(2, 6); VCS_ME8G(2, 4); }
    if (w == 8) { if (p <= 3) VCS_ME8G(8, 2); if (p >= 6) VCS_ME8G(8, 6); VCS_ME8G(8, 4); }
    if (p <= 3) VCS_ME8G(4, 3);
    if (p >= 8) VCS_ME8G(4, 8);
    if (p >= 6) VCS_ME8G(4, 6);
    if (p == 5) VCS_ME8G(4, 5);
    VCS_ME8G(4, 4);
#undef VCS_ME8G
}

}  // namespace vcs
