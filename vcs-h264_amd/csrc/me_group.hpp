// me_group.hpp -- k_me16g<W>: the rolling-window motion search (me_roll.hpp) with the
// window SHARED BY W FRAMES.
//
// Every P frame of a GOP is searched against the same raw I frame (encoder.py:51-52),
// so the waves that work on block column c / row segment s of W different P frames all
// need the same 64-row reference window.  k_me16r gave each of them a private 18.4 KB
// copy in LDS (LDS, not registers, capped the kernel at 2 waves per SIMD) and each of
// them fetched and parked every window row itself.  Here a workgroup is W waves = W
// consecutive P frames on the same (column, segment):
//   * ONE window per workgroup, a ring of 80 rows (64 live + the 16 the next block down
//     adds) + 15 mirror rows (ring slots 0..14 repeated behind slot 79, so that a
//     candidate's 16 rows never wrap: the row loop has immediate offsets only, no
//     per-row select);
//   * the rows are fetched and parked cooperatively: 16 new rows x 13 groups = 208
//     16-byte items per block for the whole workgroup (one per lane at W = 4) instead of
//     4 per lane per wave; the 64-row fill at a segment start likewise;
//   * each wave still runs a complete, independent block search for its own frame
//     (static test, re-packed current block, two candidate passes, arg-min);
//   * one s_barrier per block: the new rows go into ring slots that the PREVIOUS block
//     read, so "everybody is done with block by-1" is the only ordering needed; the
//     barrier waits for LDS only (lgkmcnt), never for the global prefetches (vmcnt).
// The row loop keeps a single copy of the window row and the current row in registers:
// both are dead after the 13 v_lerp_u8 of a row, and the loads of the next row are
// issued right there, ahead of the 26 v_and / v_bcnt -- half the registers of the
// double-buffered form, which is what lets 3-4 waves share a SIMD.
// Results are identical to k_me16 / k_me16r / k_me_generic.
#pragma once
#include <hip/hip_ext.h>

#include "me_fast.hpp"
#include "me_roll.hpp"

namespace vcs {

constexpr int kMe16gRing = 80;     // ring rows: 64 live + 16 incoming
constexpr int kMe16gMirror = 15;   // ring slots 0..14 repeated at 80..94
constexpr int kMe16gRowB = 18 * 16;
constexpr int kMe16gWinBytes = (kMe16gRing + kMe16gMirror) * kMe16gRowB;
constexpr int kMe16gWaveBytes = 16 * 4 * kMe16CurPitch + 4 * 40;  // s_cur + s_part (36 used)
constexpr int kMe16gDumpBytes = 64 * 16;                          // where the prefetch's bytes go (all waves)

// LDS-only workgroup barrier: every wave's DS operations are complete and visible, the
// global loads in flight (register prefetch of the next block) are NOT waited for.
__device__ __forceinline__ void wg_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One row step of the single-buffered pipeline: the 13 v_lerp_u8 of the row held in (a, c),
// then -- a and c are dead -- the loads of the next row (na, nc; NEXT = false: none), then
// the 13 v_and and 13 v_bcnt of this row, which cover the LDS latency of those loads.
// KEEPC: c is read with per-lane addresses (slab pass): keep its 4th read a ds_read_b128.
template <bool NEXT, bool KEEPC>
__device__ __forceinline__ void me16g_step(Me16Q &a, Me16Q &c, const void *na, const void *nc_,
                                           uint32_t (&cnt)[4]) {
    const uint32_t r[13] = {a.q0.x, a.q0.y, a.q0.z, a.q0.w, a.q1.x, a.q1.y, a.q1.z,
                            a.q1.w, a.q2.x, a.q2.y, a.q2.z, a.q2.w, a.q3.x};
    const uint32_t nc[13] = {c.q0.x, c.q0.y, c.q0.z, c.q0.w, c.q1.x, c.q1.y, c.q1.z,
                             c.q1.w, c.q2.x, c.q2.y, c.q2.z, c.q2.w, c.q3.x};
    me16_keep4(a);
    if (KEEPC) me16_keep4(c);
    uint32_t t[13];
#pragma unroll
    for (int d = 0; d < 13; d++) t[d] = __builtin_amdgcn_lerp(r[d], nc[d], 0x01010101u);
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT) {
#ifndef VCS_EXP_NOCUR   // experiment builds only (never shipped): sensitivity to the broadcast reads
        me16_ld(c, nc_);
#endif
#ifndef VCS_EXP_NOWIN
        me16_ld(a, na);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
#ifndef VCS_EXP_NOAND
#pragma unroll
    for (int d = 0; d < 13; d++) t[d] &= 0x80808080u;
#endif
    __builtin_amdgcn_sched_barrier(0);
    // popcount + accumulate selects to ONE v_bcnt_u32_b32 (x, acc); the empty asm keeps hipcc
    // from re-associating the chain into v_bcnt(x, 0) + v_add3 (as inline asm text the
    // instruction itself drew an s_nop -- 4 issue cycles -- from the hazard recogniser
    // every time an accumulator came round again)
#pragma unroll
    for (int d = 0; d < 13; d++) {
        cnt[d & 3] = (uint32_t)__builtin_popcount(t[d]) + cnt[d & 3];
        asm volatile("" : "+v"(cnt[d & 3]));
    }
    __builtin_amdgcn_sched_barrier(0);
}

// 16 rows of one candidate.  wbase: the candidate's first row in the ring (never wraps:
// mirror rows); s_cur: the wave's re-packed current block.
template <int ROWB>
__device__ __forceinline__ void me16g_rows(const uint8_t *wbase, const uint32_t *s_cur,
                                           uint32_t (&cnt)[4]) {
    Me16Q a, c;
    me16_ld(c, s_cur);
    me16_ld(a, wbase);
#pragma unroll
    for (int k = 0; k < 15; k++)
        me16g_step<true, false>(a, c, wbase + (k + 1) * ROWB, s_cur + (k + 1) * kMe16CurPitch, cnt);
    me16g_step<false, false>(a, c, nullptr, nullptr, cnt);
}

// Second pass as 2-row slabs (me16r_slabs), on the mirrored ring: the second row of a slab
// is always the next ring row.  Bslot = B mod 80.  Same single-buffered pipeline.
template <int ROWB, int NSP>
__device__ __forceinline__ void me16g_slabs(const uint8_t *s_win, const uint32_t *s_cur,
                                            uint32_t *s_part, int Bslot, int n2, int ncj, int lane) {
    constexpr int MAXSP = NSP ? NSP : (kMe16Pass2Max + 7) / 8;
    const int nsp = NSP ? NSP : (n2 + 7) >> 3;
    const uint32_t inv = 65536u / (uint32_t)ncj + 1u;
    const int slab = lane & 7, l3 = lane >> 3;
    const uint32_t *cptr =
        (const uint32_t *)__builtin_assume_aligned(s_cur + slab * 2 * kMe16CurPitch, 16);
    auto wptr = [&](int sp) {  // first ring row of this lane's (candidate, slab) item of sub-pass sp
        const int cand = 64 + 8 * sp + l3;
        const int ci = (int)(((uint32_t)cand * inv) >> 16), cj = cand - ci * ncj;
        int slot = Bslot + 5 * ci + 2 * slab;
        slot -= slot >= kMe16gRing ? kMe16gRing : 0;
        return s_win + slot * ROWB + cj * 16;
    };
    uint32_t cnt[4] = {0, 0, 0, 0};
    Me16Q a, c;
    const uint8_t *p = wptr(0);
    me16_ld(c, cptr);
    me16_ld(a, p);
#pragma unroll
    for (int sp = 0; sp < MAXSP; sp++) {
        if (NSP || sp < nsp) {  // uniform
            me16g_step<true, true>(a, c, p + ROWB, cptr + kMe16CurPitch, cnt);
            const bool more = NSP ? sp + 1 < NSP : sp + 1 < nsp;
            if (sp + 1 < MAXSP && more) {
                p = wptr(sp + 1);
                me16g_step<true, true>(a, c, p, cptr, cnt);
            } else {
                me16g_step<false, true>(a, c, nullptr, nullptr, cnt);
            }
            const int c2 = 8 * sp + l3;
            if (c2 < n2) atomicAdd(&s_part[c2], cnt[0] + cnt[1] + cnt[2] + cnt[3]);
            cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
        }
    }
}

template <int W>
struct Me16gCfg {
    static constexpr int kLanes = 64 * W;
    static constexpr int kNewN = (16 * kMe16Groups + kLanes - 1) / kLanes;   // items per lane per roll
    static constexpr int kFillN = (64 * kMe16Groups + kLanes - 1) / kLanes;  // items per lane per fill
};

template <int W>
struct Me16gRegs {
    uint32_t rv[3], cv[3];
    uint32_t cq[4];
    uint32_t sr[2];
    uint4 nw[Me16gCfg<W>::kNewN];  // this lane's share of the 16 rows the next block adds
};

// WPS: waves per SIMD the register budget is held to (a workgroup puts W / 4 waves on each SIMD)
#ifdef VCS_EXP_PHASES   // experiment builds only: [start, end] of every workgroup on the 100 MHz clock
__device__ unsigned long long g_me16_wg_t[2 * 4096];
__device__ unsigned long long g_me16_hw[8 * 4096];   // per wave: HW_ID | XCC_ID << 32
#endif

template <int W, int WPS>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
void k_me16g(const uint8_t *__restrict__ ref, const uint8_t *__restrict__ curs, size_t cur_stride,
             Geom g, int n_frames, int seg_len, const uint32_t *__restrict__ sums,
             int16_t *__restrict__ mv, uint32_t *__restrict__ n_static,
             unsigned long long *__restrict__ cost_out, unsigned long long *__restrict__ clk_out,
             int pf_on) {
    using Cfg = Me16gCfg<W>;
    // shader-clock cycles this workgroup lives (s_memtime): the longest-lived workgroup's
    // count over the launch's duration is the clock the chip held under this load
    const unsigned long long clk0 = clk_out ? __builtin_amdgcn_s_memtime() : 0ull;
#ifdef VCS_EXP_PHASES   // experiment builds only: shader cycles per phase of the block loop -> clk_out[1..9]
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = clk0;
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
    constexpr int ROWB = kMe16gRowB, RING = kMe16gRing;
    constexpr int BS = 16, WIN = 32, STEP = 5, NG = kMe16Groups;
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[kMe16gWinBytes + W * kMe16gWaveBytes + kMe16gDumpBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the frame's addresses stay uniform
    uint8_t *const s_win = s_lds;
    uint32_t *const s_cur = (uint32_t *)(s_lds + kMe16gWinBytes + wave * kMe16gWaveBytes);
    uint32_t *const s_part = s_cur + 16 * kMe16CurPitch;
    // Infinity Cache prefetch for the tail kernel (vcs_api.hip: k_l3_prefetch explains why): every
    // workgroup streams its own slice of the launch's frames, one 16-byte LDS-direct load per lane
    // and block -- no register, no consumer; the bytes land in a dump area nobody reads.  The
    // search's own accesses (48-byte row pieces) do not leave lines in that cache.
    uint8_t *const s_dump = s_lds + kMe16gWinBytes + W * kMe16gWaveBytes;   // one area for all waves: nobody reads it
    const uint32_t pf_total = pf_on ? (uint32_t)(((size_t)n_frames * cur_stride) >> 4) : 0u;   // uint4 units
    const uint32_t pf_per = (pf_total + gridDim.x - 1) / gridDim.x;
    uint32_t pf_pos = min(blockIdx.x * pf_per, pf_total);
    const uint32_t pf_end = min(pf_pos + pf_per, pf_total);
    const __amdgpu_buffer_rsrc_t r_pf = buf_rsrc(curs, pf_total << 4);

    // ---- work units: (frame group, column segment, block column); XCD-contiguous chunks so
    // that the workgroups of one XCD sweep neighbouring columns of the same segment together.
    // (A per-XCD unit queue -- workgroups taking the next unit from a counter -- was measured in
    // round 2 and is not kept: no faster on uniform content, slower on content with static
    // regions)
    const int nseg = (g.nby + seg_len - 1) / seg_len;
    const int ngroups = (n_frames + W - 1) / W;
    const int per_group = nseg * g.nbx, total = ngroups * per_group;
    const int nx = 8;
    const int xcd = blockIdx.x % nx, wi = blockIdx.x / nx, nwg = gridDim.x / nx;
    const int chunk = (total + nx - 1) / nx;
    const int u_end = min((xcd + 1) * chunk, total);

    // the reference (with the slack rows the unclamped window loads reach into) and its
    // sum(r) table as buffers
    const __amdgpu_buffer_rsrc_t r_ref = buf_rsrc(ref, (uint32_t)((g.H + kMe16Rows + 2) * g.pitch));
    const __amdgpu_buffer_rsrc_t r_sums = buf_rsrc(sums, (uint32_t)(g.N * kMe16SumStride * 4));

    // ---- per-lane offsets, once ---------------------------------------------------
    uint32_t o_blk[3], o_cq[4], o_new_src[Cfg::kNewN], o_new_dst[Cfg::kNewN];
    int o_new_row[Cfg::kNewN];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int idx = lane + 64 * t, row = idx / 12, col = idx - row * 12;
        o_blk[t] = (uint32_t)(row * g.pitch + 4 * col);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int e = min(lane + 64 * t, 16 * 13 - 1), row = e / 13, j = e - row * 13;
        o_cq[t] = (uint32_t)(row * g.pitch + (j == 12 ? 45 : 15 * (j >> 2) + 4 * (j & 3)));
    }
#pragma unroll
    for (int t = 0; t < Cfg::kNewN; t++) {
        const int e = tid + Cfg::kLanes * t;              // may be >= 208: loaded (clamped), not stored
        const int ec = min(e, 16 * NG - 1), row = ec / NG, gi = ec - row * NG;
        o_new_src[t] = (uint32_t)(row * g.pitch + 15 * gi);
        o_new_dst[t] = (uint32_t)(row * ROWB + gi * 16);
        o_new_row[t] = e < 16 * NG ? row : 99;            // 99: never stored
    }

    for (int u = xcd * chunk + wi; u < u_end; u += nwg) {
      const int gi_ = u / per_group, r = u - gi_ * per_group;
      const int seg = r / g.nbx, bxi = r - seg * g.nbx;
      const int f = gi_ * W + wave;
      const bool active = f < n_frames;                   // wave-uniform
      const int fc = active ? f : n_frames - 1;           // idle waves: in-bounds dummy loads
      const int by0 = seg * seg_len, by1 = min(by0 + seg_len, g.nby);
      const int x = bxi * BS, j_min = max(x - WIN, 0), j_max = min(x + WIN, g.W);
      const uint32_t col_off = (uint32_t)(3 * j_min);  // window's left edge, bytes into a row
      // this unit's current frame as a buffer (uniform base; lane offsets stay 32-bit)
      const __amdgpu_buffer_rsrc_t r_cur = buf_rsrc(curs + (size_t)fc * cur_stride, (uint32_t)(g.H * g.pitch + 64));  // + the slot's slack: the last row's unaligned dwords reach past it
      // candidate columns of this block column (motion.py:137-140) and cand / ncj by
      // multiplication (cand < 128, ncj <= 10): once per unit, not a division per block
      const int rj = j_max - BS - j_min;
      const int ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
      const uint32_t inv_ncj = ncj ? 65536u / (uint32_t)ncj + 1u : 0u;

      int B = max(by0 * BS - WIN, 0);
      int Bslot = B % RING;
      uint32_t n_static_unit = 0;  // static blocks of this wave's frame in this unit: ONE atomic per unit
      // ---- the segment's first window: rows B .. B+63, cooperatively -----------------
      if (W > 1) wg_lds_barrier();  // everybody is done with the previous unit's window
      {
          const uint32_t wsrc = (uint32_t)(B * g.pitch) + col_off;
          uint4 stg[Cfg::kFillN];
#pragma unroll
          for (int q = 0; q < Cfg::kFillN; q++) {
              const int e = min(tid + Cfg::kLanes * q, 64 * NG - 1), row = e / NG, gg = e - row * NG;
              stg[q] = buf_ld128(r_ref, (uint32_t)(row * g.pitch + 15 * gg), wsrc);
          }
#pragma unroll
          for (int q = 0; q < Cfg::kFillN; q++) {
              const int e = tid + Cfg::kLanes * q, row = e / NG, gg = e - row * NG;
              uint4 v = stg[q];
              v.w &= 0x00FFFFFFu;
              int slot = Bslot + row;
              slot -= slot >= RING ? RING : 0;
              if (e < 64 * NG) {
                  *(uint4 *)(s_win + slot * ROWB + gg * 16) = v;
                  if (slot < kMe16gMirror) *(uint4 *)(s_win + (slot + RING) * ROWB + gg * 16) = v;
              }
          }
      }

      auto issue = [&](int byy, int Bwin, Me16gRegs<W> &R) {
          const uint32_t blk_off = (uint32_t)((byy * BS) * g.pitch + 48 * bxi);
#pragma unroll
          for (int t = 0; t < 3; t++) {
              R.rv[t] = buf_ld32(r_ref, o_blk[t], blk_off);
              R.cv[t] = buf_ld32(r_cur, o_blk[t], blk_off);
          }
#pragma unroll
          for (int t = 0; t < 4; t++) R.cq[t] = buf_ld32(r_cur, o_cq[t], blk_off);  // unaligned dwords
          const uint32_t sb = (uint32_t)((byy * g.nbx + bxi) * (kMe16SumStride * 4));
          R.sr[0] = buf_ld32(r_sums, (uint32_t)(4 * lane), sb);
          R.sr[1] = buf_ld32(r_sums, (uint32_t)(4 * lane + 256), sb);
          // this lane's share of rows Bwin + 64 .. Bwin + 79 (what the block below adds)
          const uint32_t nsrc = (uint32_t)((Bwin + 64) * g.pitch) + col_off;
#pragma unroll
          for (int t = 0; t < Cfg::kNewN; t++) R.nw[t] = buf_ld128(r_ref, o_new_src[t], nsrc);
      };

      Me16gRegs<W> A;
      issue(by0, B, A);
      uint32_t st_mv = 0, st_clo = 0, st_chi = 0;  // lane b: result of block by0 + 64 k + b
      VCS_TICK(8);  // unit prologue (fill) and the previous unit's epilogue

      for (int by = by0; by < by1; by++) {
        VCS_TICK(7);  // loop end (vmcnt wait, A = Bq)
        const int y = by * BS;
        const bool last = by + 1 == by1;
        const int byn = last ? by : by + 1;
        const int Bn = max(byn * BS - WIN, 0);
        const bool roll = !last && Bn != B;  // uniform over the workgroup
        Me16gRegs<W> Bq;
        issue(byn, Bn, Bq);
        if (pf_pos < pf_end) {   // uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_pf, (__attribute__((address_space(3))) void *)s_dump, 16,
                                                     (int)((pf_pos + (uint32_t)tid) << 4), 0, 0, 0);
            pf_pos += 64 * W;
        }
        VCS_TICK(0);  // issue

        // everybody has finished block by-1 (its window rows may be replaced) and the rows
        // parked during block by-1 (or the fill) are visible
#ifdef VCS_EXP_NOBARRIER   // experiment builds only (results are wrong): what the lock-step costs
        wave_lds_fence();
#else
        if (W > 1) wg_lds_barrier(); else wave_lds_fence();
#endif
        VCS_TICK(1);  // barrier
        if (roll) {
            // rows B+64 .. B+79 -> ring slots of rows B-16 .. B-1 (read by block by-1 only)
            int ns = Bslot + 64;
            ns -= ns >= RING ? RING : 0;  // a multiple of 16: the 16 rows never wrap
#pragma unroll
            for (int q = 0; q < Cfg::kNewN; q++) {
                uint4 v = A.nw[q];
                v.w &= 0x00FFFFFFu;
                if (o_new_row[q] < 16) {
                    *(uint4 *)(s_win + ns * ROWB + o_new_dst[q]) = v;
                    if (ns == 0 && o_new_row[q] < kMe16gMirror)
                        *(uint4 *)(s_win + RING * ROWB + o_new_dst[q]) = v;
                }
            }
        }

        VCS_TICK(2);  // roll
        if (active) {
        // ---- static test (motion.py:109-116), as in k_me16 ---------------------------
        uint32_t s_c = 0, s_d = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            s_d = __builtin_amdgcn_sad_u8(A.rv[q], A.cv[q], s_d);
            s_d = __builtin_amdgcn_sad_u8(A.rv[q], 0u, s_d);
            s_c = __builtin_amdgcn_sad_u8(A.cv[q], 0u, s_c);
        }
#ifdef VCS_EXP_NOREDUCE   // experiment builds only (results are wrong): cost of the wave reductions
        const uint32_t sdiff = 0xFFFFFFFFu;
        s_c = __builtin_amdgcn_readfirstlane(s_c + s_d);
#else
        s_d = wave_sum_u32(s_d - s_c);
        s_c = wave_sum_u32(s_c);
        const uint32_t sdiff = s_d >> 1;
#endif
        const bool is_static = !g.never_static && sdiff <= g.thresh;
        n_static_unit += is_static ? 1u : 0u;
        uint32_t mvw = 0u;                 // static block: [0, 0], cost = the difference
        unsigned long long cw = sdiff;
        VCS_TICK(3);  // static test
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
            const int ri = i_max - BS - i_min;
            const int nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
            const int ncand = nci * ncj;
            const int cand0 = lane, cand1 = lane + 64;
            const bool act0 = cand0 < ncand, act1 = cand1 < ncand;
            int ci0 = 0, cj0 = 0;
            if (act0) { ci0 = (int)(((uint32_t)cand0 * inv_ncj) >> 16); cj0 = cand0 - ci0 * ncj; }
            int s0 = Bslot + STEP * ci0;
            s0 -= s0 >= RING ? RING : 0;

            uint32_t acc0[4] = {0, 0, 0, 0};
            uint32_t second = 0;
            VCS_TICK(4);  // current block re-pack, grid
#ifndef VCS_EXP_NOCORE    // experiment builds only: everything but the comparisons
            me16g_rows<ROWB>(s_win + s0 * ROWB + cj0 * 16, s_cur, acc0);
#endif
#ifdef VCS_EXP_NOCORE
            if (false) {
#else
            if (ncand > 64) {
#endif
                if (ncand == 100)
                    me16g_slabs<ROWB, 5>(s_win, s_cur, s_part, Bslot, 36, ncj, lane);
                else
                    me16g_slabs<ROWB, 0>(s_win, s_cur, s_part, Bslot, ncand - 64, ncj, lane);
                wave_lds_fence();
                second = s_part[min(lane, kMe16Pass2Max - 1)];
            }

            VCS_TICK(5);  // comparisons
            const uint32_t n_lt0 = 768u + 64u - (acc0[0] + acc0[1] + acc0[2] + acc0[3]);
            const uint32_t n_lt1 = 768u + 64u - second;
            const uint32_t cost0 = A.sr[0] + 256u * n_lt0 - s_c;
            const uint32_t cost1 = A.sr[1] + 256u * n_lt1 - s_c;
            uint32_t key = 0xFFFFFFFFu;
            if (act0) key = (cost0 << 7) | (uint32_t)cand0;
            if (act1) key = min(key, (cost1 << 7) | (uint32_t)cand1);
#ifdef VCS_EXP_NOREDUCE
            key = __builtin_amdgcn_readfirstlane(key);
#else
            key = wave_min_u32(key);
#endif
            {
                int bx = 0, byy = 0;
                cw = 9999999999ull;
                if (key != 0xFFFFFFFFu) {
                    const int cb = (int)(key & 127u);
                    const int cbi = (int)(((uint32_t)cb * inv_ncj) >> 16);
                    byy = i_min + cbi * STEP;
                    bx = j_min + (cb - cbi * ncj) * STEP;
                    cw = key >> 7;
                }
                mvw = (uint32_t)(uint16_t)(bx - x) | ((uint32_t)(uint16_t)(byy - y) << 16);  // [dx, dy]
            }
            wave_lds_fence();  // s_cur / s_part are rewritten by the next block
        }
        // Results wait in lane (by - by0) % 64 of a register and leave 64 blocks at a time: a
        // store inside this loop sits between the prefetch and the s_waitcnt vmcnt(0) that
        // collects it (gfx9 counts stores in vmcnt), so every block would wait for its own
        // store to be acknowledged.
        VCS_TICK(6);  // reduction, result
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
        if (Bn != B) {
            Bslot += 16;
            Bslot -= Bslot >= RING ? RING : 0;
        }
        B = Bn;
        A = Bq;
      }
      // (real video is mostly static blocks: one atomic per block on 29 counters serialises in L2)
      if (active && n_static_unit && lane == 0) atomicAdd(&n_static[f], n_static_unit);
    }
    if (clk_out && tid == 0) atomicMax(clk_out, __builtin_amdgcn_s_memtime() - clk0);
#ifdef VCS_EXP_PHASES
    if (clk_out && lane == 0)
        for (int k = 0; k < 9; k++) atomicAdd(&clk_out[1 + k], ph[k]);
    if (lane == 0 && blockIdx.x < 4096) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
        g_me16_hw[8 * blockIdx.x + wave] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_me16_wg_t[2 * blockIdx.x] = wg_t0;
        g_me16_wg_t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#undef VCS_TICK
}

// VCS_ME16_GROUP: frames per workgroup of the shared-window kernel (1, 2, 4, 8; default 4);
// 0 = use k_me16r (one private window per wave).  VCS_ME16_WPS: waves per SIMD the kernel is
// compiled for (2, 3, 4).
inline int me16g_group() {
    static const int v = [] {
        const char *e = getenv("VCS_ME16_GROUP");
        const int w = e ? atoi(e) : 4;
        return (w >= 0 && w <= 6) || w == 8 ? w : 4;
    }();
    return v;
}
inline int me16g_wps() {
    static const int v = [] {
        const char *e = getenv("VCS_ME16_WPS");
        const int w = e ? atoi(e) : 4;
        return (w >= 2 && w <= 4) ? w : 4;
    }();
    return v;
}
// Blocks per column segment.  A unit is now W frames deep, so there are W times fewer
// units than with k_me16r: shorter segments keep the last round of the static dealing full.
inline int me16g_seg_len(int nby) {
    static const int forced = [] {
        const char *e = getenv("VCS_ME16G_SEG");
        const int s = e ? atoi(e) : 0;
        return s >= 1 && s <= 4096 ? s : 0;
    }();
    if (forced) return forced;
    const int nseg = nby <= 17 ? 1 : (nby + 8) / 17;
    return (nby + nseg - 1) / nseg;
}

// The shared-window kernel needs enough (frame group, segment, column) units to fill the
// chip; smaller batches stay on k_me16r / k_me16.  VCS_ME16_ROLL=2 (tests) forces it.
inline bool me16g_worthwhile(const Geom &g, int cus, int n) {
    const int w = me16g_group();
    if (w == 0) return false;
    if (me16r_mode() == 2) return true;
    const int seg = me16g_seg_len(g.nby);
    const long long units = (long long)((n + w - 1) / w) * ((g.nby + seg - 1) / seg) * g.nbx;
    const int per_cu = 4 * me16g_wps() / w > 0 ? 4 * me16g_wps() / w : 1;
    return units * 2 >= (long long)cus * per_cu * 3 && n >= w;
}

template <int W, int WPS>
inline int launch_me_group_t(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                             const uint8_t *curs, size_t stride, int n, uint32_t *sums, int16_t *mv,
                             uint32_t *ns, unsigned long long *cost, unsigned long long *clk,
                             bool *prefetch, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int seg = me16g_seg_len(g.nby);
    const long long units = (long long)((n + W - 1) / W) * ((g.nby + seg - 1) / seg) * g.nbx;
    // resident workgroups: WPS waves per SIMD = 4 * WPS waves per CU = 4 * WPS / W workgroups
    static const int per_cu_env = [] {
        const char *e = getenv("VCS_ME16G_WG_PER_CU");
        return e ? atoi(e) : 0;
    }();
    const int per_cu = per_cu_env > 0 ? per_cu_env : (4 * WPS / W > 0 ? 4 * WPS / W : 1);
    long long grid = (long long)cus * per_cu;
    grid = (grid + 7) / 8 * 8;
    const long long cap = (units + 7) / 8 * 8;
    dim3 gd((unsigned)(grid > cap ? cap : grid)), block(64 * W);
    // The in-kernel prefetch covers the frames only if every workgroup walks enough blocks for
    // its slice (one 16-byte load per lane and block); otherwise the caller runs k_l3_prefetch.
    int pf_on = 0;
    if (prefetch && *prefetch) {
        const long long total16 = (long long)n * (long long)stride / 16;
        const long long per = (total16 + gd.x - 1) / gd.x, need_blocks = (per + 64 * W - 1) / (64 * W);
        const int last_seg = g.nby - (g.nby - 1) / seg * seg;            // the shortest segment
        const long long have_blocks = units / gd.x * (long long)(last_seg < seg ? last_seg : seg);
        pf_on = total16 * 16 < 0xFFFFFFFFll && have_blocks >= need_blocks;
        *prefetch = pf_on != 0;
    }
    hipExtLaunchKernelGGL((k_me16g<W, WPS>), gd, block, 0, stream, ev_start, ev_stop, 0, ref, curs,
                          stride, g, n, seg, (const uint32_t *)sums, mv, ns, cost, clk, pf_on);
    return 0;
}

inline int launch_me_group(const Geom &g, hipStream_t stream, int cus, const uint8_t *ref,
                           const uint8_t *curs, size_t stride, int n, uint32_t *sums, int16_t *mv,
                           uint32_t *ns, unsigned long long *cost, unsigned long long *clk = nullptr,
                           bool *prefetch = nullptr, hipEvent_t ev_start = nullptr,
                           hipEvent_t ev_stop = nullptr) {
    if ((long long)g.N * n > 0x7FFFFFFF) return -1;
#define VCS_ME16G(Wv, Pv)                                                                         \
    return launch_me_group_t<Wv, Pv>(g, stream, cus, ref, curs, stride, n, sums, mv, ns, cost,    \
                                     clk, prefetch, ev_start, ev_stop)
    const int w = me16g_group(), p = me16g_wps();
    if (w == 1) { if (p == 2) VCS_ME16G(1, 2); if (p == 4) VCS_ME16G(1, 4); VCS_ME16G(1, 3); }
    if (w == 2) { if (p == 2) VCS_ME16G(2, 2); if (p == 4) VCS_ME16G(2, 4); VCS_ME16G(2, 3); }
    if (w == 8) { if (p == 2) VCS_ME16G(8, 2); if (p == 4) VCS_ME16G(8, 4); VCS_ME16G(8, 3); }
    if (w == 3) { if (p == 3) VCS_ME16G(3, 3); VCS_ME16G(3, 4); }
    if (w == 5) { if (p == 3) VCS_ME16G(5, 3); VCS_ME16G(5, 4); }
    if (w == 6) { if (p == 3) VCS_ME16G(6, 3); VCS_ME16G(6, 4); }
    if (p == 2) VCS_ME16G(4, 2);
    if (p == 3) VCS_ME16G(4, 3);
    VCS_ME16G(4, 4);
#undef VCS_ME16G
}

}  // namespace vcs
