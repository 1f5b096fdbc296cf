// me_bits8.hpp -- k_me8b: the bit-plane motion search (me_bits.hpp) for the reference's own default:
// 8x8 blocks, +-16 px window, stride-3 grid (main.py:15, motion.py:18,132).
//
// Same arithmetic -- the cost is sum(r) - sum(c) + 256 * #{r < c}, and #{r < c} over bit planes is an
// 8-step v_bitop3_b32 borrow chain + one v_bcnt per 32 byte pairs -- re-dimensioned:
//   * a block is 8 rows x 24 bytes = two 96-byte chunks of FOUR rows (3 dwords x 8 planes each);
//   * 64 candidates (8 x 8) = exactly one wave pass: no second pass, no LDS copy of the planes;
//   * the reference's chunk table has an entry for every picture row (a candidate's rows start at
//     i_min + 3 ci, any residue mod 4), every block column a and each of its 8 candidate columns
//     j = max(8 a - 16, 0) + 3 m: 768 bytes per (row, block column), 199 MB at 1080p;
//   * a lane keeps BOTH chunks of its candidate in registers (48) and loops over up to NF frames; a
//     frame's 48 plane dwords (192 contiguous bytes) arrive by three scalar loads, one wait per frame;
//   * the wave's planes are put into L2 first by LDS-direct loads into a dump area (a scalar load that
//     misses L2 takes ~2,000 cycles);
//   * a block costs 54 comparison instructions per frame against ~230 vector instructions in k_me8g,
//     but four times as many blocks carry the per-block fixed work (unit decode, arg-min, result), so
//     a wave takes up to 16 frames.
// k_bits_cur8 makes the planes, the static test and sum(c) (one thread per (block, row quad)),
// k_bits_ref8 the chunk table and its chunk sums, from which the first frame's workgroups of k_bits_cur8 add
// up the sum(r) table of a new reference (k_me8<kMe8Sums> makes it for the single-frame kernels).
// Results are identical to k_me8 / k_me8g / k_me_generic (tests/test_gpu_parity.py).
#pragma once
#include "me_bits.hpp"
#include "me_fast8.hpp"

namespace vcs {

constexpr int kBits8M = 8;                  // candidate columns per block column
constexpr int kBits8WA = 6 * kBits8M * 4;   // dwords per (picture row, block column): [q = t / 4][m][4]
constexpr int kBits8Blk = 2 * kBitsT;       // dwords per 8x8x3 block: two row quads

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// rtab[((w * nbx + a) * 6 + q) * 32 + m * 4 + e] = plane dword t = 4 q + e of the 96-byte string
// rows w .. w + 3, bytes 3 j .. 3 j + 23 each, j = max(8 a - 16, 0) + 3 m.  One thread per (w, a, m).
// r2 (nullable): r2[(w * nbx + a) * 8 + m] = byte sum of the chunk's 96 bytes: two of them, four rows apart,
// are a candidate's sum(r) (k_bits_cur8 adds them up: the sum(r) table without k_me8<kMe8Sums>'s launch).
__global__ __launch_bounds__(256) void k_bits_ref8(const uint8_t *__restrict__ ref, uint32_t ref_bytes, Geom g,
                                                   uint32_t *__restrict__ rtab, uint32_t *__restrict__ r2) {
    const uint32_t total = (uint32_t)g.H * (uint32_t)g.nbx * kBits8M;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total) return;
    const uint32_t m = idx % kBits8M, wa = idx / kBits8M;
    const uint32_t a = wa % (uint32_t)g.nbx, w = wa / (uint32_t)g.nbx;
    const int j = max(8 * (int)a - 16, 0) + 3 * (int)m;
    const __amdgpu_buffer_rsrc_t r_ref = buf_rsrc(ref, ref_bytes);   // reads past the slack rows return 0
    const uint32_t off = w * (uint32_t)g.pitch + 3u * (uint32_t)j;
    uint32_t s[24];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 6; i++) s[6 * r + i] = buf_ld32(r_ref, off + (uint32_t)r * (uint32_t)g.pitch + 4u * i, 0);
    if (r2) {
        uint32_t rs = 0;
#pragma unroll
        for (int i = 0; i < 24; i++) rs = __builtin_amdgcn_sad_u8(s[i], 0u, rs);
        r2[idx] = rs;
    }
    uint4 *dst = (uint4 *)(rtab + (size_t)wa * kBits8WA + m * 4);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = s[8 * d + i];
        bits_transpose8(x);
        dst[(2 * d) * kBits8M] = make_uint4(x[0], x[1], x[2], x[3]);
        dst[(2 * d + 1) * kBits8M] = make_uint4(x[4], x[5], x[6], x[7]);
    }
}

// cpl[((f * N + blk) * 2 + q) * 24 + 8 d + k]: row quad q of block blk of frame f as plane dwords;
// binfo[f * N + blk] = {sum(max(ref - cur, 0)) over the co-located block, sum(cur)}; ns_part[f][workgroup]
// = static blocks.  One thread per (block, row quad): 4 rows x 24 bytes of the frame and of the
// reference as 8-byte loads, three 32-byte transposes, 96 contiguous bytes out through LDS (one
// store instruction = 1 KB in a row); the two quads of a block meet by DPP.
constexpr int kBits8CurBlocks = 128;
__global__ __launch_bounds__(256) void k_bits_cur8(const uint8_t *__restrict__ curs, size_t cur_stride,
                                                   const uint8_t *__restrict__ ref, Geom g, int n_frames,
                                                   uint32_t *__restrict__ cpl, uint2 *__restrict__ binfo,
                                                   uint32_t *__restrict__ ns_part, const uint32_t *__restrict__ r2,
                                                   uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_ns;
    __shared__ uint4 s_out[4 * 384];
    const int tid = threadIdx.x, q = tid & 1;
    // XCD x takes the x-th eighth of every frame's blocks, frame after frame (k_bits_cur)
    const int per = (int)(gridDim.x >> 3) / n_frames;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int f = idx / per;
    const int wg_blk0 = (xcd * per + (idx - f * per)) * kBits8CurBlocks;
    const int blk = wg_blk0 + (tid >> 1);
    if (tid == 0) s_ns = 0;
    __syncthreads();
    const uint32_t fbytes = (uint32_t)(g.H * g.pitch);
    const __amdgpu_buffer_rsrc_t r_cur = buf_rsrc(curs + (size_t)f * cur_stride, fbytes);
    const __amdgpu_buffer_rsrc_t r_ref = buf_rsrc(ref, fbytes);
    const int blkc = min(blk, g.N - 1);
    const int byi = blkc / g.nbx, bxi = blkc - byi * g.nbx;
    const uint32_t o = (uint32_t)((byi * 8 + 4 * q) * g.pitch + 24 * bxi);
    uint32_t s[24], rr[24];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const uint32_t oi = o + (uint32_t)r * (uint32_t)g.pitch + 8u * i;
            const u32x2_t vc = __builtin_amdgcn_raw_buffer_load_b64(r_cur, (int)oi, 0, 0);
            const u32x2_t vr = __builtin_amdgcn_raw_buffer_load_b64(r_ref, (int)oi, 0, 0);
            s[6 * r + 2 * i] = vc.x; s[6 * r + 2 * i + 1] = vc.y;
            rr[6 * r + 2 * i] = vr.x; rr[6 * r + 2 * i + 1] = vr.y;
        }
    uint32_t sad = 0, sr = 0, sc = 0;
#pragma unroll
    for (int i = 0; i < 24; i++) {
        sad = __builtin_amdgcn_sad_u8(rr[i], s[i], sad);
        sr = __builtin_amdgcn_sad_u8(rr[i], 0u, sr);
        sc = __builtin_amdgcn_sad_u8(s[i], 0u, sc);
    }
    // sum(max(r - c, 0)) = (sum|r - c| + sum r - sum c) / 2, over the two quads of the block
    uint32_t sd2 = sad + sr - sc;
    sd2 += dpp_u32<0xB1>(sd2);
    sc += dpp_u32<0xB1>(sc);
    const uint32_t sdiff = sd2 >> 1;
    uint4 *const s_w = s_out + (tid >> 6) * 384;
    const int lane = tid & 63;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = s[8 * d + i];
        bits_transpose8(x);
        s_w[lane * 6 + 2 * d] = make_uint4(x[0], x[1], x[2], x[3]);
        s_w[lane * 6 + 2 * d + 1] = make_uint4(x[4], x[5], x[6], x[7]);
    }
    wave_lds_fence();
    {
        const int wblk0 = wg_blk0 + (tid >> 6) * 32;   // first block of this wave (32 blocks x 2 quads)
        uint4 *dst = (uint4 *)(cpl + ((size_t)f * g.N + wblk0) * kBits8Blk);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int e = i * 64 + lane;                // 16-byte piece of the wave's 32 blocks x 12 pieces
            if (wblk0 + e / 12 < g.N) __builtin_nontemporal_store(*(const u32x4_t *)&s_w[e], (u32x4_t *)&dst[e]);
        }
    }
    const bool head = q == 0 && blk < g.N;
    if (head) binfo[(size_t)f * g.N + blk] = make_uint2(sdiff, sc);
    const unsigned long long st = __ballot(head && !g.never_static && sdiff <= g.thresh);
    if ((tid & 63) == 0 && st) atomicAdd(&s_ns, (uint32_t)__builtin_popcountll(st));
    __syncthreads();
    if (tid == 0) ns_part[(size_t)f * (8 * per) + xcd * per + (idx - f * per)] = s_ns;
    // The sum(r) table of a new reference (what k_me8<kMe8Sums> makes: sums[blk * 64 + cand] = byte sum of the
    // candidate's 8 x 24 bytes, 0 for slots without a candidate), by the workgroups of the first frame: a
    // candidate's sum = the chunk sums of its two row quads (r2 != null: the table is due).  As k_bits_cur.
    if (r2 && f == 0) {
        constexpr int BS = 8, WIN = 16, STEP = 3;
        const size_t rw = (size_t)g.nbx * kBits8M;   // r2 words per picture row
#pragma unroll 1
        for (int it = 0; it < kBits8CurBlocks * kMe8SumStride / 256; it++) {
            const int e = tid + 256 * it, bi = e / kMe8SumStride, cand = e - bi * kMe8SumStride;
            const int b = wg_blk0 + bi;
            if (b >= g.N) continue;
            const int by = b / g.nbx, bx = b - by * g.nbx;
            const int x = bx * BS, y = by * BS;
            const int i_min = max(y - WIN, 0), i_max = min(y + WIN, g.H);
            const int j_min = max(x - WIN, 0), j_max = min(x + WIN, g.W);
            const int ri = i_max - BS - i_min, rj = j_max - BS - j_min;
            const int nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
            const int ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
            uint32_t v = 0;
            if (cand < nci * ncj) {
                const int ci = cand / ncj, cj = cand - ci * ncj;
                const uint32_t *p2 = r2 + (size_t)(i_min + STEP * ci) * rw + (size_t)bx * kBits8M + cj;
                v = p2[0] + p2[4 * rw];
            }
            sums[(size_t)b * kMe8SumStride + cand] = v;
        }
    }
}

#ifndef VCS_BITS8_NF
#define VCS_BITS8_NF 16
#endif
constexpr int kBits8NF = VCS_BITS8_NF;   // frames per wave the 8x8 search is unrolled for

template <int NF>
__global__ __launch_bounds__(256) void k_me8b(const uint32_t *__restrict__ rtab, const uint32_t *__restrict__ cpl,
                                              const uint2 *__restrict__ binfo, const uint32_t *__restrict__ sums,
                                              Geom g, int n_frames, int ngroups, int16_t *__restrict__ mv,
                                              unsigned long long *__restrict__ cost_out,
                                              const uint8_t *__restrict__ pf_base, uint32_t pf_total16,
                                              int n_search_wgs, const uint32_t *__restrict__ ns_part,
                                              int ns_per_frame, uint32_t *__restrict__ n_static,
                                              uint32_t per_col_magic, uint32_t ngroups_magic) {
    constexpr int BS = 8, WIN = 16, STEP = 3;
    // per wave: 1 KB where prefetched bytes go (nobody reads them), reused by the arg-min ([frame][64 keys])
    static_assert(NF <= 16 && 64 % NF == 0, "the arg-min gives frame f the lanes f * (64 / NF) ..");
    __shared__ __attribute__((aligned(16))) uint8_t s_dump[4 * NF * 256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Infinity Cache prefetch for the tail kernel: the workgroups behind the search's own stream the
    // launch's P frames (me_bits.hpp, k_me16b)
    if ((int)blockIdx.x >= n_search_wgs) {
        const uint32_t nw = (gridDim.x - (uint32_t)n_search_wgs) * 4u;
        const uint32_t w = (blockIdx.x - (uint32_t)n_search_wgs) * 4u + (uint32_t)wave;
        const __amdgpu_buffer_rsrc_t r_pf = buf_rsrc(pf_base, pf_total16 << 4);
        uint32_t it = 0;
        for (uint32_t i = w * 64u; i < pf_total16; i += nw * 64u, it++) {   // uniform trip count
            if (i + (uint32_t)lane < pf_total16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    r_pf, (__attribute__((address_space(3))) void *)(s_dump + wave * (NF * 256)), 16,
                    (int)((i + (uint32_t)lane) << 4), 0, 0, 0);
            if ((it & 7u) == 7u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        return;
    }
    if (blockIdx.x == 0) {   // static blocks per frame: the sums of k_bits_cur8's per-workgroup counts
        for (int f = wave; f < n_frames; f += 4) {
            uint32_t v = 0;
            for (int i = lane; i < ns_per_frame; i += 64) v += ns_part[(size_t)f * ns_per_frame + i];
            v = wave_sum_u32(v);
            if (lane == 0) n_static[f] = v;
        }
    }
    // units = (block position, frame group); XCD x takes the block columns x, x + 8, ... and walks
    // down each of them (k_me16b: why); a workgroup = 4 consecutive units
    const int per_col = g.nby * ngroups;
    const int chunk = (((g.nbx + 7) / 8) * per_col + 3) & ~3;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3;
    const int ul = wg * 4 + wave;
    const int lc = (int)div_magic((uint32_t)ul, per_col_magic), bxi = lc * 8 + xcd;
    if (!(ul < chunk && bxi < g.nbx)) return;
    const int rem = ul - lc * per_col;
    const int byi = (int)div_magic((uint32_t)rem, ngroups_magic);
    const int gr = rem - byi * ngroups + byi;   // k_me16b: why
    const int grp = gr - ngroups * (int)div_magic((uint32_t)gr, ngroups_magic);
    const int blk = byi * g.nbx + bxi;
    int f0, nf;
    bits_group(n_frames, ngroups, ngroups_magic, grp, f0, nf);
    const int x = bxi * BS, y = byi * BS;
    const int i_min = max(y - WIN, 0), i_max = min(y + WIN, g.H);
    const int j_min = max(x - WIN, 0), j_max = min(x + WIN, g.W);
    const int ri = i_max - BS - i_min, rj = j_max - BS - j_min;
    const int nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
    const int ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
    const int ncand = nci * ncj;
    const uint32_t inv_ncj = ncj ? 65536u / (uint32_t)ncj + 1u : 0u;   // cand / ncj for cand < 64, ncj <= 8

    // per frame: static?  Lane f reads frame f's record: one round trip (NF scalar loads, each behind a branch,
    // were NF in a row at the start of every wave; k_me16b: -5.7 % from the same change)
    uint32_t skip;   // bit f: nothing to search for frame f0 + f
    {
        const uint32_t sd = binfo[(size_t)(f0 + max(min(lane, nf - 1), 0)) * g.N + blk].x;
        const bool st = lane >= nf || (!g.never_static && sd <= g.thresh);
        skip = (uint32_t)__ballot(st) & (NF >= 32 ? 0xFFFFFFFFu : (1u << NF) - 1u);
    }
    const uint32_t all = NF >= 32 ? 0xFFFFFFFFu : (1u << NF) - 1u;
    uint32_t acc[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) acc[f] = 0;
    const bool act = lane < ncand;
    const uint32_t *__restrict__ cblk = cpl + ((size_t)f0 * g.N + blk) * kBits8Blk;
    const size_t cfs = (size_t)g.N * kBits8Blk;   // dwords per frame of cpl
    if (skip != all && ncand > 0) {
        // the wave's planes (nf x 192 bytes) -> L2, for the scalar loads below
        {
            const __amdgpu_buffer_rsrc_t r_c = buf_rsrc(cblk, 0xFFFFFFFFu);
#pragma unroll
            for (int i = 0; i < (NF * 12 + 63) / 64; i++) {
                const int e = lane + 64 * i, fr = min(e / 12, nf - 1), piece = e - (e / 12) * 12;
                if (e < NF * 12)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        r_c, (__attribute__((address_space(3))) void *)(s_dump + wave * (NF * 256)), 16,
                        (int)((uint32_t)fr * (uint32_t)(cfs * 4) + 16u * (uint32_t)piece), 0, 0, 0);
            }
        }
        const int cc = act ? lane : 0;
        const int ci = (int)(((uint32_t)cc * inv_ncj) >> 16), cj = cc - ci * ncj;
        const size_t wstride = (size_t)g.nbx * kBits8WA;   // dwords per picture row of rtab
        const uint32_t *__restrict__ rp = rtab + (size_t)(i_min + STEP * ci) * wstride + (size_t)bxi * kBits8WA + cj * 4;
        uint32_t r0[kBitsT], r1[kBitsT];
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const uint4 v = *(const uint4 *)(rp + q * (kBits8M * 4));
            r0[4 * q] = v.x; r0[4 * q + 1] = v.y; r0[4 * q + 2] = v.z; r0[4 * q + 3] = v.w;
            const uint4 w4 = *(const uint4 *)(rp + 4 * wstride + q * (kBits8M * 4));
            r1[4 * q] = w4.x; r1[4 * q + 1] = w4.y; r1[4 * q + 2] = w4.z; r1[4 * q + 3] = w4.w;
        }
#pragma unroll
        for (int f = 0; f < NF; f++) {
            if (!((skip >> f) & 1u)) {   // scalar branch
#ifdef VCS_EXP_BITS_SAMEC   // experiment builds only (results are wrong): every plane load hits the scalar cache
                const uint32_t *__restrict__ c = cblk + (f & 1) * kBits8Blk;
#else
                const uint32_t *__restrict__ c = cblk + (size_t)f * cfs;
#endif
                uint32_t c0[kBitsT], c1[kBitsT];
#pragma unroll
                for (int t = 0; t < kBitsT; t++) { c0[t] = c[t]; c1[t] = c[kBitsT + t]; }
                acc[f] = bits_count_lt(r0, c0, acc[f]);
                acc[f] = bits_count_lt(r1, c1, acc[f]);
            }
        }
    }
    // ---- arg-min per frame (motion.py:149: strict <, raster order = lowest index on ties) -------
    // key = (cost << 6) | candidate, cost = sum(r) - sum(c) + 256 * #{r < c} without the -sum(c), which is the
    // same for all candidates; a lane without a candidate has compared candidate 0's chunks: candidate 0's cost
    // with a higher index, which never wins.  Through LDS (k_me16b: why): the 64 keys of frame f go to row f,
    // the 64 / NF lanes of frame f take 16 keys each, DPP finishes inside them.
    const uint32_t base = (sums[(size_t)blk * kMe8SumStride + (act ? lane : 0)] << 6) | (uint32_t)lane;
    constexpr int LPF = 64 / NF, KPL = 64 / LPF;   // lanes per frame, keys per lane
    uint32_t my_key = 0xFFFFFFFFu;
    const bool searching = skip != all && ncand > 0;
    if (searching) {
        uint32_t *s_red = (uint32_t *)(s_dump + wave * (NF * 256));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-direct loads into this area have landed
        wave_lds_fence();
#pragma unroll
        for (int f = 0; f < NF; f++) s_red[f * 64 + lane] = base + (acc[f] << 14);
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < KPL / 4; q++) {
            const uint4 v = *(const uint4 *)(s_red + lane * KPL + 4 * q);
            my_key = min(my_key, min(min(v.x, v.y), min(v.z, v.w)));
        }
        if (LPF >= 2) my_key = min(my_key, dpp_u32<0xB1>(my_key));    // quad_perm:[1,0,3,2]
        if (LPF >= 4) my_key = min(my_key, dpp_u32<0x4E>(my_key));    // quad_perm:[2,3,0,1]
        if (LPF >= 8) my_key = min(my_key, dpp_u32<0x141>(my_key));   // row_half_mirror
    }
    const int fr = lane / LPF;
    if (lane % LPF == 0 && fr < nf) {
        const size_t o = (size_t)(f0 + fr) * g.N + blk;
        const bool searched = !((skip >> fr) & 1u), found = searched && searching;
        uint32_t mvw = 0u;                         // static block: [0, 0]
        if (searched) {
            int bx = 0, byy = 0;                   // motion.py:120: no candidate
            if (found) {
                const int cb = (int)(my_key & 63u);
                const int cbi = (int)(((uint32_t)cb * inv_ncj) >> 16);
                byy = i_min + cbi * STEP;
                bx = j_min + (cb - cbi * ncj) * STEP;
            }
            mvw = (uint32_t)(uint16_t)(bx - x) | ((uint32_t)(uint16_t)(byy - y) << 16);  // [dx, dy]
        }
        *(uint32_t *)(mv + 2 * o) = mvw;
        if (cost_out) {   // a debugging output: only then is the record read again
            const uint2 bi = binfo[o];
            cost_out[o] = !searched ? (unsigned long long)bi.x : found ? (unsigned long long)((my_key >> 6) - bi.y) : 9999999999ull;
        }
    }
}

// ---- host side ----------------------------------------------------------------------------
inline size_t me_bits8_table_words(const Geom &g) { return (size_t)g.H * g.nbx * kBits8WA; }
// the chunk table + the chunk sums behind it
inline size_t me_bits8_rtab_bytes(const Geom &g) {
    return me_bits8_table_words(g) * 4 + (size_t)g.H * g.nbx * kBits8M * 4 + 256;
}
inline size_t me_bits8_cpl_bytes(const Geom &g, int n) { return (size_t)n * g.N * kBits8Blk * 4; }
inline int me_bits8_cur_wgs(const Geom &g) {   // workgroups of k_bits_cur8 per frame (a multiple of 8)
    return 8 * (((g.N + kBits8CurBlocks - 1) / kBits8CurBlocks + 7) / 8);
}
inline size_t me_bits8_binfo_bytes(const Geom &g, int n) {
    return (size_t)n * g.N * sizeof(uint2) + (size_t)n * me_bits8_cur_wgs(g) * sizeof(uint32_t);
}
// VCS_ME8_BITS: 0 = off (k_me8g / k_me8), 1 = default (batches of kBits8MinFrames frames or more),
// 2 = every launch (tests)
inline int me_bits8_mode() {
    static const int v = [] {
        const char *e = getenv("VCS_ME8_BITS");
        const int m = e ? atoi(e) : 1;
        return m >= 0 && m <= 2 ? m : 1;
    }();
    return v;
}
// Measured on MI355X, 1080p, whole step incl. k_bits_cur8 and the per-reference tables
// (tools/gpu_bits8_ab.sh), bit-plane search vs k_me8g / k_me8: n = 1: 0.124 vs 0.098 ms; n = 2: 0.140
// vs 0.140; n = 3: 0.156 vs 0.181; n = 4: 0.177 vs 0.218; n = 8: 0.253 vs 0.397; n = 29: 0.63 vs 0.87
// (search 0.197 vs 0.54).  Frames per wave 8 / 16 / 32: search 0.218 / 0.197 / 0.218.
constexpr int kBits8MinFrames = 3;
inline bool me_bits8_worthwhile(const Geom &g, int n) {
    const int m = me_bits8_mode();
    if (m == 0 || !me_fast8_supported(g)) return false;
    if ((long long)g.N * n * kBits8Blk > 0x7FFFFFFFll * 4 || me_bits8_rtab_bytes(g) > ((size_t)1 << 30)) return false;
    {   // div_magic in the unit decode is exact while ceil(columns / 8) * (nby * groups)^2 < 2^32 (see me_bits_worthwhile)
        const unsigned long long per_col = (unsigned long long)g.nby * (unsigned long long)((n + kBits8NF - 1) / kBits8NF);
        if ((unsigned long long)((g.nbx + 7) / 8) * per_col * per_col >= (1ull << 32)) return false;
    }
    return m == 2 || n >= kBits8MinFrames;
}
inline int me_bits8_groups(int n) { return (n + kBits8NF - 1) / kBits8NF; }

inline void launch_bits_ref8(const Geom &g, hipStream_t stream, const uint8_t *ref, size_t ref_bytes, uint32_t *rtab,
                             uint32_t *r2 = nullptr) {
    const uint32_t total = (uint32_t)g.H * (uint32_t)g.nbx * kBits8M;
    hipLaunchKernelGGL(k_bits_ref8, dim3((total + 255) / 256), dim3(256), 0, stream, ref, (uint32_t)ref_bytes, g, rtab, r2);
}
inline void launch_bits_cur8(const Geom &g, hipStream_t stream, const uint8_t *curs, size_t stride, const uint8_t *ref,
                             int n, uint32_t *cpl, uint2 *binfo, uint32_t *ns_part, const uint32_t *r2 = nullptr,
                             uint32_t *sums = nullptr, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    hipExtLaunchKernelGGL(k_bits_cur8, dim3(me_bits8_cur_wgs(g) * n), dim3(256), 0, stream, ev_start, ev_stop, 0, curs,
                          stride, ref, g, n, cpl, binfo, ns_part, r2, sums);
}
inline void launch_bits_search8(const Geom &g, hipStream_t stream, const uint32_t *rtab, const uint32_t *cpl,
                                const uint2 *binfo, const uint32_t *sums, int n, int16_t *mv, const uint32_t *ns_part,
                                uint32_t *ns, unsigned long long *cost, const uint8_t *pf_base = nullptr,
                                size_t pf_bytes = 0, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    const int ngroups = me_bits8_groups(n);
    const int chunk = (((g.nbx + 7) / 8) * g.nby * ngroups + 3) & ~3;   // units per XCD (k_me8b)
    const int nsearch = 8 * (chunk / 4);
    static const int npf_env = [] { const char *e = getenv("VCS_BITS_PF_WGS"); return e ? atoi(e) & ~7 : 0; }();
    const int npf = pf_base ? (npf_env > 0 ? npf_env : 384) : 0;   // prefetch workgroups behind the search's (k_me16b: why last; 128 / 256 / 384 / 512: step 0.632 / 0.622 / 0.6185 / 0.621 ms)
    hipExtLaunchKernelGGL((k_me8b<kBits8NF>), dim3(nsearch + npf), dim3(256), 0, stream, ev_start, ev_stop, 0, rtab, cpl,
                          binfo, sums, g, n, ngroups, mv, cost, pf_base, (uint32_t)(pf_base ? pf_bytes >> 4 : 0), nsearch,
                          ns_part, me_bits8_cur_wgs(g), ns, magic_u32((uint32_t)(g.nby * ngroups)), magic_u32((uint32_t)ngroups));
}

}  // namespace vcs
