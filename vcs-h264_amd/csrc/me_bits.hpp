// me_bits.hpp -- k_me16b: the 16x16 / +-32 / stride-5 motion search on BIT PLANES.
//
// motion.py:146 scores a candidate with sum((ref - cur) mod 256) over 768 bytes, which is
// sum(r) - sum(c) + 256 * #{r < c}.  sum(r) comes from a per-reference table,
// sum(c) is one number per block, so the work is counting byte pairs with r < c: 76,800
// pairs per block.  k_me16g counts them four at a time (v_lerp_u8 -> v_and -> v_bcnt, 3
// vector instructions per 4 pairs).  Here the bytes are stored as bit planes, 32 pairs per
// dword and plane, and [r < c] is the borrow of c - r ... the carry chain of an 8-bit
// compare run from plane 0 to plane 7:   b' = majority(~r_k, c_k, b).
// gfx950 has v_bitop3_b32 (any 3-input boolean function in one instruction), so a chain is
// 8 instructions + one v_bcnt per 32 pairs: 9 / 32 instead of 24 / 32.
//
// Operand routing (lane = candidate, as in k_me16):
//   * r, the window: every (row pair, candidate column) position of the reference is
//     pre-cut into a 96-byte chunk -- two 48-byte rows = 3 dwords x 8 planes -- by k_bits_ref,
//     once per reference frame ("rtab").  A lane loads its candidate's chunk (6 x 16 bytes,
//     contiguous across the 10 candidate columns) and keeps it in 24 registers ...
//   * ... while it loops over the FRAMES of the GOP: all P frames search the same I frame
//     (encoder.py:51-52), so a wave owns one block position and up to NF frames; the window
//     operand is loaded once per NF frames (double-buffered across row pairs);
//   * c, the current block: the same for every lane of the wave (one block, one frame, one
//     row pair) -> scalar registers.  k_bits_cur writes every block of every P frame as 8 row
//     pairs x 24 plane dwords ("cpl"), the search reads them with s_load, two frames per
//     wait; v_bitop3_b32 takes the plane from an SGPR.  k_bits_cur also does the static test
//     (motion.py:109-116), sum(c) and -- from k_bits_ref's row-pair sums -- the sum(r) table;
//   * a scalar load that misses L2 takes ~2,000 cycles: one LDS-direct load per row pair,
//     two row pairs ahead, brings the wave's planes into L2 for the scalar loads AND into
//     LDS for the second pass;
//   * candidates 64..99 of an interior block: lane = (candidate, row pair), 288 tasks = 4.5
//     wave passes, chunk per lane, planes per lane from the LDS copy, counts meet in LDS.
// No barrier; a static block costs nothing in the search (a scalar branch over its frame),
// divergence-free.  The launch's last workgroups stream the P frames into the Infinity
// Cache for the tail kernel.  DESIGN.md 5 has the measurements and what was tried.
// Results are identical to k_me16 / k_me16g / k_me_generic (tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_ext.h>

#include "me_fast.hpp"

namespace vcs {

constexpr int kBitsT = 24;                    // dwords per row-pair chunk: t = 8 * d + k (d: dword of the 96-byte string, k: plane)
constexpr int kBitsBlk = 8 * kBitsT;          // dwords per 16x16x3 block (8 row pairs)
constexpr int kBitsM = 10;                    // candidate columns per block column
constexpr int kBitsWA = 6 * kBitsM * 4;       // dwords per (window row, block column): [q = t / 4][m][4]

// 8 dwords = 32 bytes -> 8 plane dwords, in place: x[k] bit (8 t + j) = bit k of byte t of the
// old x[j].  Four 8x8 bit-matrix transposes at once (one per byte column), three butterfly
// stages.  The order of the 32 bits inside a plane dword is arbitrary but the same for the
// reference chunks and the current blocks, which is all the comparison needs.
__device__ __forceinline__ void bits_transpose8(uint32_t (&x)[8]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t t = ((x[j] >> 4) ^ x[j + 4]) & 0x0F0F0F0Fu;
        x[j + 4] ^= t;
        x[j] ^= t << 4;
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int j = 4 * h; j < 4 * h + 2; j++) {
            const uint32_t t = ((x[j] >> 2) ^ x[j + 2]) & 0x33333333u;
            x[j + 2] ^= t;
            x[j] ^= t << 2;
        }
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const uint32_t t = ((x[j] >> 1) ^ x[j + 1]) & 0x55555555u;
        x[j + 1] ^= t;
        x[j] ^= t << 1;
    }
}

// ---- reference chunks ---------------------------------------------------------------------
// rtab[((w * na + (a - a0)) * 6 + q) * 40 + m * 4 + e] = plane dword t = 4 q + e of the 96-byte string
// (row w, bytes 3 j .. 3 j + 47) ++ (row w + 1, same bytes), j = max(16 a - 32, 0) + 5 m: the
// m-th candidate column of block column a (motion.py:125-140).  Every w: a candidate's rows
// start at i_min + 5 ci, of either parity.  One thread per (w, a, m).
// The table covers a BAND of block columns [a0, a0 + na) (the whole picture when it fits the
// Infinity Cache beside the planes; 4K goes in four bands): rtab's block-column index is a - a0.
// r2 (nullable): r2[(w * na + (a - a0)) * 10 + m] = byte sum of the chunk's 96 bytes (rows w, w + 1): eight
// of them are a candidate's sum(r) (k_bits_cur adds them up: the sum(r) table without k_me16_sums).
__global__ __launch_bounds__(256) void k_bits_ref(const uint8_t *__restrict__ ref, uint32_t ref_bytes, Geom g,
                                                  uint32_t *__restrict__ rtab, int a0, int na,
                                                  uint32_t *__restrict__ r2) {
    const uint32_t total = (uint32_t)g.H * (uint32_t)na * kBitsM;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total) return;
    const uint32_t m = idx % kBitsM, wa = idx / kBitsM;
    const uint32_t a = (uint32_t)a0 + wa % (uint32_t)na, w = wa / (uint32_t)na;
    const int j = max(16 * (int)a - 32, 0) + 5 * (int)m;
    const __amdgpu_buffer_rsrc_t r_ref = buf_rsrc(ref, ref_bytes);   // reads past the slack rows return 0
    const uint32_t off = w * (uint32_t)g.pitch + 3u * (uint32_t)j;
    uint32_t s[24];
#pragma unroll
    for (int i = 0; i < 12; i++) {   // unaligned dwords (tools/buf_test.hip)
        s[i] = buf_ld32(r_ref, off + 4u * i, 0);
        s[12 + i] = buf_ld32(r_ref, off + (uint32_t)g.pitch + 4u * i, 0);
    }
    if (r2) {
        uint32_t rs = 0;
#pragma unroll
        for (int i = 0; i < 24; i++) rs = __builtin_amdgcn_sad_u8(s[i], 0u, rs);
        r2[idx] = rs;
    }
    uint4 *dst = (uint4 *)(rtab + (size_t)wa * kBitsWA + m * 4);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = s[8 * d + i];
        bits_transpose8(x);
        dst[(2 * d) * kBitsM] = make_uint4(x[0], x[1], x[2], x[3]);
        dst[(2 * d + 1) * kBitsM] = make_uint4(x[4], x[5], x[6], x[7]);
    }
}

// ---- current blocks -----------------------------------------------------------------------
// cpl[((f * N + blk) * 8 + p) * 24 + 8 d + k]: row pair p of block blk of frame f as plane
// dwords; binfo[f * N + blk] = {sum(max(ref - cur, 0)) over the co-located block, sum(cur)}
// (motion.py:109-116, the static test, and the -sum(c) of the cost); ns_part[f][workgroup] =
// static blocks.  One thread per (block, row pair): its 96-byte string (2 rows x 48 bytes) as six
// 16-byte pieces, the co-located reference pieces for the static test, three 32-byte
// transposes, 96 contiguous bytes out (a workgroup = 32 blocks = 24 KB of planes in one
// run).  The block sums meet in the 8 lanes of a block by DPP.
#ifndef VCS_BITS_CUR_BLOCKS
#define VCS_BITS_CUR_BLOCKS 32
#endif
constexpr int kBitsCurBlocks = VCS_BITS_CUR_BLOCKS;     // blocks per workgroup (8 threads each): 16 / 32 / 64 / 128 = 0.097 / 0.095 / 0.097 / 0.103 ms
constexpr int kBitsCurThreads = kBitsCurBlocks * 8;
#ifndef VCS_BITS_CUR_FPT
#define VCS_BITS_CUR_FPT 4
#endif
// frames per thread: a thread keeps its co-located reference pieces (and their byte sum) in registers for
// kBitsCurFpt frames of the batch -- read per frame they are L2 hits, but half of the kernel's loads
// (0.022 of its 0.064 ms at 29 x 1080p, VCS_EXP_BITS_NOREF)
constexpr int kBitsCurFpt = VCS_BITS_CUR_FPT;
__global__ __launch_bounds__(kBitsCurThreads) void k_bits_cur(const uint8_t *__restrict__ curs, size_t cur_stride,
                                                  const uint8_t *__restrict__ ref, Geom g, int n_frames,
                                                  uint32_t *__restrict__ cpl, uint2 *__restrict__ binfo,
                                                  uint32_t *__restrict__ ns_part, const uint32_t *__restrict__ r2,
                                                  uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_ns[kBitsCurFpt];
    __shared__ uint4 s_out[kBitsCurThreads / 64 * 384];
    const int tid = threadIdx.x, p = tid & 7;
    // workgroups go round-robin to the 8 XCDs: XCD x takes the x-th eighth of every frame's blocks,
    // frame group after frame group, so that its band of the reference (0.8 MB at 1080p) stays in its L2
    // while the frames stream through
    const int nfg = (n_frames + kBitsCurFpt - 1) / kBitsCurFpt;   // frame groups
    const int per = (int)(gridDim.x >> 3) / nfg;                   // workgroups per XCD and frame group
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int fg = idx / per, f_first = fg * kBitsCurFpt, nf = min(kBitsCurFpt, n_frames - f_first);
    const int wg_in_frame = xcd * per + (idx - fg * per);
    const int blockIdx_blk0 = wg_in_frame * kBitsCurBlocks;
    const int blk = blockIdx_blk0 + (tid >> 3);
    if (tid < kBitsCurFpt) s_ns[tid] = 0;
    __syncthreads();
    const uint32_t fbytes = (uint32_t)(g.H * g.pitch);
    const __amdgpu_buffer_rsrc_t r_ref = buf_rsrc(ref, fbytes);
    const int blkc = min(blk, g.N - 1);
    const int byi = blkc / g.nbx, bxi = blkc - byi * g.nbx;
    const uint32_t o = (uint32_t)((byi * 16 + 2 * p) * g.pitch + 48 * bxi);
    uint32_t oi[6];
#pragma unroll
    for (int i = 0; i < 6; i++) oi[i] = o + (i >= 3 ? (uint32_t)g.pitch + 16u * (i - 3) : 16u * i);
    uint4 vr[6], vc[6];
#pragma unroll
    for (int i = 0; i < 6; i++) vr[i] = buf_ld128(r_ref, oi[i], 0);
    {
        const __amdgpu_buffer_rsrc_t r_cur = buf_rsrc(curs + (size_t)f_first * cur_stride, fbytes);
#pragma unroll
        for (int i = 0; i < 6; i++) vc[i] = buf_ld128(r_cur, oi[i], 0);
    }
    uint32_t sr = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        sr = __builtin_amdgcn_sad_u8(vr[i].x, 0u, sr); sr = __builtin_amdgcn_sad_u8(vr[i].y, 0u, sr);
        sr = __builtin_amdgcn_sad_u8(vr[i].z, 0u, sr); sr = __builtin_amdgcn_sad_u8(vr[i].w, 0u, sr);
    }
    uint4 *const s_w = s_out + (tid >> 6) * 384;
    const int lane = tid & 63;
    const bool head = p == 0 && blk < g.N;
#pragma unroll 1
    for (int ff = 0; ff < nf; ff++) {
        const int f = f_first + ff;
        uint4 vn[6];   // the next frame's pieces, requested before this frame's are worked on
        if (ff + 1 < nf) {
            const __amdgpu_buffer_rsrc_t r_nxt = buf_rsrc(curs + (size_t)(f + 1) * cur_stride, fbytes);
#pragma unroll
            for (int i = 0; i < 6; i++) vn[i] = buf_ld128(r_nxt, oi[i], 0);
        }
        uint32_t s[24], sad = 0, sc = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            s[4 * i] = vc[i].x; s[4 * i + 1] = vc[i].y; s[4 * i + 2] = vc[i].z; s[4 * i + 3] = vc[i].w;
            const uint32_t r4[4] = {vr[i].x, vr[i].y, vr[i].z, vr[i].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                sad = __builtin_amdgcn_sad_u8(r4[e], s[4 * i + e], sad);
                sc = __builtin_amdgcn_sad_u8(s[4 * i + e], 0u, sc);
            }
        }
        // sum(max(r - c, 0)) = (sum|r - c| + sum r - sum c) / 2, over the 8 row pairs of the block
        uint32_t sd2 = sad + sr - sc;
        sd2 += dpp_u32<0xB1>(sd2); sd2 += dpp_u32<0x4E>(sd2); sd2 += dpp_u32<0x141>(sd2);
        sc += dpp_u32<0xB1>(sc); sc += dpp_u32<0x4E>(sc); sc += dpp_u32<0x141>(sc);
        const uint32_t sdiff = sd2 >> 1;
        // a thread's 96 bytes are contiguous in cpl and so are the wave's 6 KB; they leave through LDS so
        // that one store instruction writes 1 KB in a row (16-byte pieces 96 bytes apart are partial-line
        // writes: 0.097 -> 0.085 ms)
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
            const int wblk0 = blockIdx_blk0 + (tid >> 6) * 8;   // first block of this wave
            uint4 *dst = (uint4 *)(cpl + ((size_t)f * g.N + wblk0) * kBitsBlk);
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int e = i * 64 + lane;                      // 16-byte piece of the wave's 8 blocks x 48 pieces
                // (non-temporal: the planes are read back once, 0.1 ms later, by kernels with bandwidth to
                // spare; step -0.9 % in 4 of 4 A/B pairs)
                if (wblk0 + e / 48 < g.N) __builtin_nontemporal_store(*(const u32x4_t *)&s_w[e], (u32x4_t *)&dst[e]);
            }
        }
        wave_lds_fence();   // the next frame rewrites the wave's staging area
        if (head) binfo[(size_t)f * g.N + blk] = make_uint2(sdiff, sc);
        const unsigned long long st = __ballot(head && !g.never_static && sdiff <= g.thresh);
        if ((tid & 63) == 0 && st) atomicAdd(&s_ns[ff], (uint32_t)__builtin_popcountll(st));
        if (ff + 1 < nf) {
#pragma unroll
            for (int i = 0; i < 6; i++) vc[i] = vn[i];
        }
    }
    __syncthreads();
    // static blocks of this workgroup per frame; k_me16b adds the 8 * per counts of a frame up (no counter
    // to zero before the launch, no atomics on it)
    if (tid < nf) ns_part[(size_t)(f_first + tid) * (8 * per) + wg_in_frame] = s_ns[tid];
    // The sum(r) table of a new reference (what k_me16_sums makes: sums[blk * 128 + cand] = byte sum of
    // the candidate's 16 x 48 bytes, 0 for slots without a candidate), by the workgroups of the first
    // frame: a candidate's sum = 8 row-pair sums of k_bits_ref (r2 != null: the table is due).
#ifdef VCS_EXP_BITS_NOFOLD   // experiment builds only (results are wrong): what the folded sum(r) table costs
    if (false) {
#else
    if (r2 && fg == 0) {
#endif
        constexpr int BS = 16, WIN = 32, STEP = 5;
        const size_t rw = (size_t)g.nbx * kBitsM;   // r2 words per picture row
#pragma unroll 1
        for (int it = 0; it < kBitsCurBlocks * kMe16SumStride / kBitsCurThreads; it++) {
            const int e = tid + kBitsCurThreads * it, bi = e / kMe16SumStride, cand = e - bi * kMe16SumStride;
            const int b = blockIdx_blk0 + bi;
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
                const uint32_t *q = r2 + (size_t)(i_min + STEP * ci) * rw + (size_t)bx * kBitsM + cj;
#pragma unroll
                for (int p2 = 0; p2 < 8; p2++) v += q[(size_t)(2 * p2) * rw];
            }
            sums[(size_t)b * kMe16SumStride + cand] = v;
        }
    }
}

// ---- the search ---------------------------------------------------------------------------
// One comparison step of the borrow chain on 32 pairs: r < c so far (b), planes r, c:
// b' = majority(c, b, ~r) -- r != c: c decides (c = 1, r = 0: r < c); r == c: keep b.
// v_bitop3_b32 with truth table 0xD4 (bit index = src0 * 4 + src1 * 2 + src2).
__device__ __forceinline__ uint32_t bits_lt_step(uint32_t r, uint32_t c, uint32_t b) {
    return __builtin_amdgcn_bitop3_b32(c, b, r, 0xD4);
}
// #{r < c} over the 96 byte pairs of one chunk: 3 dwords x 8 planes, t = 8 d + k; added to n
__device__ __forceinline__ uint32_t bits_count_lt(const uint32_t (&r)[kBitsT], const uint32_t (&c)[kBitsT], uint32_t n) {
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t b = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) b = bits_lt_step(r[8 * d + k], c[8 * d + k], b);
        n = (uint32_t)__builtin_popcount(b) + n;   // one v_bcnt_u32_b32 (x, acc) ...
        asm volatile("" : "+v"(n));                // ... not v_bcnt(x, 0) + v_add3: the chain stays
    }
    return n;
}
// the same with the first step of the first chain already taken (b0): the search kernel puts its
// next loads between that step and the rest
__device__ __forceinline__ uint32_t bits_count_lt_from(const uint32_t (&r)[kBitsT], const uint32_t (&c)[kBitsT],
                                                      uint32_t b0, uint32_t n) {
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t b = d == 0 ? b0 : 0u;
#pragma unroll
        for (int k = d == 0 ? 1 : 0; k < 8; k++) b = bits_lt_step(r[8 * d + k], c[8 * d + k], b);
        n = (uint32_t)__builtin_popcount(b) + n;
        asm volatile("" : "+v"(n));
    }
    return n;
}
__device__ __forceinline__ void bits_ld_chunk(uint32_t (&r)[kBitsT], const uint32_t *__restrict__ rp) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const uint4 v = *(const uint4 *)(rp + q * (kBitsM * 4));
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
}

// frames [f0, f0 + nf) of group `grp` when n frames are dealt to `ngroups` waves as evenly as possible
// (dealing them in pairs -- 29 -> 8, 8, 8, 5, two pair comparisons fewer per candidate than 8, 7, 7, 7
// -- measures the same: a workgroup lives as long as its longest wave either way)
__device__ __forceinline__ void bits_group(int n, int ngroups, uint32_t ngroups_magic, int grp, int &f0, int &nf) {
    const int q = (int)div_magic((uint32_t)n, ngroups_magic), r = n - q * ngroups;
    f0 = grp * q + min(grp, r);
    nf = q + (grp < r ? 1 : 0);
}

// LDS of one wave: the planes of its NF current blocks, [row pair][frame][96 bytes] (filled by
// LDS-direct loads, one per row pair), and the second pass's counts, [frame pair][40] (two 16-bit
// counts per word: a count is at most 768) -- 6,784 bytes at NF = 8: six workgroups per CU
template <int NF>
struct BitsLds {
    static constexpr int kPlanes = 8 * NF * 96, kPart = ((NF + 1) / 2) * 40 * 4, kWave = kPlanes + kPart;
};

#ifndef VCS_BITS_WPW
#define VCS_BITS_WPW 4
#endif
constexpr int kBitsWpw = VCS_BITS_WPW;   // waves (units) per k_me16b workgroup: 4 = the frame groups of one block position when ngroups = 4
                                         // (2 / 4 / 8: search 0.243 / 0.243 / 0.269 ms -- two positions per workgroup live too long)
template <int NF>
__global__ __launch_bounds__(64 * kBitsWpw) void k_me16b(const uint32_t *__restrict__ rtab, const uint32_t *__restrict__ cpl,
                                               const uint2 *__restrict__ binfo, const uint32_t *__restrict__ sums,
                                               Geom g, int n_frames, int ngroups, int16_t *__restrict__ mv,
                                               unsigned long long *__restrict__ cost_out,
                                               unsigned long long *__restrict__ clk_out,
                                               const uint8_t *__restrict__ pf_base, uint32_t pf_total16,
                                               int n_search_wgs, int pf_octets, int pf_ratio,
                                               const uint32_t *__restrict__ ns_part, int ns_per_frame,
                                               uint32_t *__restrict__ n_static, int a0, int na,
                                               uint32_t per_col_magic, uint32_t ngroups_magic) {
    // (per_col_magic, ngroups_magic: magic_u32 of nby * ngroups and of ngroups -- the unit decode below
    // is scalar, and a scalar division by a runtime value is a 25-instruction vector sequence)
    constexpr int BS = 16, WIN = 32, STEP = 5;
    using L = BitsLds<NF>;
#ifndef VCS_EXP_BITS_LDSPAD
#define VCS_EXP_BITS_LDSPAD 0   // experiment builds: extra LDS per workgroup (fewer workgroups per CU)
#endif
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[kBitsWpw * L::kWave + VCS_EXP_BITS_LDSPAD];
    // clk_out (timing on): shader-clock cycles and 100 MHz ticks this workgroup lived, summed over
    // the workgroups: their quotient is the clock the chip held under this load
    const unsigned long long clk0 = clk_out ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long rt0 = clk_out ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Infinity Cache prefetch for the tail kernel (vcs_api.hip, k_l3_prefetch, explains why).  The
    // workgroups behind the search's own are dispatched last, i.e. while the last generation of
    // search workgroups drains and SIMDs fall idle: they stream the launch's P frames into LDS
    // (LDS-direct, no register, no consumer) right before the tail wants them.  Earlier in the
    // launch the lines would be gone again (the search moves 300 MB of planes and chunks).
    // grid layout in octets (8 workgroups, one per XCD): `head` search octets, then pf_octets times
    // [pf_ratio search octets, 1 prefetch octet]
    int oct = (int)(blockIdx.x >> 3);
    bool is_pf = false;
    if (pf_octets) {
        const int head = n_search_wgs / 8 - pf_ratio * pf_octets;
        if (oct >= head) {
            const int o2 = oct - head, grp = o2 / (pf_ratio + 1), r = o2 - grp * (pf_ratio + 1);
            is_pf = r == pf_ratio;
            oct = is_pf ? grp : head + grp * pf_ratio + r;
        }
    }
    if (is_pf) {
        const uint32_t nw = (uint32_t)pf_octets * 8u * (uint32_t)kBitsWpw;
        const uint32_t w = ((uint32_t)oct * 8u + (blockIdx.x & 7u)) * (uint32_t)kBitsWpw + (uint32_t)wave;
        const __amdgpu_buffer_rsrc_t r_pf = buf_rsrc(pf_base, pf_total16 << 4);
        uint32_t it = 0;
        for (uint32_t i = w * 64u; i < pf_total16; i += nw * 64u, it++) {   // uniform trip count
            if (i + (uint32_t)lane < pf_total16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    r_pf, (__attribute__((address_space(3))) void *)(s_lds + wave * 1024), 16,
                    (int)((i + (uint32_t)lane) << 4), 0, 0, 0);
            if ((it & 7u) == 7u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // at most 8 KB in flight per wave
        }
        return;
    }
    uint8_t *const s_c = s_lds + wave * L::kWave;
    uint32_t *const s_part = (uint32_t *)(s_c + L::kPlanes);
    // a workgroup = 4 consecutive units (the 4 frame groups of one block position when ngroups = 4:
    // they read the same chunks through one L1)
    // units = (block position, frame group); XCD x takes the block columns a0 + x, a0 + x + 8, ... and
    // walks down each of them (the chunk rows of vertically adjacent blocks overlap: they meet in the
    // XCD's L2; nothing is shared between columns, so interleaving the columns costs no reuse and
    // keeps the XCDs equally busy when a region of the picture is static)
    const int per_col = g.nby * ngroups;
    const int chunk = (((na + 7) / 8) * per_col + kBitsWpw - 1) / kBitsWpw * kBitsWpw;   // units per XCD (the XCDs with one column less idle at the end)
    if (blockIdx.x == 0 && a0 == 0) {   // static blocks per frame: the sums of k_bits_cur's per-workgroup counts
        for (int f = wave; f < n_frames; f += kBitsWpw) {
            uint32_t v = 0;
            for (int i = lane; i < ns_per_frame; i += 64) v += ns_part[(size_t)f * ns_per_frame + i];
            v = wave_sum_u32(v);
            if (lane == 0) n_static[f] = v;
        }
    }
    const int xcd = blockIdx.x & 7, wg = oct;
    const int ul = wg * kBitsWpw + wave;
    const int lc = (int)div_magic((uint32_t)ul, per_col_magic), bl = lc * 8 + xcd;   // bl: column inside the band
    if (ul < chunk && bl < na) {
        const int rem = ul - lc * per_col;
        const int byi = (int)div_magic((uint32_t)rem, ngroups_magic), bxi = a0 + bl;
        // the groups of a position differ in size (29 frames: 8, 7, 7, 7): rotated from position to
        // position, so that the same wave slot (SIMD) does not get the large one every time (-0.4 % of
        // the step in 3 of 3 A/B pairs)
        const int gr = rem - byi * ngroups + byi;
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
        const uint32_t inv_ncj = ncj ? 65536u / (uint32_t)ncj + 1u : 0u;   // cand / ncj for cand < 128, ncj <= 10

        // per frame: static? (scalar; the records are read again at the end rather than kept: 16
        // SGPRs the loops below would otherwise spill around)
        uint32_t skip = 0;   // bit f: nothing to search for frame f0 + f
#pragma unroll
        for (int f = 0; f < NF; f++) {
            if (f < nf) {
                const uint32_t sd = binfo[(size_t)(f0 + f) * g.N + blk].x;
                if (!g.never_static && sd <= g.thresh) skip |= 1u << f;
            } else {
                skip |= 1u << f;
            }
        }

        uint32_t best[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) best[f] = 0xFFFFFFFFu;

        const size_t wstride = (size_t)na * kBitsWA;      // dwords per window row of rtab
        const uint32_t *__restrict__ cblk = cpl + ((size_t)f0 * g.N + blk) * kBitsBlk;
        const size_t cfs = (size_t)g.N * kBitsBlk;         // dwords per frame of cpl

        if (skip != (1u << NF) - 1u && ncand > 0) {
            // The planes of the wave's NF current blocks arrive by LDS-direct loads, one instruction
            // per row pair (lane = (frame, 16-byte piece)), two rounds ahead of their use.  They are
            // read twice: by the first pass through s_load from L2 -- a scalar load that misses L2
            // takes ~2,000 cycles and a wave has one in flight; the LDS-direct load is what brings the
            // line into L2 in time (measured: 0.49 ms without, 0.22 ms with every plane load hitting)
            // -- and by the second pass from LDS.
            const __amdgpu_buffer_rsrc_t r_c = buf_rsrc(cblk, 0xFFFFFFFFu);
            const int pf_f = min(lane / 6, nf - 1), pf_q = lane - (lane / 6) * 6;
            const uint32_t pf_off = (uint32_t)pf_f * (uint32_t)(cfs * 4) + 16u * (uint32_t)pf_q;
            auto prefetch = [&](int p) {
                if (lane < NF * 6)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        r_c, (__attribute__((address_space(3))) void *)(s_c + p * (NF * 96)), 16,
                        (int)(pf_off + (uint32_t)p * 96u), 0, 0, 0);
            };
            prefetch(0);
            prefetch(1);

            // ---- first pass: candidates 0..63, lane = candidate, planes of the block in SGPRs ------
            {
                const bool act = lane < ncand;
                const int cc = act ? lane : 0;
                const int ci = (int)(((uint32_t)cc * inv_ncj) >> 16), cj = cc - ci * ncj;
                const uint32_t sr = sums[(size_t)blk * kMe16SumStride + cc];
                const uint32_t *__restrict__ rp =
                    rtab + (size_t)(i_min + STEP * ci) * wstride + (size_t)bl * kBitsWA + cj * 4;
                uint32_t acc[NF];
#pragma unroll
                for (int f = 0; f < NF; f++) acc[f] = 0;
                // one round = one row pair: the chunk of the round is in `cur`; the next round's chunk
                // and the planes two rounds ahead are requested behind the first frame (the wait that
                // frame's first use of `cur` draws covers everything issued a round ago, nothing newer)
                // Frames go in pairs: the planes of both (2 x 24 SGPRs) are requested together and
                // waited for once -- a scalar load that hits L2 still takes a few hundred cycles and a
                // wave has nothing else to do meanwhile; per pair the wait is paid once.  A pair is
                // compared if either frame is searched (the other one's counts are never used).
                auto round = [&](const uint32_t (&cur)[kBitsT], uint32_t (&nxt)[kBitsT], int p, bool has_next) {
                    auto planes = [&](uint32_t (&cs)[kBitsT], int f) {
#ifdef VCS_EXP_BITS_SAMEC   // experiment builds only (results are wrong): every plane load hits the scalar cache
                        const uint32_t *__restrict__ c = cblk + (f & 1) * kBitsT;
#else
                        const uint32_t *__restrict__ c = cblk + (size_t)min(f, nf - 1) * cfs + p * kBitsT;
#endif
#pragma unroll
                        for (int t = 0; t < kBitsT; t++) cs[t] = c[t];
                    };
#pragma unroll
                    for (int f = 0; f < NF; f += 2) {
                        if (((skip >> f) & 3u) != 3u) {   // scalar branch
                            uint32_t ca[kBitsT], cb[kBitsT];
                            planes(ca, f);
                            if (f + 1 < NF) planes(cb, f + 1);
                            acc[f] = bits_count_lt(cur, ca, acc[f]);
                            if (f + 1 < NF) acc[f + 1] = bits_count_lt(cur, cb, acc[f + 1]);
                        }
                        if (f == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (has_next) {
                                rp += 2 * wstride;
                                bits_ld_chunk(nxt, rp);
                            }
                            if (p + 2 < 8) prefetch(p + 2);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                uint32_t ra[kBitsT], rb[kBitsT];
                bits_ld_chunk(ra, rp);
#pragma unroll 1
                for (int p = 0; p < 8; p += 2) {
                    round(ra, rb, p, true);
                    round(rb, ra, p + 1, p + 2 < 8);
                }
#pragma unroll
                for (int f = 0; f < NF; f++)
                    if (act) best[f] = ((sr + 256u * acc[f]) << 7) | (uint32_t)lane;
            }

            // ---- second pass: candidates 64.., lane = (candidate, row pair), planes from LDS ------
#ifdef VCS_EXP_BITS_NOP2   // experiment builds only (results are wrong): no second pass
            if (false) {
#else
            if (ncand > 64) {
#endif
                const int n2 = ncand - 64, ntask = 8 * n2;
                const uint32_t inv_n2 = 65536u / (uint32_t)n2 + 1u;   // t / n2 for t < 288, n2 <= 36
                if (lane < 40) {
#pragma unroll
                    for (int f = 0; f < (NF + 1) / 2; f++) s_part[f * 40 + lane] = 0;
                }
                // the LDS-direct loads have landed (this wave's own: its vmcnt is what orders them)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                wave_lds_fence();
                for (int j = 0; j * 64 < ntask; j++) {
                    const int t = lane + 64 * j;
                    const bool act = t < ntask;
                    const int tc = act ? t : 0;
                    const int p = (int)(((uint32_t)tc * inv_n2) >> 16), ck = tc - p * n2;
                    const int cand = 64 + ck;
                    const int ci = (int)(((uint32_t)cand * inv_ncj) >> 16), cj = cand - ci * ncj;
                    uint32_t r[kBitsT];
                    bits_ld_chunk(r, rtab + (size_t)(i_min + STEP * ci + 2 * p) * wstride + (size_t)bl * kBitsWA + cj * 4);
                    const uint4 *cq = (const uint4 *)(s_c + p * (NF * 96));
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        if (!((skip >> f) & 1u)) {   // scalar branch
                            uint32_t cs[kBitsT];
#pragma unroll
                            for (int q = 0; q < 6; q++) {
                                const uint4 v = cq[f * 6 + q];
                                cs[4 * q] = v.x; cs[4 * q + 1] = v.y; cs[4 * q + 2] = v.z; cs[4 * q + 3] = v.w;
                            }
                            const uint32_t cnt = bits_count_lt(r, cs, 0u);
                            if (act) atomicAdd(&s_part[(f >> 1) * 40 + ck], cnt << (16 * (f & 1)));
                        }
                    }
                }
                wave_lds_fence();
                if (lane < n2) {
                    const uint32_t sr2 = sums[(size_t)blk * kMe16SumStride + 64 + lane];
#pragma unroll
                    for (int f = 0; f < NF; f++)
                        best[f] = min(best[f], ((sr2 + 256u * ((s_part[(f >> 1) * 40 + lane] >> (16 * (f & 1))) & 0xFFFFu)) << 7) |
                                               (uint32_t)(64 + lane));
                }
                wave_lds_fence();   // the next unit would rewrite s_part (one unit per wave today)
            }
        }

        // ---- arg-min per frame (motion.py:149: strict <, raster order = lowest index on ties) -----
        // the winning key of frame f goes to lane f, which decodes and stores its frame's result
        uint32_t my_key = 0xFFFFFFFFu;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            if (!((skip >> f) & 1u)) {   // scalar
                const uint32_t key = wave_min_u32(best[f]);
                if (lane == f) my_key = key;
            }
        }
        if (lane < nf) {
            const size_t o = (size_t)(f0 + lane) * g.N + blk;
            const uint2 bi = binfo[o];
            uint32_t mvw = 0u;
            unsigned long long cw = bi.x;              // static block: [0, 0], cost = the difference
            if (!((skip >> lane) & 1u)) {
                int bx = 0, byy = 0;
                cw = 9999999999ull;                    // motion.py:120: no candidate
                if (my_key != 0xFFFFFFFFu) {
                    const int cb = (int)(my_key & 127u);
                    const int cbi = (int)(((uint32_t)cb * inv_ncj) >> 16);
                    byy = i_min + cbi * STEP;
                    bx = j_min + (cb - cbi * ncj) * STEP;
                    cw = (my_key >> 7) - bi.y;
                }
                mvw = (uint32_t)(uint16_t)(bx - x) | ((uint32_t)(uint16_t)(byy - y) << 16);  // [dx, dy]
            }
            *(uint32_t *)(mv + 2 * o) = mvw;
            if (cost_out) cost_out[o] = cw;
        }
    }
    // (one workgroup in 64 reports: 8,040 workgroups adding to the same two words serialise in L2 --
    // ~12 ns per atomic -- and a wave keeps its slot until its atomics are done: with every workgroup
    // reporting, the instrumented launch could not end before 8,040 x 25 ns = 0.2 ms)
    if (clk_out && threadIdx.x == 0 && (blockIdx.x & 63u) == 0) {
        atomicAdd(clk_out + 12, __builtin_amdgcn_s_memtime() - clk0);
        atomicAdd(clk_out + 13, __builtin_amdgcn_s_memrealtime() - rt0);
    }
}

// ---- host side ----------------------------------------------------------------------------
#ifndef VCS_BITS_NF
#define VCS_BITS_NF 8
#endif
constexpr int kBitsNF = VCS_BITS_NF;   // frames per wave the search kernel is unrolled for

// Bands of block columns: the chunk table of one band stays below kBitsBandBytes -- 1 GiB, so 1080p
// (124 MB) and 4K (497 MB) are one band; measured at 4K, four 124 MB bands are slower than the one
// table (2.86 vs 2.75 ms per 29 frames: the column-major unit order already keeps a chunk's reuse
// inside one XCD's L2, the bands only add launches).  The machinery is for frames whose table would
// not fit a sensible allocation, and for tests (VCS_BITS_BAND_KB).
constexpr size_t kBitsBandBytes = (size_t)1 << 30;
inline size_t me_bits_band_bytes() {   // VCS_BITS_BAND_KB: tests force many narrow bands on small frames
    static const size_t v = [] {
        const char *e = getenv("VCS_BITS_BAND_KB");
        const long long kb = e ? atoll(e) : 0;
        return kb > 0 ? (size_t)kb << 10 : kBitsBandBytes;
    }();
    return v;
}
inline int me_bits_bands(const Geom &g) {
    const size_t per_col = (size_t)g.H * kBitsWA * 4;
    const int max_cols = (int)(me_bits_band_bytes() / per_col) > 0 ? (int)(me_bits_band_bytes() / per_col) : 1;
    return (g.nbx + max_cols - 1) / max_cols;
}
inline int me_bits_band_cols(const Geom &g) { const int nb = me_bits_bands(g); return (g.nbx + nb - 1) / nb; }
inline size_t me_bits_table_words(const Geom &g) { return (size_t)g.H * me_bits_band_cols(g) * kBitsWA; }
// the chunk table + (one band only) the row-pair sums behind it
inline size_t me_bits_rtab_bytes(const Geom &g) {
    return me_bits_table_words(g) * 4 + (size_t)g.H * g.nbx * kBitsM * 4 + 256;
}
inline size_t me_bits_cpl_bytes(const Geom &g, int n) { return (size_t)n * g.N * kBitsBlk * 4; }
inline size_t me_bits_binfo_bytes(const Geom &g, int n) {   // {static-test sum, sum(c)} per block + the static counts per workgroup
    return (size_t)n * g.N * sizeof(uint2) + (size_t)n * (8 * (((g.N + kBitsCurBlocks - 1) / kBitsCurBlocks + 7) / 8)) * sizeof(uint32_t);
}

// VCS_ME16_BITS: 0 = off (k_me16g / k_me16r / k_me16), 1 = default (batches of >= kBitsMinFrames
// frames), 2 = every launch (tests run small and ragged frames through it).
inline int me_bits_mode() {
    static const int v = [] {
        const char *e = getenv("VCS_ME16_BITS");
        const int m = e ? atoi(e) : 1;
        return m >= 0 && m <= 2 ? m : 1;
    }();
    return v;
}
// Measured on MI355X (whole step incl. k_bits_cur and the tables), bit-plane search vs k_me16g:
// 1080p (tools/gpu_bits_sizes.sh), n = 4: 0.212 vs 0.196 ms; n = 8: 0.290 vs 0.344; n = 29: 0.67 vs 0.87.
// 4K (tools/gpu_4k_ab.sh; one 497 MB table, the search over the whole batch, the tail in cache-sized
// sub-batches), n = 6: 0.92 vs 0.87 ms; n = 8: 1.13 vs 0.96; n = 12: 1.38 vs 1.47; n = 15: 1.59 vs
// 1.82; n = 29: 2.75 vs 3.55.  (The table in four bands of 60 block columns, each rebuilt per launch:
// n = 29 2.86 ms -- more launches, nothing gained from the smaller table; with the SEARCH in sub-batches
// of 8 frames, each re-reading the table: 3.8 ms.)
constexpr int kBitsMinFrames = 6, kBitsMinFramesBig = 10;
inline bool me_bits_worthwhile(const Geom &g, int n) {
    const int m = me_bits_mode();
    if (m == 0 || !me_fast_supported(g)) return false;
    if ((long long)g.N * n * kBitsBlk > 0x7FFFFFFFll * 4) return false;
    const int need = (g.N > 16384 ? kBitsMinFramesBig : kBitsMinFrames) + 3 * (me_bits_bands(g) - 1);
    return m == 2 || n >= need;
}
inline int me_bits_groups(int n) { return (n + kBitsNF - 1) / kBitsNF; }

inline void launch_bits_ref(const Geom &g, hipStream_t stream, const uint8_t *ref, size_t ref_bytes, uint32_t *rtab,
                            int a0, int na, uint32_t *r2 = nullptr) {
    const uint32_t total = (uint32_t)g.H * (uint32_t)na * kBitsM;
    hipLaunchKernelGGL(k_bits_ref, dim3((total + 255) / 256), dim3(256), 0, stream, ref, (uint32_t)ref_bytes, g, rtab,
                       a0, na, r2);
}
inline int me_bits_cur_wgs(const Geom &g) {   // workgroups of k_bits_cur per frame (a multiple of 8)
    return 8 * (((g.N + kBitsCurBlocks - 1) / kBitsCurBlocks + 7) / 8);
}
inline void launch_bits_cur(const Geom &g, hipStream_t stream, const uint8_t *curs, size_t stride, const uint8_t *ref,
                            int n, uint32_t *cpl, uint2 *binfo, uint32_t *ns_part, const uint32_t *r2 = nullptr,
                            uint32_t *sums = nullptr, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
    hipExtLaunchKernelGGL(k_bits_cur, dim3(me_bits_cur_wgs(g) * ((n + kBitsCurFpt - 1) / kBitsCurFpt)), dim3(kBitsCurThreads), 0, stream, ev_start, ev_stop, 0, curs,
                          stride, ref, g, n, cpl, binfo, ns_part, r2, sums);
}
inline void launch_bits_search(const Geom &g, hipStream_t stream, const uint32_t *rtab, const uint32_t *cpl,
                               const uint2 *binfo, const uint32_t *sums, int n, int16_t *mv, const uint32_t *ns_part,
                               uint32_t *ns, unsigned long long *cost, unsigned long long *clk, int a0, int na,
                               const uint8_t *pf_base = nullptr, size_t pf_bytes = 0, hipEvent_t ev_start = nullptr,
                               hipEvent_t ev_stop = nullptr) {
    const int ngroups = me_bits_groups(n);
    const int chunk = (((na + 7) / 8) * g.nby * ngroups + kBitsWpw - 1) / kBitsWpw * kBitsWpw;   // units per XCD (k_me16b)
    const int nsearch = 8 * (chunk / kBitsWpw);
    // prefetch workgroups: octets interleaved 1 : ratio with the last search octets (VCS_BITS_PF_WGS, VCS_BITS_PF_RATIO)
    static const int npf_env = [] { const char *e = getenv("VCS_BITS_PF_WGS"); return e ? atoi(e) : 0; }();
    static const int ratio_env = [] { const char *e = getenv("VCS_BITS_PF_RATIO"); return e ? atoi(e) : 0; }();
    // measured (tools/gpu_pf_sweep.sh): lines prefetched more than ~30 us before the tail starts are
    // gone again -- 256 workgroups, 1 : 1 behind the last search octets: tail 0.295 ms; the last 19 %
    // of the launch (1 : 3, 512): 0.32; the last 60 %: 0.36 (= no prefetch).  Count, 1 : 1 (step, no
    // events, 3 runs each): 256 / 384 / 512 / 768 / 1024 = 0.6433 / 0.6383 / 0.6439 / 0.6514 / 0.6749 ms.
    const int ratio = ratio_env > 0 ? ratio_env : 1;
    int pf8 = pf_base ? (npf_env > 0 ? npf_env : 384 * 4 / kBitsWpw) / 8 : 0;   // 1,536 prefetch waves
    if (pf8 * ratio > nsearch / 8) pf8 = nsearch / 8 / ratio;
    if (pf8 == 0) pf_base = nullptr;
    hipExtLaunchKernelGGL((k_me16b<kBitsNF>), dim3(nsearch + 8 * pf8), dim3(64 * kBitsWpw), 0, stream, ev_start, ev_stop, 0,
                          rtab, cpl, binfo, sums, g, n, ngroups, mv, cost, clk, pf_base,
                          (uint32_t)(pf_base ? pf_bytes >> 4 : 0), nsearch, pf8, ratio, ns_part, me_bits_cur_wgs(g), ns,
                          a0, na, magic_u32((uint32_t)(g.nby * ngroups)), magic_u32((uint32_t)ngroups));
}

}  // namespace vcs
