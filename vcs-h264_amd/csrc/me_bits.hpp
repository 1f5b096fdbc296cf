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
    uint32_t off = w * (uint32_t)g.pitch + 3u * (uint32_t)j;
    asm volatile("" : "+v"(off));   // (one register + immediate offsets; seen through, hipcc forms 24 addresses)
    const uint32_t off2 = off + (uint32_t)g.pitch;
    // One third of the chunk at a time (8 dwords in, 8 plane dwords out): at most 32 registers, so that a
    // wave of this kernel fits into what six k_me16b waves leave of a SIMD's register file -- the next GOP's
    // table is cut beside the search kernel (vcs_ref_next_ready), not in front of the next GOP.
    uint4 *dst = (uint4 *)(rtab + (size_t)wa * kBitsWA + m * 4);
    uint32_t rs = 0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {   // unaligned dwords (tools/buf_test.hip); string = row w's 12 dwords, then row w + 1's
            const int t = 8 * d + i;
            x[i] = t >= 12 ? buf_ld32(r_ref, off2 + 4u * (t - 12), 0) : buf_ld32(r_ref, off + 4u * t, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) rs = __builtin_amdgcn_sad_u8(x[i], 0u, rs);
        asm volatile("" : "+v"(rs));   // (summed here, not at the end with all 24 dwords kept for it)
        bits_transpose8(x);
#ifdef VCS_EXP_REF_NT   // experiment build (same results): non-temporal table stores
        { u32x4_t v0 = {x[0], x[1], x[2], x[3]}, v1 = {x[4], x[5], x[6], x[7]};
          __builtin_nontemporal_store(v0, (u32x4_t *)&dst[(2 * d) * kBitsM]);
          __builtin_nontemporal_store(v1, (u32x4_t *)&dst[(2 * d + 1) * kBitsM]); }
#else
        dst[(2 * d) * kBitsM] = make_uint4(x[0], x[1], x[2], x[3]);
        dst[(2 * d + 1) * kBitsM] = make_uint4(x[4], x[5], x[6], x[7]);
#endif
        __builtin_amdgcn_sched_barrier(0);   // (the next third's loads stay behind this third's stores)
    }
    if (r2) r2[idx] = rs;
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
// (0.022 of its 0.064 ms at 29 x 1080p, measured with a build that skipped them)
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
__device__ __forceinline__ void bits_ld_chunk(uint32_t (&r)[kBitsT], const uint32_t *__restrict__ rp) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const uint4 v = *(const uint4 *)(rp + q * (kBitsM * 4));
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
}
// the same through a buffer resource: 32-bit lane offset (bytes), no 64-bit address arithmetic per load
__device__ __forceinline__ void bits_ld_chunk_buf(uint32_t (&r)[kBitsT], __amdgpu_buffer_rsrc_t rs, uint32_t off) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
        const uint4 v = buf_ld128(rs, off + (uint32_t)q * (kBitsM * 16), 0);
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
}

// ---- the planes of one (frame, row pair) in 24 SGPRs, requested and collected BY HAND -----------
// hipcc waits for every outstanding scalar load before the first use of any of them (scalar loads
// return out of order, lgkmcnt is only exact at 0), so loads it knows about cannot be software-
// pipelined.  These two statements hide the load from it: splanes_issue requests the 96 bytes
// (s_load_dwordx16 + x8) and returns at once -- the registers are NOT valid yet --, splanes_wait is
// the s_waitcnt and, through its "+s" operands, the only path from the request to a use.  Between
// the two nothing may touch the registers (no copy, no spill): tools/check_smem_pipeline.py reads
// the compiler's assembly and fails the build if anything does.
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p) {   // a wave-uniform pointer the compiler may hold for divergent: into SGPRs
    const unsigned long long v = (unsigned long long)p;
    return (T *)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
struct SPlanes { u32x16_t lo; u32x8_t hi; };
__device__ __forceinline__ void splanes_issue(SPlanes &x, const uint32_t *base, uint32_t off) {   // off: bytes
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %4"
                 : "=&s"(x.lo), "=&s"(x.hi) : "s"(base), "s"(off), "s"(off + 64u));
}
__device__ __forceinline__ void splanes_wait(SPlanes &x) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x.lo), "+s"(x.hi));
}
// #{r < c} over one chunk with the planes in SGPRs
__device__ __forceinline__ uint32_t bits_count_lt_s(const uint32_t (&r)[kBitsT], const SPlanes &c, uint32_t n) {
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t b = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int t = 8 * d + k;
            b = bits_lt_step(r[t], t < 16 ? c.lo[t] : c.hi[t - 16], b);
        }
        n = (uint32_t)__builtin_popcount(b) + n;
        asm volatile("" : "+v"(n));
    }
    return n;
}

// frames [f0, f0 + nf) of group `grp` when n frames are dealt to `ngroups` waves as evenly as possible
__device__ __forceinline__ void bits_group(int n, int ngroups, uint32_t ngroups_magic, int grp, int &f0, int &nf) {
    const int q = (int)div_magic((uint32_t)n, ngroups_magic), r = n - q * ngroups;
    f0 = grp * q + min(grp, r);
    nf = q + (grp < r ? 1 : 0);
}

// LDS of one wave: the planes of its NF current blocks, [row pair][frame][96 bytes] (filled by
// LDS-direct loads, one per row pair), and the second pass's counts, [frame pair][40] (two 16-bit
// counts per word: a count is at most 768) -- 6,784 bytes at NF = 8: six workgroups per CU.  The
// arg-min reuses the planes' area ([frame][64 lanes] keys, NF * 256 bytes).
template <int NF>
struct BitsLds {
    static constexpr int kPlanes = 8 * NF * 96, kPart = ((NF + 1) / 2) * 40 * 4, kWave = kPlanes + kPart;
};

#ifndef VCS_BITS_WPW
#define VCS_BITS_WPW 4
#endif
constexpr int kBitsWpw = VCS_BITS_WPW;   // waves (units) per k_me16b workgroup: 4 = the frame groups of one block position when ngroups = 4
                                         // (2 / 4 / 8: search 0.243 / 0.243 / 0.269 ms -- two positions per workgroup live too long)

// The kernel's arguments travel as ONE struct, and the kernel reads them where it needs them through
// the kernarg pointer made opaque (bits_args_again): after the first pass, whose 48 plane registers
// leave the compiler ~50 scalar registers, everything is loaded and derived AGAIN instead of being
// kept -- kept, hipcc spilled ~50 scalars to vector lanes around the loops (v_writelane / v_readlane
// are vector instructions, in a kernel bound by vector issue; scalar loads and scalar arithmetic are
// not).  Round 3's build: 156 lane moves per wave.
#define BITS_ARGS_FIELDS(P32, PPTR)                                                                                   \
    PPTR(rtab, const uint32_t *) PPTR(cpl, const uint32_t *) PPTR(binfo, const uint2 *) PPTR(sums, const uint32_t *)     \
    PPTR(mv, int16_t *) PPTR(cost_out, unsigned long long *) PPTR(clk_out, unsigned long long *)                        \
    PPTR(pf_base, const uint8_t *) PPTR(ns_part, const uint32_t *) PPTR(n_static, uint32_t *)                           \
    P32(H, int) P32(W, int) P32(N, int) P32(nbx, int) P32(nby, int) P32(thresh, uint32_t) P32(never_static, int)        \
    P32(n_frames, int) P32(ngroups, int) P32(pf_total16, uint32_t) P32(n_search_wgs, int) P32(pf_octets, int)           \
    P32(pf_ratio, int) P32(ns_per_frame, int) P32(a0, int) P32(na, int) P32(per_col_magic, uint32_t)                    \
    P32(ngroups_magic, uint32_t)
struct BitsArgs {
#define BITS_DECL32(f, T) T f;
#define BITS_DECLP(f, T) T f;
    BITS_ARGS_FIELDS(BITS_DECL32, BITS_DECLP)
#undef BITS_DECL32
#undef BITS_DECLP
    // (per_col_magic, ngroups_magic: magic_u32 of nby * ngroups and of ngroups -- the unit decode is scalar, and a
    // scalar division by a runtime value is a 25-instruction vector sequence)
    uint32_t pad_[6];   // 176 bytes = 16 + 16 + 8 + 4 dwords: what bits_args_fetch loads
};
static_assert(sizeof(BitsArgs) == 176, "bits_args_fetch loads 44 dwords");
typedef const __attribute__((address_space(4))) BitsArgs *BitsArgsP;
__device__ __forceinline__ BitsArgsP bits_args() { return (BitsArgsP)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ __forceinline__ BitsArgsP bits_args_again() {   // the same pointer, but nothing loaded through the old one is reused
    BitsArgsP p = bits_args();
    asm volatile("" : "+s"(p));
    return p;
}
// All of the arguments in ONE batch of scalar loads and one wait.  Read field by field where they are used, hipcc
// places a scalar load and a wait of its own at every first use: twelve round trips to the scalar cache in a row
// at the start of every wave, before the first request for data could leave (2-3 k cycles of the ~5.7 k a wave
// spends before its first comparison).
typedef uint32_t u32x16a_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8a_t __attribute__((ext_vector_type(8)));
struct BitsWords { u32x16a_t a, b; u32x8a_t c; u32x4_t d; };
template <int OFF>
__device__ __forceinline__ uint32_t bits_word(const BitsWords &w) {
    static_assert(OFF % 4 == 0 && OFF >= 0 && OFF < 176, "inside the 44 dwords");
    if constexpr (OFF < 64) return w.a[OFF / 4];
    else if constexpr (OFF < 128) return w.b[(OFF - 64) / 4];
    else if constexpr (OFF < 160) return w.c[(OFF - 128) / 4];
    else return w.d[(OFF - 160) / 4];
}
__device__ __forceinline__ BitsArgs bits_args_fetch(BitsArgsP p) {
    BitsWords w;
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx8 %2, %4, 0x80\n\t"
                 "s_load_dwordx4 %3, %4, 0xa0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(w.a), "=&s"(w.b), "=&s"(w.c), "=&s"(w.d) : "s"(p));
    BitsArgs a;
#define BITS_GET32(f, T) a.f = (T)bits_word<offsetof(BitsArgs, f)>(w);
#define BITS_GETP(f, T)                                                                                    \
    a.f = (T)(((unsigned long long)bits_word<offsetof(BitsArgs, f) + 4>(w) << 32) | bits_word<offsetof(BitsArgs, f)>(w));
    BITS_ARGS_FIELDS(BITS_GET32, BITS_GETP)
#undef BITS_GET32
#undef BITS_GETP
    return a;
}
// 65536 / n + 1 (x / n for x < 512, n <= 36, as a multiply and a shift): looked up, not divided
__device__ __constant__ const uint32_t kBitsInv[40] = {
    0, 65537, 32769, 21846, 16385, 13108, 10923, 9363, 8193, 7282, 6554, 5958, 5462, 5042, 4682, 4370, 4097, 3856, 3641, 3450,
    3277, 3121, 2979, 2850, 2731, 2622, 2521, 2428, 2341, 2260, 2185, 2115, 2049, 1986, 1928, 1873, 1821, 1772, 1725, 1681};

// what a wave needs to know about its unit (block position, frame group): all scalar
struct BitsUnit {
    int bl, blk, f0, nf, x, y, i_min, j_min, nci, ncj, ncand;
    uint32_t inv_ncj;
};
__device__ __forceinline__ bool bits_unit(int ul, const BitsArgs &A, int xcd, BitsUnit &u) {
    constexpr int BS = 16, WIN = 32, STEP = 5;
    const int ngroups = A.ngroups, na = A.na;
    const uint32_t ngroups_magic = A.ngroups_magic;
    const int per_col = A.nby * ngroups;
    const int chunk = (((na + 7) / 8) * per_col + kBitsWpw - 1) / kBitsWpw * kBitsWpw;   // units per XCD (the XCDs with one column less idle at the end)
    const int lc = (int)div_magic((uint32_t)ul, A.per_col_magic);
    u.bl = lc * 8 + xcd;   // column inside the band
    if (!(ul < chunk && u.bl < na)) return false;
    const int rem = ul - lc * per_col;
    const int byi = (int)div_magic((uint32_t)rem, ngroups_magic), bxi = A.a0 + u.bl;
    // the groups of a position differ in size (29 frames: 8, 7, 7, 7): rotated from position to
    // position, so that the same wave slot (SIMD) does not get the large one every time (-0.4 % of
    // the step in 3 of 3 A/B pairs)
    const int gr = rem - byi * ngroups + byi;
    const int grp = gr - ngroups * (int)div_magic((uint32_t)gr, ngroups_magic);
    u.blk = byi * A.nbx + bxi;
    bits_group(A.n_frames, ngroups, ngroups_magic, grp, u.f0, u.nf);
    u.x = bxi * BS; u.y = byi * BS;
    u.i_min = max(u.y - WIN, 0);
    u.j_min = max(u.x - WIN, 0);
    const int i_max = min(u.y + WIN, A.H), j_max = min(u.x + WIN, A.W);
    const int ri = i_max - BS - u.i_min, rj = j_max - BS - u.j_min;
    u.nci = ri >= 1 ? (ri - 1) / STEP + 1 : 0;
    u.ncj = rj >= 1 ? (rj - 1) / STEP + 1 : 0;
    u.ncand = u.nci * u.ncj;
    u.inv_ncj = kBitsInv[u.ncj];   // cand / ncj for cand < 128, ncj <= 10
    return true;
}

// The second pass's task of a lane in wave pass j (candidates 64.., lane = (candidate, row pair); the
// half-filled last pass gives the two halves of the wave the same tasks for different frames).
struct BitsTask { uint32_t p, ck, off; int half; bool act; };
__device__ __forceinline__ BitsTask bits_task(int j, int lane, int n2, uint32_t inv_n2, const BitsUnit &u, uint32_t wstride4) {
    constexpr int STEP = 5;
    const int ntask = 8 * n2;
    const bool split = ntask - 64 * j <= 32;
    BitsTask k;
    k.half = split ? lane >> 5 : 0;
    const int t = 64 * j + (split ? lane & 31 : lane);
    k.act = t < ntask;
    const uint32_t tc = k.act ? (uint32_t)t : 0u;
    // (24-bit multiplies: a full v_mul_lo_u32 issues at a quarter of the rate)
    k.p = __umul24(tc, inv_n2) >> 16;
    k.ck = tc - __umul24(k.p, (uint32_t)n2);
    const uint32_t cand = 64u + k.ck;
    const uint32_t ci = __umul24(cand, u.inv_ncj) >> 16, cj = cand - __umul24(ci, (uint32_t)u.ncj);
    k.off = __umul24((uint32_t)u.i_min + STEP * ci + 2u * k.p, wstride4) + ((uint32_t)u.bl * (kBitsWA * 4) + cj * 16u);
    return k;
}

#ifdef VCS_EXP_PHASES   // experiment builds only (tools/gpu_phases_bits.py): where a search wave's time goes
__device__ unsigned long long g_me16b_phases[8];   // [0..4]: shader cycles in prologue / first pass / between / second pass / end; [5]: waves sampled
#define BITS_PROBE(i) do { if (probing) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] = t_; } } while (0)
#else
#define BITS_PROBE(i) do { } while (0)
#endif

template <int NF>
__global__ __launch_bounds__(64 * kBitsWpw) void k_me16b(BitsArgs args_by_value) {
    constexpr int STEP = 5;
    static_assert(NF % 4 == 0 && NF <= 8, "two plane buffers in turn; the split pass pairs frame f with f + NF / 2 in one counter word; NF * 6 lanes prefetch");
    using L = BitsLds<NF>;
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[kBitsWpw * L::kWave];
    const BitsArgs A_ = bits_args_fetch(bits_args());
    const BitsArgs *const A = &A_;
    // clk_out (timing on): shader-clock cycles and 100 MHz ticks this workgroup lived, summed over
    // the workgroups: their quotient is the clock the chip held under this load
    const bool clocked = A->clk_out != nullptr && threadIdx.x == 0 && (blockIdx.x & 63u) == 0;
    const unsigned long long clk0 = clocked ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long rt0 = clocked ? __builtin_amdgcn_s_memrealtime() : 0ull;
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
    if (A->pf_octets) {
        const int pf_ratio = A->pf_ratio;
        const int head = A->n_search_wgs / 8 - pf_ratio * A->pf_octets;
        if (oct >= head) {
            const int o2 = oct - head, grp = o2 / (pf_ratio + 1), r = o2 - grp * (pf_ratio + 1);
            is_pf = r == pf_ratio;
            oct = is_pf ? grp : head + grp * pf_ratio + r;
        }
    }
    if (is_pf) {
        const uint32_t pf_total16 = A->pf_total16;
        const uint32_t nw = (uint32_t)A->pf_octets * 8u * (uint32_t)kBitsWpw;
        const uint32_t w = ((uint32_t)oct * 8u + (blockIdx.x & 7u)) * (uint32_t)kBitsWpw + (uint32_t)wave;
        const __amdgpu_buffer_rsrc_t r_pf = buf_rsrc(A->pf_base, pf_total16 << 4);
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
    if (blockIdx.x == 0 && A->a0 == 0) {   // static blocks per frame: the sums of k_bits_cur's per-workgroup counts
        const int ns_per_frame = A->ns_per_frame;
        for (int f = wave; f < A->n_frames; f += kBitsWpw) {
            uint32_t v = 0;
            for (int i = lane; i < ns_per_frame; i += 64) v += A->ns_part[(size_t)f * ns_per_frame + i];
            v = wave_sum_u32(v);
            if (lane == 0) A->n_static[f] = v;
        }
    }
    const int xcd = blockIdx.x & 7;
    int ul = oct * kBitsWpw + wave;
    BitsUnit u;
#ifdef VCS_EXP_PHASES
    const bool probing = ((blockIdx.x >> 3) & 15) == 3;   // one workgroup octet in 16
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
#endif
    BITS_PROBE(0);
    if (bits_unit(ul, A_, xcd, u)) {
        // A wave's first loads -- the static-test records, the first chunk, the planes of the first two
        // row pairs -- are all requested before the first of them is waited for: one round trip at the
        // start of a wave instead of ten in a row (round 3 read the eight records by scalar loads, each
        // behind a branch; they were written by another XCD's k_bits_cur, so every one misses L2:
        // search -5.7 %).  The chunk and the planes are requested on spec: a wave whose frames all turn
        // out static has loaded 8 KB for nothing.
        // (All of these addresses are valid whatever the unit's geometry: row i_min of the table, frame
        // f0 .. f0 + nf - 1 >= f0 of the planes and the records -- a group has at least one frame.)
        // per frame: static?  Lane f reads frame f's record (read again at the end rather than kept)
        const int N = A->N;
        uint32_t sd = A->binfo[(size_t)(u.f0 + max(min(lane, u.nf - 1), 0)) * N + u.blk].x;

        const uint32_t wstride4 = (uint32_t)A->na * (kBitsWA * 4);   // bytes per window row of rtab (a band is at most 1 GiB)
        const __amdgpu_buffer_rsrc_t r_t = buf_rsrc(A->rtab, 0xFFFFFFFFu);
        const uint32_t *__restrict__ cblk = uniform_ptr(A->cpl + ((size_t)u.f0 * N + u.blk) * kBitsBlk);
        const uint32_t cfs4 = (uint32_t)N * (kBitsBlk * 4);   // bytes per frame of cpl (a group spans < 4 GiB: me_bits_worthwhile)
        // The planes of the wave's NF current blocks arrive by LDS-direct loads, one instruction
        // per row pair (lane = (frame, 16-byte piece)), two rounds ahead of their use.  They are
        // read twice: by the first pass through s_load from L2 -- a scalar load that misses L2
        // takes ~2,000 cycles and a wave has one in flight; the LDS-direct load is what brings the
        // line into L2 in time (measured: 0.49 ms without, 0.22 ms with every plane load hitting)
        // -- and by the second pass from LDS.
        const __amdgpu_buffer_rsrc_t r_c = buf_rsrc(cblk, 0xFFFFFFFFu);
        const uint32_t pf_fr = (uint32_t)max(min(lane / 6, u.nf - 1), 0);
        const uint32_t pf_off = pf_fr * cfs4 + 16u * (uint32_t)(lane - (lane / 6) * 6);
        auto prefetch = [&](int p) {
            if (lane < NF * 6)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    r_c, (__attribute__((address_space(3))) void *)(s_c + p * (NF * 96)), 16,
                    (int)(pf_off + (uint32_t)p * 96u), 0, 0, 0);
        };
        // first pass: candidates 0..63, lane = candidate; a lane without a candidate compares candidate 0's chunk
        // (24-bit multiplies: a full v_mul_lo_u32 issues at a quarter of the rate)
        const uint32_t cc = lane < u.ncand ? (uint32_t)lane : 0u;
        const uint32_t ci = __umul24(cc, u.inv_ncj) >> 16, cj = cc - __umul24(ci, (uint32_t)u.ncj);
        uint32_t roff = __umul24((uint32_t)u.i_min + STEP * ci, wstride4) + ((uint32_t)u.bl * (kBitsWA * 4) + cj * 16u);
        uint32_t ra[kBitsT], rb[kBitsT];
        SPlanes pa, pb;
        prefetch(0);
        bits_ld_chunk_buf(ra, r_t, roff);
        prefetch(1);
        const uint32_t sr = A->sums[(size_t)u.blk * kMe16SumStride + cc];
        splanes_issue(pa, cblk, 0u);
        asm volatile("" : "+v"(sd));   // the record's first use (and the wait it draws) stays behind the requests
        const bool st = lane >= u.nf || (!A->never_static && sd <= A->thresh);
        const uint32_t skip = (uint32_t)__ballot(st) & ((1u << NF) - 1u);   // bit f: nothing to search for frame f0 + f

        uint32_t best[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) best[f] = 0xFFFFFFFFu;

        const bool searching = u.ncand > 0 && skip != (1u << NF) - 1u;
        // The planes of (frame 0, row pair 0) are collected here, beside the records' round trip: from here
        // on the compiler may move them (it did, into other registers at the loop's head, while they were
        // still in flight: tools/check_smem_pipeline.py).
        splanes_wait(pa);
        if (!searching)   // nothing may be in flight when the wave ends (the LDS-direct loads write LDS)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BITS_PROBE(1);
        if (searching) {
            // ---- first pass: planes of the block in SGPRs ----------------------------------------
            uint32_t acc[NF];
#pragma unroll
            for (int f = 0; f < NF; f++) acc[f] = 0;
            // One round = one row pair: the chunk of the round is in `cur`; the next round's chunk
            // and the LDS-direct load two rounds ahead are requested behind the first frame.
            // Frames go ONE AT A TIME through two plane buffers: while frame f is compared out of
            // one, frame f + 1 (the next round's frame 0 behind the last) is on its way into the
            // other -- a scalar load that hits L2 still takes a few hundred cycles, which the 27
            // vector instructions of a frame (x the other waves of the SIMD) cover.  (Round 3
            // waited once per PAIR of frames with nothing in flight, and compared a group's
            // missing eighth frame because the pair was one block of code: 29 frames = 8, 7, 7, 7
            // cost 32 frame comparisons per candidate, now 29.)
            // (what the loop needs is derived inside it from three scalars made opaque per round --
            // otherwise hipcc hoists the 8 frame offsets and the 8 skip tests, as 64-bit lane masks,
            // out of the loop and spills them around the 48 plane registers)
            const int last = u.nf - 1;
            auto round = [&](const uint32_t (&cur)[kBitsT], uint32_t (&nxt)[kBitsT], int p, bool has_next) {
                uint32_t sk = skip, fs = cfs4, lf = (uint32_t)last;
                asm volatile("" : "+s"(sk), "+s"(fs), "+s"(lf));
                const uint32_t po = (uint32_t)p * 96u;
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    SPlanes &mine = (f & 1) ? pb : pa;
                    SPlanes &other = (f & 1) ? pa : pb;
                    splanes_wait(mine);
                    // (behind the last round the request repeats row pair 7 of frame 0: harmless, and no
                    // branch around a statement whose registers are in flight)
                    splanes_issue(other, cblk, f + 1 < NF ? min((uint32_t)(f + 1), lf) * fs + po : (uint32_t)min(p + 1, 7) * 96u);
                    if (!((sk >> f) & 1u)) acc[f] = bits_count_lt_s(cur, mine, acc[f]);   // scalar branch
                    if (f == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (has_next) {
                            roff += 2u * wstride4;
                            bits_ld_chunk_buf(nxt, r_t, roff);
                        } else if (u.ncand > 64) {   // the second pass's first chunk, into the buffer the last round leaves free
                            const int n2 = u.ncand - 64;
                            bits_ld_chunk_buf(nxt, r_t, bits_task(0, lane, n2, kBitsInv[n2], u, wstride4).off);
                        }
                        if (p + 2 < 8) prefetch(p + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
#pragma unroll 1
            for (int p = 0; p < 8; p += 2) {
                round(ra, rb, p, true);
                round(rb, ra, p + 1, p + 2 < 8);
            }
            splanes_wait(pa);   // the request behind the last round
            // key = (cost << 7) | candidate, cost = sum(r) - sum(c) + 256 * #{r < c} without the -sum(c),
            // which is the same for all candidates.  A lane without a candidate has compared candidate
            // 0's chunk: its key is candidate 0's cost with a higher index, which never wins.
            const uint32_t base = (sr << 7) | (uint32_t)lane;
#pragma unroll
            for (int f = 0; f < NF; f++) best[f] = base + (acc[f] << 15);
        }
        BITS_PROBE(2);
        // everything is loaded and derived again from here on (see BitsArgs)
        asm volatile("" : "+s"(ul));
        const BitsArgs B_ = bits_args_fetch(bits_args_again());
        const BitsArgs *const B = &B_;
        bits_unit(ul, B_, xcd, u);
        if (searching && u.ncand > 64) {
            // ---- second pass: candidates 64.., lane = (candidate, row pair), planes from LDS ------
            // n2 candidates x 8 row pairs = up to 288 tasks = 4.5 wave passes.  The half-filled last
            // pass splits the FRAMES between the wave's halves (lanes 32.. repeat the tasks of lanes
            // 0..31 for frames NF/2..): 4.5 passes cost 4.5, not 5.
            const int n2 = u.ncand - 64, ntask = 8 * n2;
            const uint32_t inv_n2 = kBitsInv[n2];   // t / n2 for t < 288, n2 <= 36
            const uint32_t wstride4b = (uint32_t)B->na * (kBitsWA * 4);
            const __amdgpu_buffer_rsrc_t r_t2 = buf_rsrc(B->rtab, 0xFFFFFFFFu);
            if (lane < 40) {
#pragma unroll
                for (int f = 0; f < (NF + 1) / 2; f++) s_part[f * 40 + lane] = 0;
            }
            // the LDS-direct loads have landed (this wave's own: its vmcnt is what orders them)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            BITS_PROBE(3);
            // (Requesting EVERY pass's chunk a pass ahead -- the first pass's two chunk buffers in turn -- needs
            // 93 registers with the 24 of the planes: five waves per SIMD instead of six, or scratch traffic
            // in the frame loop when capped at 80.  Only pass 0's is ahead.)
            for (int j = 0; j * 64 < ntask; j++) {
                const bool split = ntask - 64 * j <= 32;           // the frames go to the two halves of the wave
                const BitsTask tk = bits_task(j, lane, n2, inv_n2, u, wstride4b);
                const int half = tk.half;
                const uint32_t p = tk.p, ck = tk.ck;
                const bool act = tk.act;
                if (j > 0) bits_ld_chunk_buf(ra, r_t2, tk.off);   // (pass 0's chunk was requested in the first pass's last round)
                // frame of slot f: f (whole pass) or f + half * NF / 2, slots NF / 2.. unused (split pass)
                const uint4 *cq = (const uint4 *)(s_c + p * (NF * 96) + half * (NF / 2 * 96));
                // (a lane without a task adds to slot 39, which no candidate owns: no branch around the add; a
                // lane whose own frame is static adds to that frame's count, which nobody uses)
                uint32_t *part = s_part + half * (NF / 4 * 40) + (act ? ck : 39u);
                const uint32_t sk = split ? ((skip & (skip >> (NF / 2))) | (0xFFu << (NF / 2))) : skip;   // scalar: slot has no frame to do
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    if (!((sk >> f) & 1u)) {   // scalar branch
                        uint32_t cs[kBitsT];
#pragma unroll
                        for (int q = 0; q < 6; q++) {
                            const uint4 v = cq[f * 6 + q];
                            cs[4 * q] = v.x; cs[4 * q + 1] = v.y; cs[4 * q + 2] = v.z; cs[4 * q + 3] = v.w;
                        }
                        const uint32_t cnt = bits_count_lt(ra, cs, 0u);
                        atomicAdd(&part[(f >> 1) * 40], cnt << (16 * (f & 1)));
                    }
                }
            }
            wave_lds_fence();
            BITS_PROBE(4);
            if (lane < n2) {
                const uint32_t base2 = (B->sums[(size_t)u.blk * kMe16SumStride + 64 + lane] << 7) | (uint32_t)(64 + lane);
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    const uint32_t w = s_part[(f >> 1) * 40 + lane];
                    best[f] = min(best[f], base2 + (((f & 1) ? w >> 16 : w & 0xFFFFu) << 15));
                }
            }
        }

        // ---- arg-min per frame (motion.py:149: strict <, raster order = lowest index on ties) -----
        // Through LDS (the planes' area is free now): the 64 keys of frame f go to row f, lane
        // (f, s) takes the minimum of keys 8 s .. 8 s + 7 and three DPP steps finish it inside the
        // 8 lanes of a frame -- 7 vector instructions for all frames (a wave reduction per frame: 13 each).
        // Lane 8 f then decodes and stores frame f's result.
        uint32_t my_key = 0xFFFFFFFFu;
        if (searching) {
            uint32_t *s_red = (uint32_t *)s_c;
            wave_lds_fence();   // the second pass's reads of the planes are done
#pragma unroll
            for (int f = 0; f < NF; f++) s_red[f * 64 + lane] = best[f];
            wave_lds_fence();
            if (lane < NF * 8) {
                const uint4 a = *(const uint4 *)(s_red + lane * 8), b = *(const uint4 *)(s_red + lane * 8 + 4);
                my_key = min(min(min(a.x, a.y), min(a.z, a.w)), min(min(b.x, b.y), min(b.z, b.w)));
            }
            my_key = min(my_key, dpp_u32<0xB1>(my_key));    // quad_perm:[1,0,3,2]
            my_key = min(my_key, dpp_u32<0x4E>(my_key));    // quad_perm:[2,3,0,1]
            my_key = min(my_key, dpp_u32<0x141>(my_key));   // row_half_mirror: the other quad of the 8 lanes
        }
        const int fr = lane >> 3;
        if ((lane & 7) == 0 && fr < u.nf) {
            const size_t o = (size_t)(u.f0 + fr) * B->N + u.blk;
            const bool searched = !((skip >> fr) & 1u), found = searched && my_key != 0xFFFFFFFFu;
            uint32_t mvw = 0u;                         // static block: [0, 0]
            if (searched) {
                int bx = 0, byy = 0;                   // motion.py:120: no candidate
                if (found) {
                    const int cb = (int)(my_key & 127u);
                    const int cbi = (int)(((uint32_t)cb * u.inv_ncj) >> 16);
                    byy = u.i_min + cbi * STEP;
                    bx = u.j_min + (cb - cbi * u.ncj) * STEP;
                }
                mvw = (uint32_t)(uint16_t)(bx - u.x) | ((uint32_t)(uint16_t)(byy - u.y) << 16);  // [dx, dy]
            }
            *(uint32_t *)(B->mv + 2 * o) = mvw;
            // costs are a debugging output: only then is the block's record read again (a round trip at the end
            // of every wave otherwise)
            if (B->cost_out) {
                const uint2 bi = B->binfo[o];
                B->cost_out[o] = !searched ? (unsigned long long)bi.x      // static: the difference
                                 : found ? (unsigned long long)((my_key >> 7) - bi.y) : 9999999999ull;
            }
        }
    }
#ifdef VCS_EXP_PHASES
    if (probing && lane == 0 && ph[1] && ph[2]) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (!ph[3]) ph[3] = ph[4] = ph[2];
        atomicAdd(&g_me16b_phases[0], ph[1] - ph[0]);
        atomicAdd(&g_me16b_phases[1], ph[2] - ph[1]);
        atomicAdd(&g_me16b_phases[2], ph[3] - ph[2]);
        atomicAdd(&g_me16b_phases[3], ph[4] - ph[3]);
        atomicAdd(&g_me16b_phases[4], t_end - ph[4]);
        atomicAdd(&g_me16b_phases[5], 1ull);
    }
#endif
    // (one workgroup in 64 reports: 8,040 workgroups adding to the same two words serialise in L2 --
    // ~12 ns per atomic -- and a wave keeps its slot until its atomics are done: with every workgroup
    // reporting, the instrumented launch could not end before 8,040 x 25 ns = 0.2 ms)
    if (clocked) {
        unsigned long long *clk_out = bits_args_again()->clk_out;
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
// Round 4 (the restructured k_me16b; tools/exp sizes run, k macroblocks/ms, bit-plane vs k_me16g): 1080p n = 3: 135 vs 151;
// n = 4: 163 vs 164; n = 5: 187 vs 171; n = 6: 206 vs 177; n = 8: 235 vs 186.  4K n = 4: 166 vs 231; n = 6: 200 vs 228;
// n = 8: 226 vs 273; n = 10: 282 vs 251; n = 12: 301 vs 266.
constexpr int kBitsMinFrames = 5, kBitsMinFramesBig = 10;
inline bool me_bits_worthwhile(const Geom &g, int n) {
    const int m = me_bits_mode();
    if (m == 0 || !me_fast_supported(g)) return false;
    if ((long long)g.N * n * kBitsBlk > 0x7FFFFFFFll * 4) return false;
    // the unit decode divides by multiply-high (div_magic: exact while x * d < 2^32): unit < ceil(columns / 8) *
    // per_col is divided by per_col = nby * groups -- beyond that (1080p: ~2,000 frames per launch) the
    // shared-window kernels take the batch
    {
        const unsigned long long per_col = (unsigned long long)g.nby * (unsigned long long)((n + kBitsNF - 1) / kBitsNF);
        if ((unsigned long long)((me_bits_band_cols(g) + 7) / 8) * per_col * per_col >= (1ull << 32)) return false;
    }
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
    BitsArgs a;
    a.rtab = rtab; a.cpl = cpl; a.binfo = binfo; a.sums = sums; a.mv = mv; a.cost_out = cost; a.clk_out = clk;
    a.pf_base = pf_base; a.ns_part = ns_part; a.n_static = ns;
    a.H = g.H; a.W = g.W; a.N = g.N; a.nbx = g.nbx; a.nby = g.nby; a.thresh = g.thresh; a.never_static = g.never_static;
    a.n_frames = n; a.ngroups = ngroups; a.pf_total16 = (uint32_t)(pf_base ? pf_bytes >> 4 : 0);
    a.n_search_wgs = nsearch; a.pf_octets = pf8; a.pf_ratio = ratio; a.ns_per_frame = me_bits_cur_wgs(g);
    a.a0 = a0; a.na = na; a.per_col_magic = magic_u32((uint32_t)(g.nby * ngroups)); a.ngroups_magic = magic_u32((uint32_t)ngroups);
    hipExtLaunchKernelGGL((k_me16b<kBitsNF>), dim3(nsearch + 8 * pf8), dim3(64 * kBitsWpw), 0, stream, ev_start, ev_stop, 0, a);
}

}  // namespace vcs
