// dct_kernels.hpp -- 8x8 DCT / quantise and dequantise / IDCT on gfx950
// (DCTcompressor.py:49-93), fused with motion compensation and the residual
// (motion.py:38-69, encoder.py:58-66, decoder.py:52-69).
//
// Arithmetic contract (bit-exact with oracle/vcs_oracle.c, which is pinned to
// the reference's outputs): every 8x8 matrix product is a k-ascending chain
// of fused multiply-adds in float64 starting from +0.0; the quantiser is an
// IEEE division (np.true_divide, :71), the dequantiser an IEEE product (:86).
// This translation unit is compiled with -ffp-contract=off so that only the
// explicit fma() calls fuse.
//
// Mapping: one wavefront owns 8 horizontally adjacent tiles (64 px x 8 rows).
// Stage 1, lane = (tile t, column j): the lane holds column j of its tile for
// the three channels in registers and multiplies by the DCT matrix (compile-
// time constants, kDct8).  The 8x8 intermediate goes through LDS once to turn
// columns into rows; stage 2, lane = (tile t, row i), produces 8 horizontally
// adjacent outputs per lane, i.e. 64 contiguous bytes of a float64 plane.
#pragma once
#include "vcs_common.hpp"

namespace vcs {

constexpr int kTilePad = 72;  // doubles per tile in the LDS transpose buffer
constexpr int kOutPad = 9;    // elements per output row in the LDS store buffer (8 + 1 pad)
#ifndef VCS_FDCT_WAVES
#define VCS_FDCT_WAVES 8
#endif
// strips (waves) per k_fdct workgroup: eight adjacent strips on one CU (4 KB of every plane row in a run,
// shared pixel lines through one L1).  Round 3, 2-D grid: 4 / 6 / 8 / 10 / 12 waves = 0.288 / 0.278 / 0.275 /
// 0.279 / 0.277 ms per 29 x 1080p (round 2, 1-D grid: 1 / 2 / 4 / 8 = 0.300 / 0.286 / 0.288 / 0.282).
constexpr int kFdctWaves = VCS_FDCT_WAVES;
#ifndef VCS_IDCT_WAVES
#define VCS_IDCT_WAVES 4
#endif
constexpr int kIdctWaves = VCS_IDCT_WAVES;   // strips (waves) per k_idct workgroup: 1 / 2 / 4 = 0.311 / 0.299 / 0.298 ms (round 3, with non-temporal plane loads)

// coefficient store / load for the three coefficient modes (vcs_hip.h)
template <int CM> struct CoeffT;
template <> struct CoeffT<0> { using type = double; };
template <> struct CoeffT<1> { using type = float; };
template <> struct CoeffT<2> { using type = int16_t; };

template <int CM>
__device__ __forceinline__ typename CoeffT<CM>::type coeff_pack(double v) {
    if constexpr (CM == 0) return v;
    else if constexpr (CM == 1) return (float)v;
    else return (int16_t)(int)__builtin_rint(v);  // np.round: half to even
}

// The 8x8 DCT-II matrix of DCTcompressor.py:124-133, D[i][j] = sqrt(2/8) cos((2j+1) i pi/16)
// (row 0: 1/sqrt(8)), as float64 bit patterns: the values the reference's NumPy expression
// yields (tests/golden/dct_small.npz "dctmat"; also what libm's cos gives, oracle/vcs_oracle.c
// vcso_dct_matrix).  Not symmetric to the last bit (36 distinct magnitudes), so it is kept as
// 64 literals.  Compile-time constants on purpose: as a runtime table the 128 SGPRs it needs
// do not fit the scalar register file and hipcc spilled them through VGPR lanes -- a third of
// k_fdct's and half of k_idct's vector instructions were v_readlane / v_writelane.
#define VCS_DCT8_TABLE \
    0x1.6a09e667f3bccp-2, 0x1.6a09e667f3bccp-2, 0x1.6a09e667f3bccp-2, 0x1.6a09e667f3bccp-2, \
    0x1.6a09e667f3bccp-2, 0x1.6a09e667f3bccp-2, 0x1.6a09e667f3bccp-2, 0x1.6a09e667f3bccp-2, \
    0x1.f6297cff75cb0p-2, 0x1.a9b66290ea1a3p-2, 0x1.1c73b39ae68c9p-2, 0x1.8f8b83c69a60dp-4, \
    -0x1.8f8b83c69a608p-4, -0x1.1c73b39ae68c6p-2, -0x1.a9b66290ea1a4p-2, -0x1.f6297cff75cb0p-2, \
    0x1.d906bcf328d46p-2, 0x1.87de2a6aea964p-3, -0x1.87de2a6aea962p-3, -0x1.d906bcf328d46p-2, \
    -0x1.d906bcf328d47p-2, -0x1.87de2a6aea96dp-3, 0x1.87de2a6aea967p-3, 0x1.d906bcf328d44p-2, \
    0x1.a9b66290ea1a3p-2, -0x1.8f8b83c69a608p-4, -0x1.f6297cff75cb0p-2, -0x1.1c73b39ae68c8p-2, \
    0x1.1c73b39ae68c5p-2, 0x1.f6297cff75cb0p-2, 0x1.8f8b83c69a61dp-4, -0x1.a9b66290ea1a2p-2, \
    0x1.6a09e667f3bcdp-2, -0x1.6a09e667f3bccp-2, -0x1.6a09e667f3bcep-2, 0x1.6a09e667f3bcbp-2, \
    0x1.6a09e667f3bcep-2, -0x1.6a09e667f3bc5p-2, -0x1.6a09e667f3bc9p-2, 0x1.6a09e667f3bc4p-2, \
    0x1.1c73b39ae68c9p-2, -0x1.f6297cff75cb0p-2, 0x1.8f8b83c69a60cp-4, 0x1.a9b66290ea1a5p-2, \
    -0x1.a9b66290ea1a2p-2, -0x1.8f8b83c69a602p-4, 0x1.f6297cff75cb2p-2, -0x1.1c73b39ae68c2p-2, \
    0x1.87de2a6aea964p-3, -0x1.d906bcf328d47p-2, 0x1.d906bcf328d44p-2, -0x1.87de2a6aea965p-3, \
    -0x1.87de2a6aea971p-3, 0x1.d906bcf328d46p-2, -0x1.d906bcf328d43p-2, 0x1.87de2a6aea95fp-3, \
    0x1.8f8b83c69a60dp-4, -0x1.1c73b39ae68c8p-2, 0x1.a9b66290ea1a5p-2, -0x1.f6297cff75cb2p-2, \
    0x1.f6297cff75cb0p-2, -0x1.a9b66290ea1a1p-2, 0x1.1c73b39ae68c2p-2, -0x1.8f8b83c69a616p-4
__device__ const double kDct8[64] = {VCS_DCT8_TABLE};
static const double kDct8Host[64] = {VCS_DCT8_TABLE};

// One BGR pixel as the low 3 bytes of an unaligned dword load (the buffers carry
// slack behind the last pixel): one load per pixel instead of three.
__device__ __forceinline__ uint32_t load_px(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// the same from a wave-uniform base and an unsigned 32-bit byte offset: the base stays in
// scalar registers and the load needs no 64-bit vector address arithmetic
__device__ __forceinline__ uint32_t load_px_o(const uint8_t *__restrict__ base, uint32_t off) {
    uint32_t v;
    __builtin_memcpy(&v, base + off, 4);
    return v;
}
// a / b, correctly rounded, from y = RN(1 / b) in three fused operations
// (Markstein): q = RN(a*y); r = RN(a - q*b) (exact); result = RN(q + r*y).
// Valid when nothing over/underflows, b > 0 and b's significand is not all ones; the
// host only selects this path for such Q tables (vcs_api.hip, refresh_q).
// Bit-identical to IEEE division: 1.5e9 random / adversarial cases checked
// against a / b on the CPU for every integer Q in 1..255, 0 mismatches.  The one
// input where the three operations differ from a / b is a = -0.0 (+0.0 instead of
// -0.0), and k_fdct never divides it: its numerators are k-ascending fma chains that
// start from +0.0, and round-to-nearest sums are -0.0 only if every term is
// (tests/test_gpu_parity.py compares the planes as raw 64-bit patterns).
__device__ __forceinline__ double div_markstein(double a, double b, double y) {
    const double q = a * y;
    const double r = __builtin_fma(-q, b, a);
    return __builtin_fma(r, y, q);
}

// Predicted (motion-compensated) pixel for frame position (y, x), or 0 where
// no full motion block covers it (motion.py:45-46, :83-87).  Low 3 bytes: B G R.
// Branch-free: the coordinates are clamped into the covered area so that both
// loads are unconditional (a conditional load makes hipcc wait for it on the
// spot, which serialises the eight rows of a strip), and the result is
// selected afterwards.  Needs g.N > 0.
// pred_mv: the vector (dx | dy << 16) of the block that covers (y, x) clamped into the covered
// area; pred_px_m: the pixel it points at.  Motion blocks are multiples of 8 in every
// BASELINE configuration, so the 8 rows of a k_fdct strip column share one block: the kernel
// then fetches the vector once instead of 8 times (tail 0.335 -> 0.303 ms).
__device__ __forceinline__ uint32_t pred_mv(const int16_t *__restrict__ mv, const Geom &g, int y, int x) {
    const int yc = min(y, g.nby * g.bs - 1), xc = min(x, g.nbx * g.bs - 1);
    const int blk = (int)div_magic((uint32_t)yc, g.bs_magic) * g.nbx + (int)div_magic((uint32_t)xc, g.bs_magic);
    return load_px_o((const uint8_t *)mv, 4u * (uint32_t)blk);
}
// (the vectors the search kernels produce, and the ones the host lets through, keep the
// predicted block inside the frame: the offset below is never negative)
__device__ __forceinline__ uint32_t pred_px_m(const uint8_t *__restrict__ ref, const Geom &g, int y,
                                              int x, uint32_t m) {
    const int yc = min(y, g.nby * g.bs - 1), xc = min(x, g.nbx * g.bs - 1);
    const int dx = (int16_t)(m & 0xFFFFu), dy = (int16_t)(m >> 16);
    const uint32_t v = load_px_o(ref, (uint32_t)((yc + dy) * g.pitch + 3 * (xc + dx)));
    return (yc == y && xc == x) ? v : 0u;
}
__device__ __forceinline__ uint32_t pred_px(const uint8_t *__restrict__ ref,
                                            const int16_t *__restrict__ mv, const Geom &g,
                                            int y, int x) {
    return pred_px_m(ref, g, y, x, pred_mv(mv, g, y, x));
}


// Strip (64 px x 8 rows) of frame f that this workgroup owns.  Workgroups are
// dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2, and a
// strip's 192-byte pixel rows share cache lines with the strips left and right
// of it, so the linear id is remapped to keep neighbours on one XCD:
//   map 0: raster (linear id = f * per_frame + strip);
//   map 1: XCD x owns a contiguous run of (frame, strip) ids;
//   map 2: XCD x owns the same band of strips in every frame.
// The grid is 1-D with tail_grid() workgroups; false = nothing to do.
// Raster (map 0, the default) launches a 2-D grid -- x: the frame's strips, `waves` per workgroup,
// y: the frame -- so that the frame is blockIdx.y and the strip's row and 64-pixel group come from
// one multiply-high (the linear dispatch order is the same raster, x fastest).  Maps 1 and 2 (tests,
// experiments) keep the 1-D grid and the 64-bit division sequences.
struct TailUnit { int f, ty, gx; };
__device__ __forceinline__ bool tail_unit(const Geom &g, int n_frames, int map, TailUnit &u, int waves, int wave) {
    const int groups_x = (g.W + 63) >> 6, per_frame = groups_x * (g.H >> 3);
    u.f = 0, u.ty = 0, u.gx = 0;   // false: a valid unit all the same (callers that go on past the barrier store nothing)
    if (map == 0) {
        const int s0 = (int)blockIdx.x * waves + wave, s = min(s0, per_frame - 1);
        u.f = (int)blockIdx.y;
        u.ty = (int)div_magic((uint32_t)s, g.gx_magic);
        u.gx = s - u.ty * groups_x;
        return s0 < per_frame;
    }
    const long long total = (long long)per_frame * n_frames;
    long long L = (long long)blockIdx.x * waves + wave;
    long long id;
    if (map == 1) {
        const long long chunk = (total + 7) >> 3;
        id = (L & 7) * chunk + (L >> 3);
        if ((L >> 3) >= chunk || id >= total) return false;
    } else if (map == 2) {
        const int cs = (per_frame + 7) >> 3;
        const long long i = L >> 3;
        const int s = (int)(L & 7) * cs + (int)(i % cs);
        if (s >= per_frame || i / cs >= n_frames) return false;
        id = (i / cs) * per_frame + s;
    } else {
        return false;
    }
    u.f = (int)(id / per_frame);
    const int s = (int)(id - (long long)u.f * per_frame);
    u.ty = s / groups_x;
    u.gx = s - u.ty * groups_x;
    return true;
}
inline unsigned tail_grid(const Geom &g, int n_frames, int map) {
    const long long per_frame = (long long)((g.W + 63) >> 6) * (g.H >> 3);
    if (map == 1) return (unsigned)(8 * ((per_frame * n_frames + 7) >> 3));
    if (map == 2) return (unsigned)(8 * ((per_frame + 7) >> 3) * n_frames);
    return (unsigned)(per_frame * n_frames);
}
inline dim3 tail_grid_dim(const Geom &g, int n_frames, int map, int waves) {
    if (map == 0) {
        const unsigned per_frame = (unsigned)(((g.W + 63) >> 6) * (g.H >> 3));
        return dim3((per_frame + waves - 1) / waves, (unsigned)n_frames);
    }
    return dim3((tail_grid(g, n_frames, map) + waves - 1) / waves);
}

// The coefficient planes (k_fdct) and the decoded frames (k_idct) are written once and
// never read back by the kernel that writes them: non-temporal stores keep them from
// displacing the reference frame and the pixel rows in L2 (k_fdct 0.49 -> 0.42 ms at
// 29 x 1080p).  Non-temporal *loads* of the pixel rows were measured too and lose
// (0.52 ms): neighbouring lanes' unaligned dword loads share cache lines.
#ifdef VCS_EXP_PLAIN_STORES   // experiment builds only
#define VCS_NT_STORE(v, p) (*(p) = (v))
constexpr int kBufNt = 0;
#else
#define VCS_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
constexpr int kBufNt = 2;   // raw buffer aux: bit 1 = nt on gfx940+ (what __builtin_nontemporal_store sets)
#endif
template <class T> struct Vec2 { typedef T type __attribute__((ext_vector_type(2))); };

// --------------------------------------------------------------------------
// k_fdct: MODE 0 = DCTCompressor.compress on a BGR image (DCTcompressor.py:
//         49-74); MODE 1 = the encoder tail for a P frame: motion
//         compensation + wrap-around residual + compress, fused
//         (encoder.py:58-66).  grid = tail_grid() strips.
// DQ: D[64] (unused here: kDct8), Q[3][64], then 1/Q[3][64] (correctly rounded), float64, in HBM.
// --------------------------------------------------------------------------
template <int MODE, int CM, bool FASTDIV>
__global__ __launch_bounds__(64 * kFdctWaves) void k_fdct(
    const uint8_t *__restrict__ imgs, size_t img_stride, const uint8_t *__restrict__ ref,
    const int16_t *__restrict__ mvs, Geom g, const double *__restrict__ DQ,
    typename CoeffT<CM>::type *__restrict__ planes, size_t planes_stride, int n_frames, int map) {
    // s_t: stage-1 result, [tile][row][col]; s_o: the quantised strip parked for the
    // row-contiguous store, [tile][row][col] with padded rows.  They share one buffer:
    // every lane has its row of s_t in registers before the first s_o write (LDS
    // operations of one wave execute in order).
    static_assert(64 * kOutPad * sizeof(typename CoeffT<CM>::type) <= 8 * kTilePad * sizeof(double),
                  "s_o must fit in s_t");
    // A workgroup is kFdctWaves waves, one strip each; they have nothing in common but the
    // quantiser table in LDS, staged once per workgroup instead of once per strip.
    __shared__ double s_tw[kFdctWaves][8 * kTilePad];
    __shared__ double s_q[(FASTDIV ? 48 : 24) * kOutPad];  // Q[3][8][8] (then 1/Q), rows padded to 9
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *const s_t = s_tw[wave];
    typename CoeffT<CM>::type *const s_o = (typename CoeffT<CM>::type *)s_t;
    const int t = lane >> 3, j = lane & 7;
    TailUnit tu;
    const bool has_unit = tail_unit(g, n_frames, map, tu, kFdctWaves, wave);   // false: a valid unit, nothing stored
    // (wave-uniform by construction; said so, they and everything derived from them live in SGPRs)
    const int ty = __builtin_amdgcn_readfirstlane(tu.ty), gx = __builtin_amdgcn_readfirstlane(tu.gx),
              f = __builtin_amdgcn_readfirstlane(tu.f);
    const int x = gx * 64 + lane;  // stage-1 pixel column of this lane
    const uint8_t *__restrict__ img = imgs + (size_t)f * img_stride;
    const int16_t *__restrict__ mv = MODE ? mvs + (size_t)f * g.N * 2 : nullptr;
    // ---- fetch column x of the 8-row strip (all loads unconditional) ---------
    // The block's vector and the frame's pixels are requested BEFORE the quantiser table is staged: the
    // barrier behind the staging waits for every outstanding load anyway, so the three round trips
    // (table, vector, then the predicted pixels that need the vector) become two.  Measured on the tail
    // (experiment builds): the predicted-pixel loads cost 0.038 ms of its 0.274, 0.015 of them the wait
    // for the vector in front of them.
    const int xl = min(x, g.W - 1);
    const bool one_block = MODE == 1 && (g.bs & 7) == 0;   // uniform: one vector for the column's 8 rows
    uint32_t px[8], pp[8], m = 0u;
    if (one_block) m = pred_mv(mv, g, ty * 8, xl);
    // The table is staged by the workgroup's first wave alone, before its own pixel loads; the other waves
    // request their pixels at once and meet it at a barrier that is LDS traffic only (s_waitcnt lgkmcnt(0) +
    // s_barrier, not __syncthreads, whose fence would wait for the pixels too): their predicted pixels are
    // requested as soon as the vector is back (loads return in order), the pixels still on their way from
    // the Infinity Cache.
    const double *__restrict__ Qg = DQ + 64;
    constexpr int kQN = FASTDIV ? 384 : 192;
    if (wave == 0 || kFdctWaves == 1) {
#pragma unroll
        for (int it = 0; it < kQN / 64; it++) {
            const int e = lane + 64 * it;
            s_q[(e >> 3) * kOutPad + (e & 7)] = Qg[e];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) px[k] = load_px_o(img + (size_t)(ty * 8 + k) * g.pitch, 3u * (uint32_t)xl);
    // (Q and 1/Q are in LDS: their reads must not share vmcnt with the stores)
    if (kFdctWaves > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // s_q complete (every wave arrives exactly once)
    else wave_lds_fence();
    if (one_block) {
#pragma unroll
        for (int k = 0; k < 8; k++) pp[k] = pred_px_m(ref, g, ty * 8 + k, xl, m);
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) pp[k] = MODE == 1 ? pred_px(ref, mv, g, ty * 8 + k, xl) : 0u;
    }
    wave_lds_fence();
    double X[3][8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // residual = cur - pred, uint8 wrap (motion.py:39); MODE 0: pred = 0
        const uint32_t rs = sub_u8x4(px[k], pp[k]);
        int Y, Cr, Cb;
        bgr2ycc(rs & 255, (rs >> 8) & 255, (rs >> 16) & 255, Y, Cr, Cb);  // DCTcompressor.py:55
        X[0][k] = (double)(Y - 128);  // :57-59
        X[1][k] = (double)(Cr - 128);
        X[2][k] = (double)(Cb - 128);
    }

    const size_t plane = (size_t)g.H * g.W;
    using OutT = typename CoeffT<CM>::type;
    OutT *__restrict__ out = planes + (size_t)f * planes_stride;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        // stage 1: T = D * X (DCTcompressor.py:113), lane holds column j
#pragma unroll
        for (int i = 0; i < 8; i++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) acc = __builtin_fma(kDct8[i * 8 + k], X[ch][k], acc);
            s_t[t * kTilePad + i * 8 + j] = acc;
        }
        wave_lds_fence();
        // stage 2: R = T * D^T (:114), lane (t, i = j) holds row i of T
        double T[8];
#pragma unroll
        for (int k = 0; k < 8; k++) T[k] = s_t[t * kTilePad + j * 8 + k];
        wave_lds_fence();  // s_o overwrites s_t
        {
            const double *q = s_q + (ch * 8 + j) * kOutPad;
            const double *rq = q + 24 * kOutPad;  // correctly rounded 1 / Q
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 8; k++) acc = __builtin_fma(T[k], kDct8[jj * 8 + k], acc);
                // :71 np.true_divide; parked for the row-contiguous store below
                s_o[(t * 8 + j) * kOutPad + jj] = coeff_pack<CM>(
                    FASTDIV ? div_markstein(acc, q[jj], rq[jj]) : acc / q[jj]);
            }
        }
        wave_lds_fence();
        // row-contiguous store: one wave instruction covers two whole 64-column rows
        // of the strip (lane l -> columns 2*(l%32), +1 of row 2*s + l/32) instead of
        // 64 scattered 16-byte pieces -- the planes are 80 % of the kernel's HBM bytes
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const int e = s4 * 64 + lane, row = e >> 5, c2 = e & 31;
            const int tt = c2 >> 2, jj = (c2 & 3) * 2;
            const int xc = gx * 64 + 2 * c2;
            if (xc < g.W && has_unit) {
                const OutT v0 = s_o[(tt * 8 + row) * kOutPad + jj];
                const OutT v1 = s_o[(tt * 8 + row) * kOutPad + jj + 1];
                // uniform plane base + 32-bit element offset (a 4K plane is 8.3 M elements)
                OutT *o = (out + ch * plane) + (uint32_t)((ty * 8 + row) * g.W + xc);
                typename Vec2<OutT>::type pr;
                pr.x = v0, pr.y = v1;
                VCS_NT_STORE(pr, (typename Vec2<OutT>::type *)o);
            }
        }
        wave_lds_fence();  // the next channel's s_t overwrites s_o
    }
}

// --------------------------------------------------------------------------
// k_idct: DCTCompressor.decompress (DCTcompressor.py:76-93); MODE 1 adds the
// motion-compensated prediction with uint8 wrap (decoder.py:52-69).
// --------------------------------------------------------------------------
#ifdef VCS_EXP_IDCT_WPE   // experiment builds: occupancy hint for k_idct
#define IDCT_WPE __attribute__((amdgpu_waves_per_eu(VCS_EXP_IDCT_WPE, 8)))
#else
#define IDCT_WPE
#endif
// ONEBLK (host: MODE == 1 and bs % 8 == 0, every BASELINE configuration): the 8 pixels a lane
// reconstructs share one motion block -- one vector load, issued first
template <int MODE, int CM, bool ONEBLK = false>
__global__ __launch_bounds__(64 * kIdctWaves) IDCT_WPE void k_idct(
    const typename CoeffT<CM>::type *__restrict__ planes, size_t planes_stride,
    const uint8_t *__restrict__ ref, const int16_t *__restrict__ mvs, Geom g,
    const double *__restrict__ DQ, uint8_t *__restrict__ outs, size_t out_stride, int n_frames,
    int map) {
    __shared__ double s_tw[kIdctWaves][8 * kTilePad];   // one strip per wave, as in k_fdct
    __shared__ double s_q[24 * kOutPad];  // Q[3][8][8], rows padded to 9
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *const s_t = s_tw[wave];
    const int t = lane >> 3, j = lane & 7;
    for (int e = threadIdx.x; e < 192; e += 64 * kIdctWaves) s_q[(e >> 3) * kOutPad + (e & 7)] = DQ[64 + e];
    TailUnit tu;
    const bool has_unit = tail_unit(g, n_frames, map, tu, kIdctWaves, wave);
    if (kIdctWaves > 1) __syncthreads();
    if (!has_unit) return;
    const int ty = __builtin_amdgcn_readfirstlane(tu.ty), gx = __builtin_amdgcn_readfirstlane(tu.gx),
              f = __builtin_amdgcn_readfirstlane(tu.f);
    const int x = gx * 64 + lane;
    const int xl = min(x, g.W - 1);  // lanes past the frame edge: clamped loads, no stores
    const size_t plane = (size_t)g.H * g.W;
    const typename CoeffT<CM>::type *__restrict__ in = planes + (size_t)f * planes_stride;
    const int16_t *__restrict__ mv = MODE ? mvs + (size_t)f * g.N * 2 : nullptr;
    // ---- every global load up front, branch-free ----------------------------------
    // stage-2 ownership: row j of tile t -> 8 pixels starting at column xo
    const int y = ty * 8 + j, xo = min(gx * 64 + t * 8, g.W - 8);
    // The block's vector first, the planes behind it, the 8 predicted pixels (which need the vector,
    // not the planes: loads return in order) behind those -- and nothing that could move them: the
    // kernel has no branch after this point (lanes past the frame edge store out of the buffer's
    // range, which the hardware drops), because hipcc sinks loads into the block of their first use;
    // with the early return of the first version the 16 vector / pixel loads sat behind the whole
    // IDCT and every wave ended in two dependent round trips (0.345 -> 0.3xx ms per 29 frames).
    const uint32_t m8 = ONEBLK ? pred_mv(mv, g, y, xo) : 0u;
    // (buffer loads: one lane offset for all 24, the row / channel offsets are scalar -- the flat form
    // spends a 64-bit vector add and a register pair per load)
    double X[3][8];
    {
        using T = typename CoeffT<CM>::type;
        const uint32_t esz = (uint32_t)sizeof(T);
        const __amdgpu_buffer_rsrc_t r_in = buf_rsrc(in, (uint32_t)min((size_t)0xFFFFFFF0u, 3 * plane * esz));
        const uint32_t voff = (uint32_t)xl * esz;
        // the planes are read once: non-temporal loads (with four strips per workgroup: 0.315 -> 0.298 ms;
        // either alone: -2 / -3 %).  VCS_EXP_PLAIN_STORES builds drop the hint with the stores'.
        constexpr int kAux = kBufNt;
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t soff = ((uint32_t)ch * (uint32_t)plane + (uint32_t)(ty * 8 + k) * (uint32_t)g.W) * esz;
                if constexpr (CM == 0) {
                    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
                    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r_in, (int)voff, (int)soff, kAux);
                    X[ch][k] = __builtin_bit_cast(double, v);
                } else if constexpr (CM == 1) {
                    X[ch][k] = (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_in, (int)voff, (int)soff, kAux));
                } else {
                    X[ch][k] = (double)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(r_in, (int)voff, (int)soff, kAux);
                }
            }
    }
    uint32_t pp[8];
#pragma unroll
    for (int jj = 0; jj < 8; jj++)
        pp[jj] = MODE != 1 ? 0u : ONEBLK ? pred_px_m(ref, g, y, xo + jj, m8) : pred_px(ref, mv, g, y, xo + jj);
    __builtin_amdgcn_sched_barrier(0);
    wave_lds_fence();

    int ycc[3][8];  // row i = j of the tile, 8 pixels, after stage 2
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        // dequantise (:86 np.multiply), column j of the tile
#pragma unroll
        for (int k = 0; k < 8; k++) X[ch][k] = X[ch][k] * s_q[(ch * 8 + k) * kOutPad + j];
        // stage 1: T = D^T * X (:119)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) acc = __builtin_fma(kDct8[k * 8 + i], X[ch][k], acc);
            s_t[t * kTilePad + i * 8 + j] = acc;
        }
        wave_lds_fence();
        double T[8];
#pragma unroll
        for (int k = 0; k < 8; k++) T[k] = s_t[t * kTilePad + j * 8 + k];
        wave_lds_fence();
        // stage 2: R = T * D (:120); truncating uint8 store (:88), + 128 (:89)
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) acc = __builtin_fma(T[k], kDct8[k * 8 + jj], acc);
            ycc[ch][jj] = ((int)acc + 128) & 255;
        }
    }
    // 8 pixels = 24 bytes, packed into three 8-byte stores (24*t is 8-byte aligned)
    uint8_t b[24];
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        int B, G, R;
        ycc2bgr(ycc[0][jj], ycc[1][jj], ycc[2][jj], B, G, R);  // :92
        const uint32_t p = pp[jj];                             // decoder.py:57 uint8 wrap add
        b[3 * jj + 0] = (uint8_t)(B + (int)(p & 255));
        b[3 * jj + 1] = (uint8_t)(G + (int)((p >> 8) & 255));
        b[3 * jj + 2] = (uint8_t)(R + (int)((p >> 16) & 255));
    }
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t r_out = buf_rsrc(outs + (size_t)f * out_stride, (uint32_t)(g.H * g.pitch));
    u32x2 v[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        v[q].x = b[8 * q] | (b[8 * q + 1] << 8) | (b[8 * q + 2] << 16) | ((uint32_t)b[8 * q + 3] << 24);
        v[q].y = b[8 * q + 4] | (b[8 * q + 5] << 8) | (b[8 * q + 6] << 16) | ((uint32_t)b[8 * q + 7] << 24);
    }
    if (gx * 64 + 64 <= g.W) {   // (uniform) a whole strip: its 8 rows x 192 bytes leave as 16-byte pieces, 12 in a row
        // Round 3 stored a lane's 24 bytes as three 8-byte pieces: one store instruction wrote 8 of every 24 bytes
        // of a row, the next the following 8 -- WRITE_SIZE 294 MB for 180 MB of frames.  Through LDS (s_t is
        // free: every lane has its row of T in registers) a row's 192 bytes are written by 12 adjacent lanes.
        uint8_t *s_b = (uint8_t *)s_t;
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 3; q++) *(u32x2 *)(s_b + j * 192 + t * 24 + 8 * q) = v[q];
        wave_lds_fence();
        const uint32_t row0 = (uint32_t)(ty * 8) * (uint32_t)g.pitch + (uint32_t)gx * 192u;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int e = lane + 64 * it;
            if (e < 96) {
                const int row = e / 12, piece = e - row * 12;
                const u32x4_t w = *(const u32x4_t *)(s_b + row * 192 + piece * 16);
                __builtin_amdgcn_raw_buffer_store_b128(w, r_out, (int)(row0 + (uint32_t)row * (uint32_t)g.pitch + 16u * (uint32_t)piece), 0, kBufNt);
            }
        }
    } else {
        // a strip that crosses the right edge (W not a multiple of 64): tiles past the edge get an offset beyond the
        // frame, dropped by the buffer's range check
        const uint32_t oo = gx * 64 + t * 8 < g.W ? (uint32_t)(y * g.pitch + 3 * xo) : 0xFFFFFF00u;
#pragma unroll
        for (int q = 0; q < 3; q++) __builtin_amdgcn_raw_buffer_store_b64(v[q], r_out, (int)(oo + 8u * q), 0, kBufNt);
    }
}

// --------------------------------------------------------------------------
// Stand-alone pieces of the class API (not on the fused path).
// --------------------------------------------------------------------------
// MotionProcessor.reconstruct_from_motion_vectors (motion.py:42-69): one
// wavefront per block copies bs rows of 3*bs bytes.  `pred` is pre-zeroed.
__global__ __launch_bounds__(64) void k_mc(const uint8_t *__restrict__ ref,
                                           const int16_t *__restrict__ mv,
                                           const int32_t *__restrict__ coords, Geom g,
                                           uint8_t *__restrict__ pred) {
    const int k = blockIdx.x, rowb = 3 * g.bs;
    const int x = coords ? coords[2 * k] : (k % g.nbx) * g.bs;
    const int y = coords ? coords[2 * k + 1] : (k / g.nbx) * g.bs;
    const int sx = x + mv[2 * k], sy = y + mv[2 * k + 1];  // :60-61
    for (int e = threadIdx.x; e < g.bs * rowb; e += 64) {
        int r = e / rowb, b = e - r * rowb;
        pred[(size_t)(y + r) * g.pitch + (size_t)(3 * x + b)] =
            ref[(size_t)(sy + r) * g.pitch + (size_t)(3 * sx + b)];
    }
}

// 4:2:0 chroma subsampling (reference ChromaSubsampling/chroma.py:8-21):
// BGR -> YCrCb (the same 8-bit conversion as k_fdct), a normalised 2x2 box filter over Cr and
// Cb as cv2.boxFilter(ksize=(2, 2)) computes it -- anchor (1, 1): output (y, x) averages rows
// y-1..y and columns x-1..x, BORDER_REFLECT_101 (index -1 -> 1), sum * 0.25 rounded half to
// even -- and the samples at even rows / columns (`[::2, ::2]`).  Third-party semantics
// restated from OpenCV 4.x (parity unpinned, like cvtColor).  One thread per chroma sample;
// the full-resolution Y plane is written by the threads that own its four pixels.
__global__ __launch_bounds__(256) void k_chroma420(const uint8_t *__restrict__ bgr, int H, int W,
                                                   uint8_t *__restrict__ Yp, uint8_t *__restrict__ crs,
                                                   uint8_t *__restrict__ cbs) {
    const int Hc = (H + 1) >> 1, Wc = (W + 1) >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Hc * Wc) return;
    const int i = idx / Wc, j = idx - i * Wc;
    const size_t pitch = (size_t)3 * W;
    int scr = 0, scb = 0;
#pragma unroll
    for (int dy = -1; dy <= 0; dy++)
#pragma unroll
        for (int dx = -1; dx <= 0; dx++) {
            int y = 2 * i + dy, x = 2 * j + dx;
            y = y < 0 ? min(1, H - 1) : y;   // reflect-101 (a 1-pixel-high image repeats its row)
            x = x < 0 ? min(1, W - 1) : x;
            const uint8_t *p = bgr + (size_t)y * pitch + (size_t)(3 * x);
            int Yv, Cr, Cb;
            bgr2ycc(p[0], p[1], p[2], Yv, Cr, Cb);
            scr += Cr;
            scb += Cb;
        }
    // sum / 4 rounded half to even: sums with remainder 2 go to the even neighbour
    auto rne4 = [](int s) { const int q = s >> 2, r = s & 3; return q + (r == 3 || (r == 2 && (q & 1))); };
    crs[idx] = (uint8_t)rne4(scr);
    cbs[idx] = (uint8_t)rne4(scb);
    if (Yp) {
#pragma unroll
        for (int dy = 0; dy <= 1; dy++)
#pragma unroll
            for (int dx = 0; dx <= 1; dx++) {
                const int y = 2 * i + dy, x = 2 * j + dx;
                if (y < H && x < W) {
                    const uint8_t *p = bgr + (size_t)y * pitch + (size_t)(3 * x);
                    int Yv, Cr, Cb;
                    bgr2ycc(p[0], p[1], p[2], Yv, Cr, Cb);
                    Yp[(size_t)y * W + x] = (uint8_t)Yv;
                }
            }
    }
}

// MotionProcessor.get_residuals (motion.py:38-40) and the decoder's wrap-add
// (decoder.py:57,60): out = a -/+ b per byte; 16 bytes per lane per step.
template <int SIGN>
__global__ __launch_bounds__(256) void k_bytes_addsub(const uint8_t *__restrict__ a,
                                                      const uint8_t *__restrict__ b,
                                                      size_t n, uint8_t *__restrict__ out) {
    const size_t nv = n >> 4;
    const uint4 *a4 = (const uint4 *)a, *b4 = (const uint4 *)b;
    uint4 *o4 = (uint4 *)out;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
         i += (size_t)gridDim.x * blockDim.x) {
        uint4 va = a4[i], vb = b4[i], vo;
        if (SIGN < 0) {
            vo.x = sub_u8x4(va.x, vb.x); vo.y = sub_u8x4(va.y, vb.y);
            vo.z = sub_u8x4(va.z, vb.z); vo.w = sub_u8x4(va.w, vb.w);
        } else {
            vo.x = add_u8x4(va.x, vb.x); vo.y = add_u8x4(va.y, vb.y);
            vo.z = add_u8x4(va.z, vb.z); vo.w = add_u8x4(va.w, vb.w);
        }
        o4[i] = vo;
    }
    if (blockIdx.x == 0)
        for (size_t i = (nv << 4) + threadIdx.x; i < n; i += blockDim.x)
            out[i] = SIGN < 0 ? (uint8_t)(a[i] - b[i]) : (uint8_t)(a[i] + b[i]);
}

}  // namespace vcs
