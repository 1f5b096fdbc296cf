/*
 * oracle/vcs_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, CPU restatement of the inter-frame encode hot path of
 * miatang13/VCS-h264 (reference mounted read-only at /root/reference).  It is
 * the checker for the HIP kernels and the `cpu_baseline` leg of bench.py; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it.
 * The product path (vcs-h264_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED for everything except the BGR<->YCrCb conversion.
 *   - oracle/gen_golden.py imports the reference's own motion.py /
 *     DCTcompressor.py / encoder.py / decoder.py (unmodified, with the cv2
 *     stand-in of oracle/cv2_standin) and writes tests/golden/ *.npz;
 *     tests/test_oracle_golden.py holds this file bit-exact to those vectors:
 *     motion vectors, MC image, residual, float64 coefficient planes, decoded
 *     pixels.
 *   - The colour conversion is OpenCV code (third-party, absent from the
 *     reference tree, version unpinned): restated from OpenCV 4.x
 *     imgproc/src/color_yuv.simd.hpp (8-bit fixed-point path).  That piece is
 *     "parity unpinned" against a real OpenCV binary.
 *
 * Float determinism: the reference's np.matmul on 8x8 float64 operands
 * (numpy 2.2.6 / scipy-openblas 0.3.29, Haswell/SkylakeX dgemm kernels) was
 * measured in the build container to be bit-identical to a k-ascending chain
 * of fused multiply-adds starting from +0.0 (2000/2000 random blocks, both
 * matmul stages).  mm8_*() below restate exactly that chain with fma(); the
 * file must be compiled with -ffp-contract=off so nothing else is fused.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CH 3 /* motion.py:9 CHANNELS */

/* ------------------------------------------------------------------------ */
/* Block grid: motion.py:74-98 (_split_frame_into_mblocks).  Raster order, i
 * (rows) outer, j (cols) inner, only full blocks; coords are [x, y].        */
int vcso_num_blocks(int H, int W, int bs, int *nby, int *nbx) {
    int by = H / bs, bx = W / bs;
    if (nby) *nby = by;
    if (nbx) *nbx = bx;
    return by * bx;
}

void vcso_block_coords(int H, int W, int bs, int32_t *coords /* N x [x,y] */) {
    int nby, nbx;
    vcso_num_blocks(H, W, bs, &nby, &nbx);
    for (int bi = 0; bi < nby; bi++)
        for (int bj = 0; bj < nbx; bj++) {
            coords[2 * (bi * nbx + bj) + 0] = bj * bs; /* x  motion.py:95 */
            coords[2 * (bi * nbx + bj) + 1] = bi * bs; /* y */
        }
}

/* ------------------------------------------------------------------------ */
/* motion.py:100-154 (_find_match) + :156-161 (_get_motion_vector) for one
 * block at (x, y).  Returns 1 when the static early-out was taken.          */
static int find_match(const uint8_t *cur, const uint8_t *ref, int H, int W,
                      int bs, int win, int step, uint32_t thresh, int x, int y,
                      int *best_x, int *best_y, uint64_t *best_cost) {
    const size_t pitch = (size_t)W * CH;
    /* motion.py:109-111: np.sum(np.abs(cv2.subtract(ref_blk, cur_blk))):
     * cv2.subtract saturates at 0 for uint8, so this is sum(max(r-c,0)).   */
    uint64_t diff = 0;
    for (int r = 0; r < bs; r++) {
        const uint8_t *pr = ref + (size_t)(y + r) * pitch + (size_t)x * CH;
        const uint8_t *pc = cur + (size_t)(y + r) * pitch + (size_t)x * CH;
        for (int b = 0; b < bs * CH; b++)
            diff += pr[b] > pc[b] ? (uint64_t)(pr[b] - pc[b]) : 0;
    }
    if (diff <= thresh) { /* motion.py:113-116 */
        *best_x = x;
        *best_y = y;
        *best_cost = diff;
        return 1;
    }
    /* motion.py:118-152 */
    uint64_t best = 9999999999ULL; /* :120 */
    int bx = 0, by = 0;            /* :102 best_coord = [0, 0] */
    int i_min = y - win > 0 ? y - win : 0; /* :125 */
    int j_min = x - win > 0 ? x - win : 0; /* :126 */
    int i_max = y + win < H ? y + win : H; /* :128 */
    int j_max = x + win < W ? x + win : W; /* :129 */
    for (int i = i_min; i < i_max; i += step) { /* :133 */
        if (i + bs >= i_max) continue;          /* :136 */
        for (int j = j_min; j < j_max; j += step) { /* :138 */
            if (j + bs >= j_max) continue;          /* :139 */
            /* :146 np.sum(np.abs(ref_block - block)) on uint8 operands: the
             * subtraction wraps mod 256 and abs is a no-op.                 */
            uint64_t d = 0;
            for (int r = 0; r < bs; r++) {
                const uint8_t *pr = ref + (size_t)(i + r) * pitch + (size_t)j * CH;
                const uint8_t *pc = cur + (size_t)(y + r) * pitch + (size_t)x * CH;
                for (int b = 0; b < bs * CH; b++)
                    d += (uint8_t)(pr[b] - pc[b]);
            }
            if (d < best) { /* :149 strict <, first candidate wins ties */
                best = d;
                bx = j;
                by = i;
            }
        }
    }
    *best_x = bx;
    *best_y = by;
    *best_cost = best;
    return 0;
}

/* motion.py:20-36 (process_motion_prediction).  mv: N x [dx, dy] int16.
 * cost_out (nullable): per block, the static-test sum (static blocks) or the
 * winning wrap-around cost (searched blocks; 9999999999 if no candidate).   */
int vcso_me(const uint8_t *cur, const uint8_t *ref, int H, int W, int bs,
            int win, int step, uint32_t thresh, int16_t *mv,
            uint32_t *n_static, uint64_t *cost_out) {
    int nby, nbx;
    int N = vcso_num_blocks(H, W, bs, &nby, &nbx);
    if (step <= 0 || bs <= 0) return -1;
    uint32_t ns = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : ns)
    for (int k = 0; k < N; k++) {
        int x = (k % nbx) * bs, y = (k / nbx) * bs;
        int bx, by;
        uint64_t c;
        ns += (uint32_t)find_match(cur, ref, H, W, bs, win, step, thresh, x, y,
                                   &bx, &by, &c);
        mv[2 * k + 0] = (int16_t)(bx - x); /* motion.py:158 dx */
        mv[2 * k + 1] = (int16_t)(by - y); /* motion.py:159 dy */
        if (cost_out) cost_out[k] = c;
    }
    if (n_static) *n_static = ns;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Extension searches -- NOT in the reference, parity unpinned: this is the
 * definition the HIP kernel k_me_ext (csrc/me_ext.hpp) is checked against.
 *   search 0 = the reference's grid (motion.py:132-140), 1 = three-step search:
 *     centre = the co-located position; S = largest power of two with 2S - 1 <= win; per
 *     stage the 8 neighbours (dy, dx) in raster order at distance S that keep the block inside
 *     the frame; move to the cheapest if STRICTLY cheaper than the centre (first wins ties);
 *     S /= 2 down to 1.
 *   cost 0 = sum (r - c) mod 256 (motion.py:146), 1 = sum |r - c|.
 * Static early-out as in find_match.                                                     */
static uint64_t block_cost(const uint8_t *cur, const uint8_t *ref, size_t pitch, int bs, int x,
                           int y, int j, int i, int cost_kind) {
    uint64_t d = 0;
    for (int r = 0; r < bs; r++) {
        const uint8_t *pr = ref + (size_t)(i + r) * pitch + (size_t)j * CH;
        const uint8_t *pc = cur + (size_t)(y + r) * pitch + (size_t)x * CH;
        for (int b = 0; b < bs * CH; b++)
            d += cost_kind ? (uint64_t)(pr[b] > pc[b] ? pr[b] - pc[b] : pc[b] - pr[b])
                           : (uint64_t)(uint8_t)(pr[b] - pc[b]);
    }
    return d;
}

int vcso_me_ext(const uint8_t *cur, const uint8_t *ref, int H, int W, int bs, int win, int step,
                uint32_t thresh, int search, int cost_kind, int16_t *mv, uint32_t *n_static,
                uint64_t *cost_out) {
    int nby, nbx;
    int N = vcso_num_blocks(H, W, bs, &nby, &nbx);
    if (step <= 0 || bs <= 0 || win < 1) return -1;
    const size_t pitch = (size_t)W * CH;
    uint32_t ns = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : ns)
    for (int k = 0; k < N; k++) {
        const int x = (k % nbx) * bs, y = (k / nbx) * bs;
        uint64_t diff = 0;
        for (int r = 0; r < bs; r++) {
            const uint8_t *pr = ref + (size_t)(y + r) * pitch + (size_t)x * CH;
            const uint8_t *pc = cur + (size_t)(y + r) * pitch + (size_t)x * CH;
            for (int b = 0; b < bs * CH; b++) diff += pr[b] > pc[b] ? (uint64_t)(pr[b] - pc[b]) : 0;
        }
        int bx = 0, by = 0;
        uint64_t best = 9999999999ULL;
        if (diff <= thresh) {
            ns++;
            bx = x; by = y; best = diff;
        } else if (search == 0) {
            int i_min = y - win > 0 ? y - win : 0, j_min = x - win > 0 ? x - win : 0;
            int i_max = y + win < H ? y + win : H, j_max = x + win < W ? x + win : W;
            for (int i = i_min; i < i_max; i += step) {
                if (i + bs >= i_max) continue;
                for (int j = j_min; j < j_max; j += step) {
                    if (j + bs >= j_max) continue;
                    uint64_t d = block_cost(cur, ref, pitch, bs, x, y, j, i, cost_kind);
                    if (d < best) { best = d; bx = j; by = i; }
                }
            }
        } else {
            int ci = y, cj = x, S = 1;
            best = block_cost(cur, ref, pitch, bs, x, y, cj, ci, cost_kind);
            while (2 * (2 * S) - 1 <= win) S *= 2;
            /* the kernel parks rows / columns [pos - win, pos + win + bs): same validity here */
            const int lo_i = y - win > 0 ? y - win : 0, lo_j = x - win > 0 ? x - win : 0;
            const int hi_i = (y + win + bs < H ? y + win + bs : H) - bs;
            const int hi_j = (x + win + bs < W ? x + win + bs : W) - bs;
            for (; S >= 1; S >>= 1) {
                uint64_t sb = ~0ULL;
                int si = 0, sj = 0;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dy && !dx) continue;
                        const int i = ci + dy * S, j = cj + dx * S;
                        if (i < lo_i || i > hi_i || j < lo_j || j > hi_j) continue;
                        uint64_t d = block_cost(cur, ref, pitch, bs, x, y, j, i, cost_kind);
                        if (d < sb) { sb = d; si = i; sj = j; }
                    }
                if (sb != ~0ULL && sb < best) { best = sb; ci = si; cj = sj; }
            }
            bx = cj; by = ci;
        }
        mv[2 * k + 0] = (int16_t)(bx - x);
        mv[2 * k + 1] = (int16_t)(by - y);
        if (cost_out) cost_out[k] = best;
    }
    if (n_static) *n_static = ns;
    return 0;
}

/* motion.py:42-69 (reconstruct_from_motion_vectors).  coords nullable =
 * raster grid.  Output zero outside the blocks (:45-46).                    */
int vcso_mc(const uint8_t *ref, const int16_t *mv, const int32_t *coords,
            int n, int H, int W, int bs, uint8_t *pred) {
    const size_t pitch = (size_t)W * CH;
    int nbx = W / bs;
    memset(pred, 0, (size_t)H * pitch);
    for (int k = 0; k < n; k++) {
        int x = coords ? coords[2 * k] : (k % nbx) * bs;
        int y = coords ? coords[2 * k + 1] : (k / nbx) * bs;
        int sx = x + mv[2 * k], sy = y + mv[2 * k + 1]; /* :60-61 */
        if (x < 0 || y < 0 || x + bs > W || y + bs > H) return -1;
        if (sx < 0 || sy < 0 || sx + bs > W || sy + bs > H) return -1;
        for (int r = 0; r < bs; r++)
            memcpy(pred + (size_t)(y + r) * pitch + (size_t)x * CH,
                   ref + (size_t)(sy + r) * pitch + (size_t)sx * CH,
                   (size_t)bs * CH);
    }
    return 0;
}

/* motion.py:38-40 (get_residuals): uint8 wrap-around subtract.  Also the
 * decoder's `img + residuals` (decoder.py:57,60) with sign = +1.            */
void vcso_residual(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(a[i] - b[i]);
}
void vcso_wrap_add(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(a[i] + b[i]);
}

/* ------------------------------------------------------------------------ */
/* OpenCV 4.x 8-bit BGR<->YCrCb (cv2.cvtColor at DCTcompressor.py:55 and :92).
 * Third-party restatement, parity unpinned -- see the file header.          */
#define YSHIFT 14
#define DESCALE(v) (((v) + (1 << (YSHIFT - 1))) >> YSHIFT)
static inline uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void vcso_bgr2ycrcb_px(const uint8_t *bgr, uint8_t *ycc) {
    int B = bgr[0], G = bgr[1], R = bgr[2];
    int Y = DESCALE(R * 4899 + G * 9617 + B * 1868);
    int Cr = DESCALE((R - Y) * 11682 + (128 << YSHIFT));
    int Cb = DESCALE((B - Y) * 9241 + (128 << YSHIFT));
    ycc[0] = sat8(Y);
    ycc[1] = sat8(Cr);
    ycc[2] = sat8(Cb);
}
void vcso_ycrcb2bgr_px(const uint8_t *ycc, uint8_t *bgr) {
    int Y = ycc[0], Cr = ycc[1] - 128, Cb = ycc[2] - 128;
    bgr[0] = sat8(Y + DESCALE(Cb * 29049));
    bgr[1] = sat8(Y + DESCALE(Cb * -5636 + Cr * -11698));
    bgr[2] = sat8(Y + DESCALE(Cr * 22987));
}
void vcso_bgr2ycrcb(const uint8_t *bgr, size_t npx, uint8_t *ycc) {
    for (size_t i = 0; i < npx; i++) vcso_bgr2ycrcb_px(bgr + 3 * i, ycc + 3 * i);
}
void vcso_ycrcb2bgr(const uint8_t *ycc, size_t npx, uint8_t *bgr) {
    for (size_t i = 0; i < npx; i++) vcso_ycrcb2bgr_px(ycc + 3 * i, bgr + 3 * i);
}

/* ------------------------------------------------------------------------ */
/* 4:2:0 chroma subsampling, ChromaSubsampling/chroma.py:8-21: cvtColor, then
 * cv2.boxFilter(plane, ddepth=-1, ksize=(2,2)) and [::2, ::2].  boxFilter restated from
 * OpenCV 4.x (third party, unpinned): anchor (1,1) -> output (y,x) covers rows y-1..y and
 * columns x-1..x; BORDER_REFLECT_101 (index -1 -> 1); normalised: sum * 0.25 through
 * saturate_cast<uchar>(double) = rint (half to even).                                  */
int vcso_chroma420(const uint8_t *bgr, int H, int W, uint8_t *Y, uint8_t *crs, uint8_t *cbs) {
    const int Hc = (H + 1) / 2, Wc = (W + 1) / 2;
    for (int i = 0; i < Hc; i++)
        for (int j = 0; j < Wc; j++) {
            int scr = 0, scb = 0;
            for (int dy = -1; dy <= 0; dy++)
                for (int dx = -1; dx <= 0; dx++) {
                    int y = 2 * i + dy, x = 2 * j + dx;
                    if (y < 0) y = H > 1 ? 1 : 0;
                    if (x < 0) x = W > 1 ? 1 : 0;
                    uint8_t ycc[3];
                    vcso_bgr2ycrcb_px(bgr + ((size_t)y * W + x) * 3, ycc);
                    scr += ycc[1];
                    scb += ycc[2];
                }
            crs[(size_t)i * Wc + j] = (uint8_t)nearbyint(scr * 0.25);
            cbs[(size_t)i * Wc + j] = (uint8_t)nearbyint(scb * 0.25);
        }
    if (Y)
        for (size_t p = 0; p < (size_t)H * W; p++) {
            uint8_t ycc[3];
            vcso_bgr2ycrcb_px(bgr + 3 * p, ycc);
            Y[p] = ycc[0];
        }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* DCTcompressor.py:124-133 (_dctMatrix), same expression order as the Python
 * (math.sqrt / math.cos are libm sqrt / cos).                               */
void vcso_dct_matrix(int n, double *D) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            if (i == 0)
                D[i * n + j] = 1.0 / sqrt((double)n);
            else
                D[i * n + j] = sqrt(2.0 / (double)n) *
                               cos((double)((2 * j + 1) * i) * M_PI / (double)(2 * n));
        }
}

/* DCTcompressor.py:11-38: JPEG tables and the QF -> scale -> Q rule.
 * np.round is round-half-to-even == nearbyint() in the default mode.        */
static const int QY_TAB[64] = { /* :11-18 (note [1][5] is 48 in the reference) */
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  48,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const int QC_TAB[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

int vcso_make_q(double qf, double *Q /* 3 x 64 */) {
    double scale;
    if (qf < 50 && qf > 1) /* :30 */
        scale = 50 / qf;
    else if (qf < 100) /* :32 */
        scale = (100 - qf) / 50;
    else
        return -1; /* :35 the reference only prints; Q would be undefined */
    for (int i = 0; i < 64; i++) {
        double y = nearbyint(QY_TAB[i] * scale), c = nearbyint(QC_TAB[i] * scale);
        y = y < 1 ? 1 : y > 255 ? 255 : y; /* :36-38 np.clip(..., 1, 255) */
        c = c < 1 ? 1 : c > 255 ? 255 : c;
        Q[i] = y;
        Q[64 + i] = c;
        Q[128 + i] = c;
    }
    return 0;
}

/* C = A * B for 8x8 row-major float64: k-ascending FMA chain from +0.0 (the
 * measured behaviour of the reference's np.matmul, see the file header).
 * tA / tB select a transposed read of the operand.                          */
static inline void mm8(const double *A, int tA, const double *B, int tB, double *C) {
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            double acc = 0.0;
            for (int k = 0; k < 8; k++) {
                double a = tA ? A[k * 8 + i] : A[i * 8 + k];
                double b = tB ? B[j * 8 + k] : B[k * 8 + j];
                acc = fma(a, b, acc);
            }
            C[i * 8 + j] = acc;
        }
}

/* DCTcompressor.py:49-74 (compress), dct block size fixed at 8 (the Q tables
 * are 8x8; any other size raises in the reference).  H and W must
 * be multiples of 8 (cv2.resize is then the identity, :52-53).
 * planes: 3 planar H x W float64 [Y, Cr, Cb].  ycc_out nullable (H*W*3).    */
int vcso_compress(const uint8_t *bgr, int H, int W, const double *Q,
                  double *planes, uint8_t *ycc_out) {
    if (H % 8 || W % 8) return -1;
    double D[64];
    vcso_dct_matrix(8, D);
    const size_t plane = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (int ti = 0; ti < H / 8; ti++)
        for (int tj = 0; tj < W / 8; tj++) {
            double X[3][64], T[64], R[64];
            for (int r = 0; r < 8; r++)
                for (int c = 0; c < 8; c++) {
                    size_t p = (size_t)(ti * 8 + r) * W + (size_t)(tj * 8 + c);
                    uint8_t ycc[3];
                    vcso_bgr2ycrcb_px(bgr + 3 * p, ycc); /* :55 */
                    if (ycc_out) memcpy(ycc_out + 3 * p, ycc, 3);
                    for (int ch = 0; ch < 3; ch++) /* :57-59 int16 - 128 */
                        X[ch][r * 8 + c] = (double)((int)ycc[ch] - 128);
                }
            for (int ch = 0; ch < 3; ch++) {
                mm8(D, 0, X[ch], 0, T); /* :113 matmul(dctmat, matrix)   */
                mm8(T, 0, D, 1, R);     /* :114 matmul(result, dctmat.T) */
                for (int r = 0; r < 8; r++)
                    for (int c = 0; c < 8; c++) /* :71 true_divide, no rounding */
                        planes[ch * plane + (size_t)(ti * 8 + r) * W + tj * 8 + c] =
                            R[r * 8 + c] / Q[ch * 64 + r * 8 + c];
            }
        }
    return 0;
}

/* float64 -> uint8 store at DCTcompressor.py:88 (`result[...] = d` into a
 * uint8 array): truncation toward zero, then wrap mod 256 (SURVEY 2.1 probe:
 * -1.5 -> 255, 0.9999999 -> 0, 256.2 -> 0, 300.7 -> 44).                    */
static inline uint8_t f64_to_u8_wrap(double v) { return (uint8_t)(int64_t)v; }

/* DCTcompressor.py:76-93 (decompress).  pre_trunc (nullable): the float64
 * IDCT output before the uint8 store, 3 planar H x W.                       */
int vcso_decompress(const double *planes, int H, int W, const double *Q,
                    uint8_t *bgr, double *pre_trunc) {
    if (H % 8 || W % 8) return -1;
    double D[64];
    vcso_dct_matrix(8, D);
    const size_t plane = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (int ti = 0; ti < H / 8; ti++)
        for (int tj = 0; tj < W / 8; tj++) {
            uint8_t ycc[64][3];
            for (int ch = 0; ch < 3; ch++) {
                double X[64], T[64], R[64];
                for (int r = 0; r < 8; r++)
                    for (int c = 0; c < 8; c++) /* :86 np.multiply(block, Q) */
                        X[r * 8 + c] =
                            planes[ch * plane + (size_t)(ti * 8 + r) * W + tj * 8 + c] *
                            Q[ch * 64 + r * 8 + c];
                mm8(D, 1, X, 0, T); /* :119 matmul(dctmat.T, matrix) */
                mm8(T, 0, D, 0, R); /* :120 matmul(result, dctmat)   */
                for (int e = 0; e < 64; e++) {
                    if (pre_trunc)
                        pre_trunc[ch * plane + (size_t)(ti * 8 + e / 8) * W + tj * 8 + e % 8] = R[e];
                    /* :88 truncating store, :89 + 128 with uint8 wrap */
                    ycc[e][ch] = (uint8_t)(f64_to_u8_wrap(R[e]) + 128);
                }
            }
            for (int e = 0; e < 64; e++) { /* :91-92 dstack + YCrCb -> BGR */
                size_t p = (size_t)(ti * 8 + e / 8) * W + (size_t)(tj * 8 + e % 8);
                vcso_ycrcb2bgr_px(ycc[e], bgr + 3 * p);
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* encoder.py:49-70 (_process_P_frame): ME -> MC -> residual -> compress.
 * me_bs is the motion block size, the DCT is always 8x8 (the
 * north-star "16x16 ME + 8x8 DCT" is the composition MotionProcessor(16) ->
 * DCTCompressor(8)).  scratch-free: allocates two frames internally.        */
int vcso_encode_p(const uint8_t *cur, const uint8_t *ref, int H, int W,
                  int me_bs, int win, int step, uint32_t thresh,
                  const double *Q, int16_t *mv, double *planes,
                  uint32_t *n_static, uint8_t *res_out /* nullable */) {
    size_t F = (size_t)H * W * CH;
    int N = vcso_num_blocks(H, W, me_bs, 0, 0);
    uint8_t *pred = (uint8_t *)malloc(F), *res = (uint8_t *)malloc(F);
    if (!pred || !res) { free(pred); free(res); return -2; }
    int rc = vcso_me(cur, ref, H, W, me_bs, win, step, thresh, mv, n_static, 0);
    if (!rc) rc = vcso_mc(ref, mv, 0, N, H, W, me_bs, pred);
    if (!rc) {
        vcso_residual(cur, pred, F, res);
        if (res_out) memcpy(res_out, res, F);
        if (planes) rc = vcso_compress(res, H, W, Q, planes, 0);
    }
    free(pred);
    free(res);
    return rc;
}

/* decoder.py:52-69 (_reconstruct_P_frame + _fully_reconstruct).             */
int vcso_decode_p(const int16_t *mv, const double *planes, const uint8_t *ref,
                  int H, int W, int me_bs, const double *Q, uint8_t *out) {
    size_t F = (size_t)H * W * CH;
    int N = vcso_num_blocks(H, W, me_bs, 0, 0);
    uint8_t *pred = (uint8_t *)malloc(F), *dec = (uint8_t *)malloc(F);
    if (!pred || !dec) { free(pred); free(dec); return -2; }
    int rc = vcso_mc(ref, mv, 0, N, H, W, me_bs, pred);
    if (!rc) rc = vcso_decompress(planes, H, W, Q, dec, 0);
    if (!rc) vcso_wrap_add(pred, dec, F, out); /* decoder.py:57 */
    free(pred);
    free(dec);
    return rc;
}

/* np.round(plane) (round-half-to-even) -> int16: the `round` coefficient mode
 * (DCTCompression/dct.py:179 semantics).                  */
void vcso_round_i16(const double *in, size_t n, int16_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = (int16_t)nearbyint(in[i]);
}

/* ======================================================================== */
/* Intra prediction (BASELINE config 5):                   */
/* /root/reference/IntraframeCompression/intraframe.py + intramodes.py.      */
/* Every value the reference computes here is integer-valued (floor          */
/* divisions of integers held in float64), so the restatement is integer     */
/* arithmetic; outputs are float64 like the reference's.  NumPy 2 (NEP 50)   */
/* quirks that change results are marked QUIRK.                              */
/* ------------------------------------------------------------------------ */
static inline int fdiv(int a, int b) { /* Python floor division, b > 0 */
    int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}
/* 3*x//4 (intramodes.py:42,135): QUIRK -- when x is an element of a uint8
 * array, 3*x is evaluated in uint8 and wraps mod 256 (NEP 50); when the
 * neighbour row is one of the float64 stand-ins (np.ones(4)*...) it does not. */
static inline int three_q(int x, int is_u8) { return is_u8 ? ((3 * x) & 255) / 4 : (3 * x) / 4; }

/* The nine 4x4 predictors, intramodes.py:7-143.  u8 flags: which neighbour
 * arrays are real uint8 views (vs float64 stand-ins).                       */
static void pred4x4(int mode, int ul, const int *u, const int *ur, const int *l,
                    int u_u8, int ur_u8, int l_u8, int p[4][4]) {
    int r, c;
    switch (mode) {
    case 0: for (r = 0; r < 4; r++) for (c = 0; c < 4; c++) p[r][c] = u[c]; break;   /* :7-11  */
    case 1: for (r = 0; r < 4; r++) for (c = 0; c < 4; c++) p[r][c] = l[r]; break;   /* :13-17 */
    case 2: {                                                                        /* :19-23 */
        int s = 0; /* QUIRK: u + l on two uint8 arrays wraps per element (:21) */
        for (c = 0; c < 4; c++) s += (u_u8 && l_u8) ? ((u[c] + l[c]) & 255) : (u[c] + l[c]);
        for (r = 0; r < 4; r++) for (c = 0; c < 4; c++) p[r][c] = s / 8;
        break; }
    case 3: {                                                                        /* :25-43 */
        int a = u[0]/4 + u[1]/2 + u[2]/4, b = u[1]/4 + u[2]/2 + u[3]/4;
        int d = u[2]/4 + u[3]/2 + ur[0]/4, e = u[3]/4 + ur[0]/2 + ur[1]/4;
        int f = ur[0]/4 + ur[1]/2 + ur[2]/4, g = ur[1]/4 + ur[2]/2 + ur[3]/4;
        int h = ur[2]/4 + three_q(ur[3], ur_u8);
        p[0][0]=a; p[0][1]=b; p[1][0]=b; p[0][2]=d; p[1][1]=d; p[2][0]=d;
        p[0][3]=e; p[1][2]=e; p[2][1]=e; p[3][0]=e; p[1][3]=f; p[2][2]=f; p[3][1]=f;
        p[2][3]=g; p[3][2]=g; p[3][3]=h;
        break; }
    case 4: {                                                                        /* :45-63 */
        int a = u[1]/4 + u[2]/2 + u[3]/4, b = u[0]/4 + u[1]/2 + u[2]/4;
        int d = ul/4 + u[0]/2 + u[1]/4, e = ul/4 + u[0]/2 + l[0]/4;
        int f = u[0]/4 + l[0]/2 + l[1]/4, g = l[0]/4 + l[1]/2 + l[2]/4, h = l[1]/4 + l[2]/2 + l[3]/4;
        p[0][3]=a; p[0][2]=b; p[1][3]=b; p[0][1]=d; p[1][2]=d; p[2][3]=d;
        p[0][0]=e; p[1][1]=e; p[2][2]=e; p[3][3]=e; p[1][0]=f; p[2][1]=f; p[3][2]=f;
        p[2][0]=g; p[3][1]=g; p[3][0]=h;
        break; }
    case 5: {                                                                        /* :65-83 */
        p[0][0] = ul/2 + u[0]/2;              p[2][1] = p[0][0];
        p[0][1] = u[0]/2 + u[1]/2;            p[2][2] = p[0][1];
        p[0][2] = u[1]/2 + u[2]/2;            p[2][3] = p[0][2];
        p[0][3] = u[2]/2 + u[3]/2;
        p[1][0] = u[0]/4 + ul/2 + l[0]/4;     p[3][1] = p[1][0];
        p[1][1] = ul/4 + u[0]/2 + u[1]/4;     p[3][2] = p[1][1];
        p[1][2] = u[0]/4 + u[1]/2 + u[2]/4;   p[3][3] = p[1][2];
        p[1][3] = u[1]/4 + u[2]/2 + u[3]/4;
        p[2][0] = ul/4 + l[0]/2 + l[1]/4;
        p[3][0] = l[0]/4 + l[1]/2 + l[2]/4;
        break; }
    case 6: {                                                                        /* :85-103 */
        p[0][0] = ul/2 + l[0]/2;              p[1][2] = p[0][0];
        p[0][1] = u[0]/4 + ul/2 + l[0]/4;     p[1][3] = p[0][1];
        p[0][2] = ul/4 + u[0]/2 + u[1]/4;
        p[0][3] = u[0]/4 + u[1]/2 + u[2]/4;
        p[1][0] = l[0]/2 + l[1]/2;            p[2][2] = p[1][0];
        p[1][1] = ul/4 + l[1]/2 + l[2]/4;     p[2][3] = p[1][1];
        p[2][0] = l[1]/2 + l[2]/2;            p[3][2] = p[2][0];
        p[2][1] = l[0]/4 + l[1]/2 + l[2]/4;   p[3][3] = p[2][1];
        p[3][0] = l[2]/2 + l[3]/2;
        p[3][1] = l[1]/4 + l[2]/2 + l[3]/4;
        break; }
    case 7: {                                                                        /* :105-123 */
        p[0][0] = u[0]/2 + u[1]/2;
        p[0][1] = u[1]/2 + u[2]/2;            p[2][0] = p[0][1];
        p[0][2] = u[2]/2 + u[3]/2;            p[2][1] = p[0][2];
        p[0][3] = u[3]/2 + ur[0]/2;           p[2][2] = p[0][3];
        p[2][3] = ur[0]/2 + ur[1]/2;
        p[1][0] = u[0]/4 + u[1]/2 + u[2]/4;
        p[1][1] = u[1]/4 + u[2]/2 + u[3]/4;   p[3][0] = p[1][1];
        p[1][2] = u[2]/4 + u[3]/2 + ur[0]/4;  p[3][1] = p[1][2];
        p[1][3] = u[3]/4 + ur[0]/2 + ur[1]/4; p[3][2] = p[1][3];
        p[3][3] = ur[0]/4 + ur[1]/2 + ur[2]/4;
        break; }
    default: {                                                                       /* :125-143 */
        p[0][0] = l[0]/2 + l[1]/2;
        p[0][1] = l[0]/4 + l[1]/2 + l[2]/4;
        p[0][2] = l[1]/2 + l[2]/2;            p[1][0] = p[0][2];
        p[0][3] = l[1]/4 + l[2]/2 + l[3]/4;   p[1][1] = p[0][3];
        p[1][2] = l[2]/2 + l[3]/2;            p[2][0] = p[1][2];
        p[1][3] = l[2]/4 + three_q(l[3], l_u8); p[2][1] = p[1][3];
        p[3][0] = p[2][2] = p[2][3] = p[3][1] = p[3][2] = p[3][3] = l[3];
        break; }
    }
}

/* intraframe.py:24-151 (luma4x4).  Y: H x W uint8 (multiples of 4, W >= 8).
 * res, pred: H x W float64; modes: (H/4) x (W/4) float64.
 * Neighbour availability (:38-54): the aliased `available` rows (:30) make no
 * visible difference in raster order; what matters is the CASE ORDER: top row
 * -> left only; left column -> up and up-right only; last column -> no
 * up-right (replicated Y[i-1, j+3] as float64, :69-70).                      */
int vcso_intra_luma4x4(const uint8_t *Y, int H, int W, double *res, double *pred, double *modes) {
    if (H % 4 || W % 4 || W < 8) return -1;
    const int rows = H / 4, cols = W / 4;
#pragma omp parallel for schedule(static)
    for (int bi = 0; bi < rows; bi++)
        for (int bj = 0; bj < cols; bj++) {
            const int i = bi * 4, j = bj * 4;
            int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            if (bi == 0 && bj == 0) { }
            else if (bi == 0) s3 = 1;
            else if (bj == 0) { s1 = 1; s2 = 1; }
            else if (bj + 1 == cols) { s0 = 1; s1 = 1; s3 = 1; }
            else { s0 = s1 = s2 = s3 = 1; }
            int ul = s0 ? Y[(i - 1) * W + j - 1] : 128, u[4], ur[4], l[4];
            for (int k = 0; k < 4; k++) {
                u[k] = s1 ? Y[(i - 1) * W + j + k] : 128;
                ur[k] = s2 ? Y[(i - 1) * W + j + 4 + k] : (s1 ? Y[(i - 1) * W + j + 3] : 128);
                l[k] = s3 ? Y[(i + k) * W + j - 1] : 128;
            }
            int best = 16 * 255, bmode = 0, bp[4][4] = {{0}}; /* :79-81 */
            for (int m = 0; m < 9; m++) {
                int p[4][4], d = 0;
                pred4x4(m, ul, u, ur, l, s1, s2, s3, p);
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c < 4; c++) d += abs(p[r][c] - (int)Y[(i + r) * W + j + c]);
                if (d < best) { best = d; bmode = m; memcpy(bp, p, sizeof bp); } /* strict < */
            }
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) {
                    pred[(i + r) * W + j + c] = bp[r][c];
                    res[(i + r) * W + j + c] = (int)Y[(i + r) * W + j + c] - bp[r][c]; /* :146 */
                }
            modes[bi * cols + bj] = bmode;
        }
    return 0;
}

/* intraframe.py:153-225 (luma16x16) with intramodes.py:145-161.             */
int vcso_intra_luma16x16(const uint8_t *Y, int H, int W, double *res, double *pred, double *modes) {
    if (H % 16 || W % 16) return -1;
    const int rows = H / 16, cols = W / 16;
#pragma omp parallel for schedule(static)
    for (int bi = 0; bi < rows; bi++)
        for (int bj = 0; bj < cols; bj++) {
            const int i = bi * 16, j = bj * 16;
            int s1 = 0, s2 = 0; /* u, l (ul is fetched but no predictor uses it) */
            if (bi == 0 && bj == 0) { }
            else if (bi == 0) s2 = 1;
            else if (bj == 0) s1 = 1;
            else { s1 = s2 = 1; }
            int u[16], l[16], su = 0, sl = 0;
            for (int k = 0; k < 16; k++) {
                u[k] = s1 ? Y[(i - 1) * W + j + k] : 128; su += u[k];
                l[k] = s2 ? Y[(i + k) * W + j - 1] : 128; sl += l[k];
            }
            const int dc = (su + sl) / 32; /* :159 np.sum of uint8 does not wrap */
            int d[3] = {0, 0, 0};
            for (int r = 0; r < 16; r++)
                for (int c = 0; c < 16; c++) {
                    int y = Y[(i + r) * W + j + c];
                    d[0] += abs(u[c] - y); d[1] += abs(l[r] - y); d[2] += abs(dc - y);
                }
            int best = 16 * 16 * 255, bmode = -1; /* :196 */
            for (int m = 0; m < 3; m++) if (d[m] < best) { best = d[m]; bmode = m; }
            for (int r = 0; r < 16; r++)
                for (int c = 0; c < 16; c++) {
                    int p = bmode == 0 ? u[c] : bmode == 1 ? l[r] : bmode == 2 ? dc : 0;
                    pred[(i + r) * W + j + c] = p;
                    res[(i + r) * W + j + c] = (int)Y[(i + r) * W + j + c] - p;
                }
            modes[bi * cols + bj] = bmode < 0 ? 0 : bmode;
        }
    return 0;
}

/* intraframe.py:228-317 (chroma8x8).  QUIRK (:266): the upper neighbour of the
 * Cb block is read from Cbres -- the RESIDUAL being built -- not from Cb, which
 * makes every block depend on the one above it (serial down each column).   */
int vcso_intra_chroma8x8(const uint8_t *Cr, const uint8_t *Cb, int H, int W, double *crres,
                         double *crpred, double *cbres, double *cbpred, double *modes) {
    if (H % 8 || W % 8) return -1;
    const int rows = H / 8, cols = W / 8;
#pragma omp parallel for schedule(static)
    for (int bj = 0; bj < cols; bj++)
        for (int bi = 0; bi < rows; bi++) {
            const int i = bi * 8, j = bj * 8;
            int s1 = 0, s2 = 0;
            if (bi == 0 && bj == 0) { }
            else if (bi == 0) s2 = 1;
            else if (bj == 0) s1 = 1;
            else { s1 = s2 = 1; }
            int ur[8], ub[8], lr[8], lb[8], sur = 0, sub = 0, slr = 0, slb = 0;
            for (int k = 0; k < 8; k++) {
                ur[k] = s1 ? Cr[(i - 1) * W + j + k] : 128;
                ub[k] = s1 ? (int)cbres[(i - 1) * W + j + k] : 128; /* :266 */
                lr[k] = s2 ? Cr[(i + k) * W + j - 1] : 128;
                lb[k] = s2 ? Cb[(i + k) * W + j - 1] : 128;
                sur += ur[k]; sub += ub[k]; slr += lr[k]; slb += lb[k];
            }
            const int dcr = fdiv(sur + slr, 16), dcb = fdiv(sub + slb, 16); /* intramodes.py:177 */
            int d[3] = {0, 0, 0};
            for (int r = 0; r < 8; r++)
                for (int c = 0; c < 8; c++) {
                    int yr = Cr[(i + r) * W + j + c], yb = Cb[(i + r) * W + j + c];
                    d[0] += abs(ur[c] - yr) + abs(ub[c] - yb);
                    d[1] += abs(lr[r] - yr) + abs(lb[r] - yb);
                    d[2] += abs(dcr - yr) + abs(dcb - yb);
                }
            int best = 2 * 8 * 8 * 255, bmode = -1; /* :280 */
            for (int m = 0; m < 3; m++) if (d[m] < best) { best = d[m]; bmode = m; }
            for (int r = 0; r < 8; r++)
                for (int c = 0; c < 8; c++) {
                    int pr = bmode == 0 ? ur[c] : bmode == 1 ? lr[r] : bmode == 2 ? dcr : 0;
                    int pb = bmode == 0 ? ub[c] : bmode == 1 ? lb[r] : bmode == 2 ? dcb : 0;
                    size_t o = (size_t)(i + r) * W + j + c;
                    crpred[o] = pr; crres[o] = (int)Cr[o] - pr;
                    cbpred[o] = pb; cbres[o] = (int)Cb[o] - pb;
                }
            modes[bi * cols + bj] = bmode < 0 ? 0 : bmode;
        }
    return 0;
}

/* intraframe.py:352-358 (commented out upstream) with DCTTransform of :14-17:
 * out = np.round(C . X . C^T) per 4x4 block.  np.matmul promotes C to float64. */
int vcso_intra_transform4x4(const double *X, int H, int W, double *out) {
    static const double C[4][4] = {{1, 1, 1, 1}, {2, 1, -1, -2}, {1, -1, -1, 1}, {1, -2, 2, -1}};
    if (H % 4 || W % 4) return -1;
#pragma omp parallel for schedule(static)
    for (int by = 0; by < H / 4; by++)
        for (int bx = 0; bx < W / 4; bx++) {
            const double *x = X + (size_t)(4 * by) * W + 4 * bx;
            double t[4][4];
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) {
                    double a = 0.0;
                    for (int k = 0; k < 4; k++) a += C[i][k] * x[(size_t)k * W + j];
                    t[i][j] = a;
                }
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) {
                    double a = 0.0;
                    for (int k = 0; k < 4; k++) a += t[i][k] * C[j][k];
                    out[(size_t)(4 * by + i) * W + 4 * bx + j] = rint(a);  /* np.round: half to even */
                }
        }
    return 0;
}
