// intra_kernels.hpp -- intra prediction + mode decision (SURVEY 8f row N1,
// BASELINE config 5): /root/reference/IntraframeCompression/intraframe.py
// (luma4x4 :24-151, luma16x16 :153-225, chroma8x8 :228-317) with the
// predictors of intramodes.py.  Every value the reference computes is
// integer-valued (floor divisions held in float64), so the kernels work in
// int32 and convert on store; results are bit-identical (tests/golden/
// intra_small.npz).  Neighbours come from the ORIGINAL plane, so luma blocks
// are independent; chroma Cb reads the residual of the block above (:266) and
// is serial down each block column.  NumPy-2 quirks reproduced: uint8 wrap in
// 3*x//4 (intramodes.py:42,135) and in u+l of dc4x4 (:21) when the operands
// are real uint8 rows rather than the float64 128-stand-ins.
#pragma once
#include "vcs_common.hpp"

namespace vcs {

__device__ __forceinline__ int b0(uint32_t v) { return (int)(v & 255u); }
__device__ __forceinline__ int b1(uint32_t v) { return (int)((v >> 8) & 255u); }
__device__ __forceinline__ int b2(uint32_t v) { return (int)((v >> 16) & 255u); }
__device__ __forceinline__ int b3(uint32_t v) { return (int)(v >> 24); }
__device__ __forceinline__ int three_q(int x, bool u8) { return u8 ? ((3 * x) & 255) / 4 : (3 * x) / 4; }

// cost of a 4x4 prediction p against the block y, both row-major int[16]
__device__ __forceinline__ int cost16(const int (&p)[16], const int (&y)[16]) {
    int d = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) d += abs(p[k] - y[k]);
    return d;
}

// --------------------------------------------------------------------------
// k_intra_luma4: one thread per 4x4 block (adjacent lanes = horizontally
// adjacent blocks, so a wave's row stores are contiguous).
// --------------------------------------------------------------------------
// (all three prediction kernels: blockIdx.y = frame of a batch; in_stride / out_stride / modes_stride are
// the distances between consecutive frames' planes in elements -- 0 for the single-frame entry points)
__global__ __launch_bounds__(256) void k_intra_luma4(const uint8_t *__restrict__ Y, int H, int W,
                                                     double *__restrict__ res,
                                                     double *__restrict__ pred,
                                                     double *__restrict__ modes, size_t in_stride = 0,
                                                     size_t out_stride = 0, size_t modes_stride = 0) {
    Y += blockIdx.y * in_stride; res += blockIdx.y * out_stride; pred += blockIdx.y * out_stride;
    modes += blockIdx.y * modes_stride;
    const int cols = W >> 2, rows = H >> 2;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * cols) return;
    const int bi = id / cols, bj = id - bi * cols;
    const int i = bi * 4, j = bj * 4;
    // availability, intraframe.py:38-54 (case order matters)
    bool s0 = false, s1 = false, s2 = false, s3 = false;
    if (bi == 0 && bj == 0) {
    } else if (bi == 0) {
        s3 = true;
    } else if (bj == 0) {
        s1 = s2 = true;
    } else if (bj + 1 == cols) {
        s0 = s1 = s3 = true;
    } else {
        s0 = s1 = s2 = s3 = true;
    }
    int y[16];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t v = *(const uint32_t *)(Y + (size_t)(i + r) * W + j);
        y[4 * r + 0] = b0(v); y[4 * r + 1] = b1(v); y[4 * r + 2] = b2(v); y[4 * r + 3] = b3(v);
    }
    int ul = 128, u[4] = {128, 128, 128, 128}, ur[4] = {128, 128, 128, 128}, l[4] = {128, 128, 128, 128};
    if (s1) {
        const uint32_t v = *(const uint32_t *)(Y + (size_t)(i - 1) * W + j);
        u[0] = b0(v); u[1] = b1(v); u[2] = b2(v); u[3] = b3(v);
        ur[0] = ur[1] = ur[2] = ur[3] = u[3];  // :69-70 replicated Y[i-1, j+3]
    }
    if (s2) {
        const uint32_t v = *(const uint32_t *)(Y + (size_t)(i - 1) * W + j + 4);
        ur[0] = b0(v); ur[1] = b1(v); ur[2] = b2(v); ur[3] = b3(v);
    }
    if (s0) ul = Y[(size_t)(i - 1) * W + j - 1];
    if (s3) {
#pragma unroll
        for (int r = 0; r < 4; r++) l[r] = Y[(size_t)(i + r) * W + j - 1];
    }

    int best = 16 * 255, bmode = 0;  // :79-81
    int bp[16];
#pragma unroll
    for (int k = 0; k < 16; k++) bp[k] = 0;
#define TRY_MODE(M, ...)                                \
    {                                                   \
        const int p[16] = {__VA_ARGS__};                \
        const int d = cost16(p, y);                     \
        if (d < best) {                                 \
            best = d;                                   \
            bmode = M;                                  \
            _Pragma("unroll") for (int k = 0; k < 16; k++) bp[k] = p[k]; \
        }                                               \
    }
    // 0 vertical (intramodes.py:7-11), 1 horizontal (:13-17)
    TRY_MODE(0, u[0], u[1], u[2], u[3], u[0], u[1], u[2], u[3], u[0], u[1], u[2], u[3], u[0], u[1], u[2], u[3])
    TRY_MODE(1, l[0], l[0], l[0], l[0], l[1], l[1], l[1], l[1], l[2], l[2], l[2], l[2], l[3], l[3], l[3], l[3])
    {   // 2 dc (:19-23): u + l wraps per element when both are uint8 rows
        int s = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) s += (s1 && s3) ? ((u[c] + l[c]) & 255) : (u[c] + l[c]);
        const int a = s / 8;
        TRY_MODE(2, a, a, a, a, a, a, a, a, a, a, a, a, a, a, a, a)
    }
    {   // 3 down-left (:25-43)
        const int a = u[0] / 4 + u[1] / 2 + u[2] / 4, b = u[1] / 4 + u[2] / 2 + u[3] / 4;
        const int d = u[2] / 4 + u[3] / 2 + ur[0] / 4, e = u[3] / 4 + ur[0] / 2 + ur[1] / 4;
        const int f = ur[0] / 4 + ur[1] / 2 + ur[2] / 4, g = ur[1] / 4 + ur[2] / 2 + ur[3] / 4;
        const int h = ur[2] / 4 + three_q(ur[3], s2);
        TRY_MODE(3, a, b, d, e, b, d, e, f, d, e, f, g, e, f, g, h)
    }
    {   // 4 down-right (:45-63)
        const int a = u[1] / 4 + u[2] / 2 + u[3] / 4, b = u[0] / 4 + u[1] / 2 + u[2] / 4;
        const int d = ul / 4 + u[0] / 2 + u[1] / 4, e = ul / 4 + u[0] / 2 + l[0] / 4;
        const int f = u[0] / 4 + l[0] / 2 + l[1] / 4, g = l[0] / 4 + l[1] / 2 + l[2] / 4;
        const int h = l[1] / 4 + l[2] / 2 + l[3] / 4;
        TRY_MODE(4, e, d, b, a, f, e, d, b, g, f, e, d, h, g, f, e)
    }
    {   // 5 vertical-right (:65-83)
        const int p00 = ul / 2 + u[0] / 2, p01 = u[0] / 2 + u[1] / 2, p02 = u[1] / 2 + u[2] / 2;
        const int p03 = u[2] / 2 + u[3] / 2, p10 = u[0] / 4 + ul / 2 + l[0] / 4;
        const int p11 = ul / 4 + u[0] / 2 + u[1] / 4, p12 = u[0] / 4 + u[1] / 2 + u[2] / 4;
        const int p13 = u[1] / 4 + u[2] / 2 + u[3] / 4, p20 = ul / 4 + l[0] / 2 + l[1] / 4;
        const int p30 = l[0] / 4 + l[1] / 2 + l[2] / 4;
        TRY_MODE(5, p00, p01, p02, p03, p10, p11, p12, p13, p20, p00, p01, p02, p30, p10, p11, p12)
    }
    {   // 6 horizontal-down (:85-103)
        const int p00 = ul / 2 + l[0] / 2, p01 = u[0] / 4 + ul / 2 + l[0] / 4;
        const int p02 = ul / 4 + u[0] / 2 + u[1] / 4, p03 = u[0] / 4 + u[1] / 2 + u[2] / 4;
        const int p10 = l[0] / 2 + l[1] / 2, p11 = ul / 4 + l[1] / 2 + l[2] / 4;
        const int p20 = l[1] / 2 + l[2] / 2, p21 = l[0] / 4 + l[1] / 2 + l[2] / 4;
        const int p30 = l[2] / 2 + l[3] / 2, p31 = l[1] / 4 + l[2] / 2 + l[3] / 4;
        TRY_MODE(6, p00, p01, p02, p03, p10, p11, p00, p01, p20, p21, p10, p11, p30, p31, p20, p21)
    }
    {   // 7 vertical-left (:105-123)
        const int p00 = u[0] / 2 + u[1] / 2, p01 = u[1] / 2 + u[2] / 2, p02 = u[2] / 2 + u[3] / 2;
        const int p03 = u[3] / 2 + ur[0] / 2, p23 = ur[0] / 2 + ur[1] / 2;
        const int p10 = u[0] / 4 + u[1] / 2 + u[2] / 4, p11 = u[1] / 4 + u[2] / 2 + u[3] / 4;
        const int p12 = u[2] / 4 + u[3] / 2 + ur[0] / 4, p13 = u[3] / 4 + ur[0] / 2 + ur[1] / 4;
        const int p33 = ur[0] / 4 + ur[1] / 2 + ur[2] / 4;
        TRY_MODE(7, p00, p01, p02, p03, p10, p11, p12, p13, p01, p02, p03, p23, p11, p12, p13, p33)
    }
    {   // 8 horizontal-up (:125-143)
        const int p00 = l[0] / 2 + l[1] / 2, p01 = l[0] / 4 + l[1] / 2 + l[2] / 4;
        const int p02 = l[1] / 2 + l[2] / 2, p03 = l[1] / 4 + l[2] / 2 + l[3] / 4;
        const int p12 = l[2] / 2 + l[3] / 2, p13 = l[2] / 4 + three_q(l[3], s3), q = l[3];
        TRY_MODE(8, p00, p01, p02, p03, p02, p03, p12, p13, p12, p13, q, q, q, q, q, q)
    }
#undef TRY_MODE
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double *pr = pred + (size_t)(i + r) * W + j, *rr = res + (size_t)(i + r) * W + j;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            pr[c] = (double)bp[4 * r + c];
            rr[c] = (double)(y[4 * r + c] - bp[4 * r + c]);  // :146
        }
    }
    modes[id] = (double)bmode;
}

// --------------------------------------------------------------------------
// k_intra_luma16: one wave per 16x16 block; lane = (row, 4-pixel group).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_intra_luma16(const uint8_t *__restrict__ Y, int H, int W,
                                                     double *__restrict__ res,
                                                     double *__restrict__ pred,
                                                     double *__restrict__ modes, size_t in_stride = 0,
                                                     size_t out_stride = 0, size_t modes_stride = 0) {
    Y += blockIdx.y * in_stride; res += blockIdx.y * out_stride; pred += blockIdx.y * out_stride;
    modes += blockIdx.y * modes_stride;
    const int cols = W >> 4;
    const int bi = blockIdx.x / cols, bj = blockIdx.x - bi * cols;
    const int i = bi * 16, j = bj * 16;
    const int lane = threadIdx.x, r = lane >> 2, c0 = (lane & 3) * 4;
    const bool s1 = bi > 0, s2 = bj > 0;  // :167-177: top row has no `u`, left column no `l`
    const uint32_t yv = *(const uint32_t *)(Y + (size_t)(i + r) * W + j + c0);
    uint32_t uv = 0x80808080u;
    int lv = 128;
    if (s1) uv = *(const uint32_t *)(Y + (size_t)(i - 1) * W + j + c0);
    if (s2) lv = Y[(size_t)(i + r) * W + j - 1];
    const int y4[4] = {b0(yv), b1(yv), b2(yv), b3(yv)}, u4[4] = {b0(uv), b1(uv), b2(uv), b3(uv)};
    // sum(u) + sum(l): lanes of row 0 contribute their 4 u's, lanes of group 0 their l
    uint32_t part = (r == 0 ? (uint32_t)(u4[0] + u4[1] + u4[2] + u4[3]) : 0u) +
                    ((lane & 3) == 0 ? (uint32_t)lv : 0u);
    const int dc = (int)(wave_sum_u32(part) / 32u);  // intramodes.py:159
    uint32_t d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        d0 += abs(u4[k] - y4[k]);
        d1 += abs(lv - y4[k]);
        d2 += abs(dc - y4[k]);
    }
    d0 = wave_sum_u32(d0);
    d1 = wave_sum_u32(d1);
    d2 = wave_sum_u32(d2);
    int best = 16 * 16 * 255, bmode = -1;  // :196
    if ((int)d0 < best) { best = (int)d0; bmode = 0; }
    if ((int)d1 < best) { best = (int)d1; bmode = 1; }
    if ((int)d2 < best) { best = (int)d2; bmode = 2; }
    double *pr = pred + (size_t)(i + r) * W + j + c0, *rr = res + (size_t)(i + r) * W + j + c0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = bmode == 0 ? u4[k] : bmode == 1 ? lv : bmode == 2 ? dc : 0;
        pr[k] = (double)p;
        rr[k] = (double)(y4[k] - p);
    }
    if (lane == 0) modes[blockIdx.x] = (double)(bmode < 0 ? 0 : bmode);
}

// --------------------------------------------------------------------------
// k_intra_chroma8: one wave per block COLUMN, walking down it (the Cb
// prediction reads the residual row above, intraframe.py:266); lane = pixel.
// --------------------------------------------------------------------------
__device__ __forceinline__ int floor_div16(int a) { return a >> 4; }  // Python // 16 (arithmetic shift)

__global__ __launch_bounds__(64) void k_intra_chroma8(const uint8_t *__restrict__ Cr,
                                                      const uint8_t *__restrict__ Cb, int H, int W,
                                                      double *__restrict__ crres,
                                                      double *__restrict__ crpred,
                                                      double *__restrict__ cbres,
                                                      double *__restrict__ cbpred,
                                                      double *__restrict__ modes, size_t in_stride = 0,
                                                      size_t out_stride = 0, size_t modes_stride = 0) {
    Cr += blockIdx.y * in_stride; Cb += blockIdx.y * in_stride;
    crres += blockIdx.y * out_stride; crpred += blockIdx.y * out_stride;
    cbres += blockIdx.y * out_stride; cbpred += blockIdx.y * out_stride;
    modes += blockIdx.y * modes_stride;
    __shared__ int s_ub[8];
    const int cols = W >> 3, rows = H >> 3;
    const int bj = blockIdx.x, j = bj * 8;
    const int lane = threadIdx.x, r = lane >> 3, c = lane & 7;
    const bool s2 = bj > 0;
    // Only `ub` depends on the block above; the pixels of the next block are fetched while this
    // one is decided (the walk is a chain of 134 dependent steps at 1080p: their load latency
    // was most of the kernel).  Index clamps keep the look-ahead of the last block in bounds.
    struct Px { int yr, yb, ur, lr, lb; };
    auto fetch = [&](int bi) {
        const int i = min(bi, rows - 1) * 8;
        Px p;
        p.yr = Cr[(size_t)(i + r) * W + j + c];
        p.yb = Cb[(size_t)(i + r) * W + j + c];
        p.ur = Cr[(size_t)max(i - 1, 0) * W + j + c];
        p.lr = Cr[(size_t)(i + r) * W + max(j - 1, 0)];
        p.lb = Cb[(size_t)(i + r) * W + max(j - 1, 0)];
        return p;
    };
    Px nx = fetch(0);
    for (int bi = 0; bi < rows; bi++) {
        const int i = bi * 8;
        const bool s1 = bi > 0;
        const size_t o = (size_t)(i + r) * W + j + c;
        const Px px = nx;
        nx = fetch(bi + 1);
        const int yr = px.yr, yb = px.yb;
        int ur = 128, ub = 128, lr = 128, lb = 128;
        if (s1) {
            ur = px.ur;
            ub = s_ub[c];  // Cbres[i-1, j+c] of the block above (:266)
        }
        if (s2) {
            lr = px.lr;
            lb = px.lb;
        }
        // sums over the 8 up / 8 left neighbours (ub may be negative: bias it)
        const uint32_t top = r == 0 ? 1u : 0u, lef = c == 0 ? 1u : 0u;
        const int sur = (int)wave_sum_u32(top * (uint32_t)ur), slr = (int)wave_sum_u32(lef * (uint32_t)lr);
        const int slb = (int)wave_sum_u32(lef * (uint32_t)lb);
        const int sub = (int)wave_sum_u32(top * (uint32_t)(ub + (1 << 20))) - (8 << 20);
        const int dcr = floor_div16(sur + slr), dcb = floor_div16(sub + slb);  // intramodes.py:177
        uint32_t d0 = abs(ur - yr) + abs(ub - yb), d1 = abs(lr - yr) + abs(lb - yb);
        uint32_t d2 = abs(dcr - yr) + abs(dcb - yb);
        d0 = wave_sum_u32(d0);
        d1 = wave_sum_u32(d1);
        d2 = wave_sum_u32(d2);
        long long best = 2 * 8 * 8 * 255;  // :280
        int bmode = -1;
        if ((long long)d0 < best) { best = d0; bmode = 0; }
        if ((long long)d1 < best) { best = d1; bmode = 1; }
        if ((long long)d2 < best) { best = d2; bmode = 2; }
        const int pr = bmode == 0 ? ur : bmode == 1 ? lr : bmode == 2 ? dcr : 0;
        const int pb = bmode == 0 ? ub : bmode == 1 ? lb : bmode == 2 ? dcb : 0;
        crpred[o] = (double)pr;
        crres[o] = (double)(yr - pr);
        cbpred[o] = (double)pb;
        cbres[o] = (double)(yb - pb);
        wave_lds_fence();  // everyone has read s_ub
        if (r == 7) s_ub[c] = yb - pb;
        wave_lds_fence();
        if (lane == 0) modes[(size_t)bi * cols + bj] = (double)(bmode < 0 ? 0 : bmode);
    }
}

// --------------------------------------------------------------------------
// k_intra_transform4: the 4x4 integer core transform the reference keeps
// commented out behind its intra prediction (intraframe.py:14-17, :352-358):
//   out = np.round(C . X . C^T) per 4x4 block of the residual plane,
//   C = [[1,1,1,1],[2,1,-1,-2],[1,-1,-1,1],[1,-2,2,-1]].
// No running upstream code pins it; for the integer-valued residuals the
// intra predictors produce every order of evaluation is exact.  Thread = one
// 4x4 block, rows as 32-byte loads; products left to right, np.round = rint.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_intra_transform4(const double *__restrict__ X, int H, int W,
                                                          double *__restrict__ out, size_t in_stride = 0,
                                                          size_t out_stride = 0) {
    X += blockIdx.y * in_stride; out += blockIdx.y * out_stride;
    const int nbx = W >> 2, nb = nbx * (H >> 2);
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int by = b / nbx, bx = b - by * nbx;
    const size_t o = (size_t)(4 * by) * W + (size_t)(4 * bx);
    typedef double d2 __attribute__((ext_vector_type(2)));
    double x[4][4], t[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const d2 a = *(const d2 *)(X + o + (size_t)r * W), c = *(const d2 *)(X + o + (size_t)r * W + 2);
        x[r][0] = a.x; x[r][1] = a.y; x[r][2] = c.x; x[r][3] = c.y;
    }
    constexpr double C[4][4] = {{1, 1, 1, 1}, {2, 1, -1, -2}, {1, -1, -1, 1}, {1, -2, 2, -1}};
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            t[i][j] = ((C[i][0] * x[0][j] + C[i][1] * x[1][j]) + C[i][2] * x[2][j]) + C[i][3] * x[3][j];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double r[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            r[j] = __builtin_rint(((t[i][0] * C[j][0] + t[i][1] * C[j][1]) + t[i][2] * C[j][2]) +
                                  t[i][3] * C[j][3]);
        d2 a, c;
        a.x = r[0]; a.y = r[1]; c.x = r[2]; c.y = r[3];
        *(d2 *)(out + o + (size_t)i * W) = a;
        *(d2 *)(out + o + (size_t)i * W + 2) = c;
    }
}

}  // namespace vcs
