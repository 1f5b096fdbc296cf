// What can a kernel with k_fdct's traffic shape reach on this chip?  It reads 3 bytes per pixel
// (unaligned dword per lane, 8 rows of a 64-pixel strip) and writes three float64 planes with the
// same row-contiguous non-temporal stores, and computes nothing.   29 x 1080p per launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ub_stream tools/ub_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(64) void k_shape(const uint8_t *img, double *planes, int H, int W, int n) {
    const int groups_x = W / 64, per_frame = groups_x * (H / 8);
    const long long id = blockIdx.x;
    const int f = (int)(id / per_frame), s = (int)(id - (long long)f * per_frame);
    const int ty = s / groups_x, gx = s - ty * groups_x, lane = threadIdx.x;
    const size_t F = (size_t)H * W * 3, plane = (size_t)H * W;
    const uint8_t *src = img + f * F;
    uint32_t px[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t v;
        __builtin_memcpy(&v, src + (size_t)(ty * 8 + k) * W * 3 + 3 * (gx * 64 + lane), 4);
        px[k] = v;
    }
    double *out = planes + (size_t)f * 3 * plane;
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const int e = s4 * 64 + lane, row = e >> 5, c2 = e & 31;
            d2 v;
            v.x = (double)((px[row] >> (8 * ch)) & 255u);
            v.y = (double)((px[(row + 1) & 7] >> (8 * ch)) & 255u);
            d2 *o = (d2 *)(out + ch * plane + (size_t)(ty * 8 + row) * W + gx * 64 + 2 * c2);
            if (NT) __builtin_nontemporal_store(v, o); else *o = v;
        }
}
// k_idct's traffic shape: a wave reads the 8 rows x 64 columns of three float64 planes (one 8-byte load per lane,
// row and plane; non-temporal or plain) and writes the strip's 8 x 192 bytes of pixels (24 bytes per lane), and
// computes nothing.  WAVES strips per workgroup.
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
template <bool NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_shape_dec(const double *planes, uint8_t *img, int H, int W, int n) {
    const int groups_x = W / 64, per_frame = groups_x * (H / 8);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = min((int)blockIdx.x * WAVES + wave, per_frame - 1), f = blockIdx.y;
    const int ty = s / groups_x, gx = s - ty * groups_x;
    const size_t F = (size_t)H * W * 3, plane = (size_t)H * W;
    const double *in = planes + (size_t)f * 3 * plane;
    double X[3][8];
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double *p = in + ch * plane + (size_t)(ty * 8 + k) * W + gx * 64 + lane;
            X[ch][k] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    const int t = lane >> 3, j = lane & 7;
    uint8_t *o = img + f * F + (size_t)(ty * 8 + j) * W * 3 + 3 * (gx * 64 + t * 8);
#pragma unroll
    for (int q = 0; q < 3; q++) {
        u2 v;
        v.x = (uint32_t)(long long)(X[q][0] + X[q][1] + X[q][2] + X[q][3]);
        v.y = (uint32_t)(long long)(X[q][4] + X[q][5] + X[q][6] + X[q][7]);
        __builtin_nontemporal_store(v, (u2 *)(o + 8 * q));
    }
}
template <bool NT, int WAVES>
static float run_dec(const double *pl[3], uint8_t *img, int H, int W, int n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    const unsigned per_frame = (unsigned)((W / 64) * (H / 8));
    for (int it = 0; it < 12; it++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_shape_dec<NT, WAVES>), dim3((per_frame + WAVES - 1) / WAVES, n), dim3(64 * WAVES), 0, 0, pl[it % 3], img, H, W, n);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it > 2 && ms < best) best = ms;
    }
    return best;
}

// write-pattern variants of the k_fdct shape (no reads at all): which row-run geometry does HBM like best?
//  RUN = bytes one store instruction writes contiguously (512: two rows per instruction as k_fdct does; 1024: one row of
//  128 columns), WAVES per workgroup side by side in x.
template <int RUN, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_wshape(double *planes, int H, int W, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t plane = (size_t)H * W;
    double *out = planes + (size_t)blockIdx.y * 3 * plane;
    d2 v; v.x = 1.0 + lane; v.y = 2.0;
    if (RUN == 512) {   // strip = 64 columns x 8 rows
        const int groups_x = W / 64, per_frame = groups_x * (H / 8);
        const int s = min((int)blockIdx.x * WAVES + wave, per_frame - 1), ty = s / groups_x, gx = s - ty * groups_x;
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const int e = s4 * 64 + lane, row = e >> 5, c2 = e & 31;
                __builtin_nontemporal_store(v, (d2 *)(out + ch * plane + (size_t)(ty * 8 + row) * W + gx * 64 + 2 * c2));
            }
    } else {            // strip = 128 columns x 4 rows: one row per instruction
        const int groups_x = W / 128, per_frame = groups_x * (H / 4);
        const int s = min((int)blockIdx.x * WAVES + wave, per_frame - 1), ty = s / groups_x, gx = s - ty * groups_x;
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
#pragma unroll
            for (int row = 0; row < 4; row++)
                __builtin_nontemporal_store(v, (d2 *)(out + ch * plane + (size_t)(ty * 4 + row) * W + gx * 128 + 2 * lane));
    }
}
__global__ __launch_bounds__(256) void k_wlinear(d2 *p, size_t n2) {
    d2 v; v.x = 1.0; v.y = 2.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, p + i);
}
template <int RUN, int WAVES>
static float run_w(double *pl[3], int H, int W, int n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    const unsigned per_frame = RUN == 512 ? (unsigned)((W / 64) * (H / 8)) : (unsigned)((W / 128) * (H / 4));
    for (int it = 0; it < 12; it++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_wshape<RUN, WAVES>), dim3((per_frame + WAVES - 1) / WAVES, n), dim3(64 * WAVES), 0, 0, pl[it % 3], H, W, n);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it > 2 && ms < best) best = ms;
    }
    return best;
}

int main() {
    {
        const int H = 1080, W = 1920, n = 29;
        const size_t F = (size_t)H * W * 3;
        double *p0, *p1, *p2;
        hipMalloc(&p0, F * n * 8); hipMalloc(&p1, F * n * 8); hipMalloc(&p2, F * n * 8);
        double *pl[3] = {p0, p1, p2};
        const double bytes = (double)F * n * 8;
        float ms;
#define WRUN(R, Wv) ms = run_w<R, Wv>(pl, H, W, n); printf("write shape, %4d-byte runs, %2d waves per workgroup: %.3f ms  %.0f GB/s\n", R, Wv, ms, bytes / ms / 1e6);
        WRUN(512, 1) WRUN(512, 4) WRUN(512, 8) WRUN(512, 15) WRUN(1024, 1) WRUN(1024, 5) WRUN(1024, 15)
        {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            float best = 1e9f;
            for (int it = 0; it < 10; it++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(k_wlinear, dim3(8192), dim3(256), 0, 0, (d2 *)pl[it % 3], (size_t)(F * n / 2));
                hipEventRecord(b); hipEventSynchronize(b);
                float t; hipEventElapsedTime(&t, a, b);
                if (it > 2 && t < best) best = t;
            }
            printf("write linear (grid-stride, 16 bytes per lane):            %.3f ms  %.0f GB/s\n", best, bytes / best / 1e6);
        }
        hipFree(p0); hipFree(p1); hipFree(p2);
    }
    {
        const int H = 1080, W = 1920, n = 29;
        const size_t F = (size_t)H * W * 3;
        double *p0, *p1, *p2; uint8_t *img;
        hipMalloc(&p0, F * n * 8); hipMalloc(&p1, F * n * 8); hipMalloc(&p2, F * n * 8); hipMalloc(&img, F * n + 64);
        hipMemset(p0, 0, F * n * 8); hipMemset(p1, 0, F * n * 8); hipMemset(p2, 0, F * n * 8);
        const double *pl[3] = {p0, p1, p2};
        const double bytes = (double)F * n * 9;
        float ms;
        ms = run_dec<true, 1>(pl, img, H, W, n); printf("decode shape, non-temporal loads, 1 strip per workgroup:  %.3f ms  %.0f GB/s\n", ms, bytes / ms / 1e6);
        ms = run_dec<true, 4>(pl, img, H, W, n); printf("decode shape, non-temporal loads, 4 strips per workgroup: %.3f ms  %.0f GB/s\n", ms, bytes / ms / 1e6);
        ms = run_dec<true, 8>(pl, img, H, W, n); printf("decode shape, non-temporal loads, 8 strips per workgroup: %.3f ms  %.0f GB/s\n", ms, bytes / ms / 1e6);
        ms = run_dec<false, 4>(pl, img, H, W, n); printf("decode shape, plain loads, 4 strips per workgroup:        %.3f ms  %.0f GB/s\n", ms, bytes / ms / 1e6);
        hipFree(p0); hipFree(p1); hipFree(p2); hipFree(img);
    }

    const int H = 1080, W = 1920, n = 29;
    const size_t F = (size_t)H * W * 3;
    uint8_t *img, *img2, *img3; double *planes, *planes2;
    hipMalloc(&img, F * n + 64); hipMalloc(&img2, F * n + 64); hipMalloc(&img3, F * n + 64);
    hipMemset(img2, 9, F * n + 64); hipMemset(img3, 11, F * n + 64); hipMalloc(&planes, F * n * 8); hipMalloc(&planes2, F * n * 8);
    double *planes3; hipMalloc(&planes3, F * n * 8);
    hipMemset(img, 7, F * n + 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned grid = (unsigned)((W / 64) * (H / 8) * n);
    // alt = 1: consecutive launches write two different 1.44 GB regions in turn (what happens
    // when alternate GOPs go to alternate slots) instead of rewriting the same one;
    // alt = 2: they read three different 180 MB inputs in turn instead (540 MB: more than the
    // 256 MB Infinity Cache holds), writing the same region
    // alt = 3: both (three inputs, three output regions)
    for (int alt = 0; alt < 4; alt++)
    for (int nt = 0; nt < 2; nt++) {
        float best = 1e9f;
        for (int it = 0; it < 12; it++) {
            double *dst = (alt == 1 && (it & 1)) ? planes2 : planes;
            if (alt == 3) dst = it % 3 == 0 ? planes : it % 3 == 1 ? planes2 : planes3;
            const uint8_t *src = alt >= 2 ? (it % 3 == 0 ? img : it % 3 == 1 ? img2 : img3) : img;
            hipEventRecord(a);
            if (nt) hipLaunchKernelGGL(k_shape<true>, dim3(grid), dim3(64), 0, 0, src, dst, H, W, n);
            else hipLaunchKernelGGL(k_shape<false>, dim3(grid), dim3(64), 0, 0, src, dst, H, W, n);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it > 1 && ms < best) best = ms;
        }
        const double bytes = (double)F * n * 9;  // 1 read + 8 written per pixel byte
        printf("shape kernel, %s stores, %s: %.3f ms  %.0f GB/s (reads + writes)\n",
               nt ? "non-temporal" : "plain", alt == 1 ? "two output regions in turn" : alt == 2 ? "three inputs in turn" : alt == 3 ? "three inputs and three output regions in turn" : "same input and region", best, bytes / best / 1e6);
    }
    return 0;
}
