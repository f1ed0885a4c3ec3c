// farneback_kernels.h -- dense optical flow after Farneback (polynomial expansion), the form OpenCV ships as
// cv::calcOpticalFlowFarneback with the dense-flow tutorial's parameters (pyr_scale 0.5, levels 3, winsize 15, iterations 3,
// poly_n 5, poly_sigma 1.2), as an ALTERNATIVE to the Lucas-Kanade path of flow_kernels.h.
//
// The reference itself only ever calls lucas_kanade (/root/reference/generate_illusion.py:549-550,
// /root/reference/fitness_calculator.py:498); this option exists because the north star names "Farneback/Lucas-Kanade flow"
// (SURVEY.md 8(f) row 4).  The dense field is sampled on the grid of OpenCV's samples/python/opt_flow.py (draw_flow: every
// `step` pixels from step/2) into the [x, y, dx, dy] vectors the scorers consume -- build-defined, documented in DESIGN.md.
//
// All kernels are HBM/L2-bound stencils over planar float images, one thread per pixel, images of a batch along blockIdx.y.
// Arithmetic = oracle/farneback.c operation by operation (fp32, one rounding per written operation, -ffp-contract=off; the 2x2
// solve in double), so the flow field is bit-identical to the CPU oracle's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eig {

constexpr int FB_MAX_POLY_N = 7;
constexpr int FB_MAX_BLUR_R = 20;   // level 4: sigma 7.5, 39 taps
constexpr int FB_MAX_WIN_R = 16;

struct FbConst {
    float g[FB_MAX_POLY_N + 1], xg[FB_MAX_POLY_N + 1], xxg[FB_MAX_POLY_N + 1];
    float ig[4];                  // inv(G) entries 11, 03, 33, 55
    int poly_n;
};
struct FbBlur {
    float k[FB_MAX_BLUR_R + 1];   // symmetric half of the Gaussian, k[0] = centre
    int r;
};

__device__ __forceinline__ int fb_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int fb_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// convertTo(CV_32F) -> GaussianBlur (REFLECT_101, rows then columns, symmetric form) -> resize(INTER_LINEAR) by 2^k (the centre
// 2x2 of every 2^k block, weights 0.5).  img: uint8 [B][H][W]; out: float [B][H>>k][W>>k].
__global__ void fb_blur_down_kernel(const uint8_t* __restrict__ img, int H, int W, int k, FbBlur bl, float* __restrict__ out)
{
    const int Hk = H >> k, Wk = W >> k;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Hk * Wk) return;
    const int oy = p / Wk, ox = p - oy * Wk;
    const uint8_t* im = img + (size_t)blockIdx.y * H * W;
    auto rowf = [&](int y, int x) {  // T[y][x]
        float a = bl.k[0] * (float)im[(size_t)y * W + x];
        for (int j = 1; j <= bl.r; ++j) {
            const float s = (float)im[(size_t)y * W + fb_reflect101(x - j, W)] + (float)im[(size_t)y * W + fb_reflect101(x + j, W)];
            a = a + bl.k[j] * s;
        }
        return a;
    };
    auto blur = [&](int y, int x) {
        float a = bl.k[0] * rowf(y, x);
        for (int j = 1; j <= bl.r; ++j) {
            const float s = rowf(fb_reflect101(y - j, H), x) + rowf(fb_reflect101(y + j, H), x);
            a = a + bl.k[j] * s;
        }
        return a;
    };
    float v;
    if (k == 0) v = blur(oy, ox);
    else {
        const int f = 1 << k, o = f / 2 - 1;
        const int y0 = oy * f + o, x0 = ox * f + o;
        const float h0 = blur(y0, x0) * 0.5f + blur(y0, x0 + 1) * 0.5f;
        const float h1 = blur(y0 + 1, x0) * 0.5f + blur(y0 + 1, x0 + 1) * 0.5f;
        v = h0 * 0.5f + h1 * 0.5f;
    }
    out[(size_t)blockIdx.y * Hk * Wk + p] = v;
}

// FarnebackPolyExp.  I: float [B][H][W] -> R: float [B][5][H][W] (OpenCV's channel order: x-gradient, y-gradient, xx, yy, xy).
// A 32x8-pixel block stages the vertical pass of its 42 columns in LDS ([8][42][3] floats), the horizontal pass reads it back.
constexpr int FB_PX = 32, FB_PY = 8;
__global__ void __launch_bounds__(FB_PX * FB_PY) fb_polyexp_kernel(const float* __restrict__ I, int H, int W, FbConst c, float* __restrict__ R)
{
    constexpr int CW = FB_PX + 2 * FB_MAX_POLY_N;
    __shared__ float row[FB_PY][CW][3];
    const int n = c.poly_n;
    const int x0 = blockIdx.x * FB_PX, y0 = blockIdx.y * FB_PY;
    const float* im = I + (size_t)blockIdx.z * H * W;
    const int tid = threadIdx.y * FB_PX + threadIdx.x;
    // vertical part for columns x0-n .. x0+FB_PX-1+n (replicated outside the image), rows y0 .. y0+FB_PY-1
    for (int i = tid; i < FB_PY * (FB_PX + 2 * n); i += FB_PX * FB_PY) {
        const int ly = i / (FB_PX + 2 * n), lx = i - ly * (FB_PX + 2 * n);
        const int y = y0 + ly, x = fb_clamp(x0 + lx - n, 0, W - 1);
        float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
        if (y < H) {
            r0 = im[(size_t)y * W + x] * c.g[0];
            for (int k = 1; k <= n; ++k) {
                const float s0 = im[(size_t)fb_clamp(y - k, 0, H - 1) * W + x], s1 = im[(size_t)fb_clamp(y + k, 0, H - 1) * W + x];
                const float p = s0 + s1, d = s1 - s0;
                r0 = r0 + c.g[k] * p;
                r1 = r1 + c.xg[k] * d;
                r2 = r2 + c.xxg[k] * p;
            }
        }
        row[ly][lx][0] = r0; row[ly][lx][1] = r1; row[ly][lx][2] = r2;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int lx = threadIdx.x + n, ly = threadIdx.y;
    float b1 = row[ly][lx][0] * c.g[0], b2 = 0.0f, b3 = row[ly][lx][1] * c.g[0], b4 = 0.0f, b5 = row[ly][lx][2] * c.g[0], b6 = 0.0f;
    for (int k = 1; k <= n; ++k) {
        const float* rp = row[ly][lx + k];
        const float* rm = row[ly][lx - k];
        const float tg = rp[0] + rm[0];
        b1 = b1 + tg * c.g[k];
        b4 = b4 + tg * c.xxg[k];
        b2 = b2 + (rp[0] - rm[0]) * c.xg[k];
        b3 = b3 + (rp[1] + rm[1]) * c.g[k];
        b6 = b6 + (rp[1] - rm[1]) * c.xg[k];
        b5 = b5 + (rp[2] + rm[2]) * c.g[k];
    }
    const size_t hw = (size_t)H * W;
    float* d = R + (size_t)blockIdx.z * 5 * hw + (size_t)y * W + x;
    d[hw] = b2 * c.ig[0];
    d[0] = b3 * c.ig[0];
    d[3 * hw] = b1 * c.ig[1] + b4 * c.ig[2];
    d[2 * hw] = b1 * c.ig[1] + b5 * c.ig[2];
    d[4 * hw] = b6 * c.ig[3];
}

// FarnebackUpdateMatrices.  R0, R1: [B][5][H][W]; flow: [B][2][H][W]; M: [B][5][H][W].
__global__ void fb_update_matrices_kernel(const float* __restrict__ R0, const float* __restrict__ R1, const float* __restrict__ flow, int H, int W,
                                          float* __restrict__ M)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const size_t hw = (size_t)H * W;
    const float* a = R0 + (size_t)blockIdx.y * 5 * hw + p;
    const float* b = R1 + (size_t)blockIdx.y * 5 * hw;
    const float dx = flow[(size_t)blockIdx.y * 2 * hw + p], dy = flow[(size_t)blockIdx.y * 2 * hw + hw + p];
    float fx = (float)x + dx, fy = (float)y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx = fx - (float)x1; fy = fy - (float)y1;
    float r2, r3, r4, r5, r6;
    if ((unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1)) {
        const float a00 = (1.0f - fx) * (1.0f - fy), a01 = fx * (1.0f - fy), a10 = (1.0f - fx) * fy, a11 = fx * fy;
        const float* q = b + (size_t)y1 * W + x1;
        r2 = a00 * q[0] + a01 * q[1] + a10 * q[W] + a11 * q[W + 1];
        q += hw; r3 = a00 * q[0] + a01 * q[1] + a10 * q[W] + a11 * q[W + 1];
        q += hw; r4 = a00 * q[0] + a01 * q[1] + a10 * q[W] + a11 * q[W + 1];
        q += hw; r5 = a00 * q[0] + a01 * q[1] + a10 * q[W] + a11 * q[W + 1];
        q += hw; r6 = a00 * q[0] + a01 * q[1] + a10 * q[W] + a11 * q[W + 1];
        r4 = (a[2 * hw] + r4) * 0.5f;
        r5 = (a[3 * hw] + r5) * 0.5f;
        r6 = (a[4 * hw] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.0f;
        r4 = a[2 * hw]; r5 = a[3 * hw]; r6 = a[4 * hw] * 0.5f;
    }
    r2 = (a[0] - r2) * 0.5f;
    r3 = (a[hw] - r3) * 0.5f;
    r2 = r2 + (r4 * dy + r6 * dx);
    r3 = r3 + (r6 * dy + r5 * dx);
    if ((unsigned)(x - 5) >= (unsigned)(W - 10) || (unsigned)(y - 5) >= (unsigned)(H - 10)) {
        const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        const float sc = (x < 5 ? border[x] : 1.0f) * (x >= W - 5 ? border[W - x - 1] : 1.0f) * (y < 5 ? border[y] : 1.0f) *
                         (y >= H - 5 ? border[H - y - 1] : 1.0f);
        r2 = r2 * sc; r3 = r3 * sc; r4 = r4 * sc; r5 = r5 * sc; r6 = r6 * sc;
    }
    float* m = M + (size_t)blockIdx.y * 5 * hw + p;
    m[0] = r4 * r4 + r6 * r6;
    m[hw] = (r4 + r5) * r6;
    m[2 * hw] = r5 * r5 + r6 * r6;
    m[3 * hw] = r4 * r2 + r6 * r3;
    m[4 * hw] = r6 * r2 + r5 * r3;
}

// Vertical half of the winsize x winsize box sums of M (rows clamped): V[c][y][x] = sum_j M[c][clamp(y+j)][x], first term first.
__global__ void fb_box_v_kernel(const float* __restrict__ M, int H, int W, int m, float* __restrict__ V)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const size_t hw = (size_t)H * W;
    for (int c = 0; c < 5; ++c) {
        const float* s = M + ((size_t)blockIdx.y * 5 + c) * hw + x;
        float a = s[(size_t)fb_clamp(y - m, 0, H - 1) * W];
        for (int j = -m + 1; j <= m; ++j) a = a + s[(size_t)fb_clamp(y + j, 0, H - 1) * W];
        V[((size_t)blockIdx.y * 5 + c) * hw + p] = a;
    }
}

// Horizontal half (columns replicated) + the per-pixel 2x2 solve in double (FarnebackUpdateFlow_Blur).  A 256-pixel row
// segment and its 2m halo go through LDS, five planes at a time.
constexpr int FB_HT = 256;
__global__ void __launch_bounds__(FB_HT) fb_box_h_solve_kernel(const float* __restrict__ V, int H, int W, int m, float* __restrict__ flow)
{
    __shared__ float seg[5][FB_HT + 2 * FB_MAX_WIN_R];
    const int y = blockIdx.y, x0 = blockIdx.x * FB_HT;
    const size_t hw = (size_t)H * W;
    for (int i = threadIdx.x; i < 5 * (FB_HT + 2 * m); i += FB_HT) {
        const int c = i / (FB_HT + 2 * m), lx = i - c * (FB_HT + 2 * m);
        seg[c][lx] = V[((size_t)blockIdx.z * 5 + c) * hw + (size_t)y * W + fb_clamp(x0 + lx - m, 0, W - 1)];
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= W) return;
    float h[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        float a = seg[c][threadIdx.x];
        for (int i = 1; i <= 2 * m; ++i) a = a + seg[c][threadIdx.x + i];
        h[c] = a;
    }
    const int win = 2 * m + 1;
    const double scale = 1.0 / (double)(win * win);
    const double g11 = h[0] * scale, g12 = h[1] * scale, g22 = h[2] * scale, h1 = h[3] * scale, h2 = h[4] * scale;
    const double idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3);
    flow[(size_t)blockIdx.z * 2 * hw + (size_t)y * W + x] = (float)((g11 * h2 - g12 * h1) * idet);
    flow[(size_t)blockIdx.z * 2 * hw + hw + (size_t)y * W + x] = (float)((g22 * h1 - g12 * h2) * idet);
}

// resize(prevFlow, INTER_LINEAR) to twice the size, times 1 / pyr_scale = 2 (oracle/farneback.c: up_taps)
__device__ __forceinline__ void fb_up_taps(int d, int n, int& i0, int& i1, float& w0, float& w1)
{
    int a = (d >> 1) - 1 + (d & 1);
    float wb = (d & 1) ? 0.25f : 0.75f;
    if (a < 0) { a = 0; wb = 0.0f; }
    if (a >= n - 1) { a = n - 1; wb = 0.0f; }
    i0 = a; i1 = (a + 1 < n) ? a + 1 : a; w1 = wb; w0 = 1.0f - wb;
}
__global__ void fb_upsample_kernel(const float* __restrict__ src, int Hs, int Ws, float* __restrict__ dst, int H, int W)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    fb_up_taps(y, Hs, y0, y1, wy0, wy1);
    fb_up_taps(x, Ws, x0, x1, wx0, wx1);
    for (int c = 0; c < 2; ++c) {
        const float* s = src + ((size_t)blockIdx.y * 2 + c) * Hs * Ws;
        const float t0 = s[(size_t)y0 * Ws + x0] * wx0 + s[(size_t)y0 * Ws + x1] * wx1;
        const float t1 = s[(size_t)y1 * Ws + x0] * wx0 + s[(size_t)y1 * Ws + x1] * wx1;
        dst[((size_t)blockIdx.y * 2 + c) * H * W + p] = (t0 * wy0 + t1 * wy1) * 2.0f;
    }
}

// dense field -> [x, y, dx, dy] vectors on the sampling grid, row-major; grid points where the field is exactly (0, 0) -- flat,
// identical neighbourhoods: no motion estimate, the counterpart of Lucas-Kanade reporting textured points only -- are dropped
// (the scorers normalise by the vector length).  One wavefront per image: ballot + prefix count keep the order.
__global__ void __launch_bounds__(64) fb_sample_kernel(const float* __restrict__ flow, int H, int W, int step, int K, float* __restrict__ vectors,
                                                       int* __restrict__ counts)
{
    const int ny = H / step, nx = W / step;
    const int n = ny * nx < K ? ny * nx : K;
    const size_t hw = (size_t)H * W;
    int out = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + threadIdx.x;
        float dx = 0.0f, dy = 0.0f;
        int x = 0, y = 0;
        if (i < n) {
            const int gy = i / nx, gx = i - gy * nx;
            y = step / 2 + gy * step; x = step / 2 + gx * step;
            dx = flow[(size_t)blockIdx.x * 2 * hw + (size_t)y * W + x];
            dy = flow[(size_t)blockIdx.x * 2 * hw + hw + (size_t)y * W + x];
        }
        const bool keep = i < n && (dx != 0.0f || dy != 0.0f);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int pos = out + __popcll(mask & ((1ull << threadIdx.x) - 1ull));
            float* v = vectors + ((size_t)blockIdx.x * K + pos) * 4;
            v[0] = (float)x; v[1] = (float)y; v[2] = dx; v[3] = dy;
        }
        out += __popcll(mask);
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = out;
}

}  // namespace eig
