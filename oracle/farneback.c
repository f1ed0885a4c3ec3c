/*
 * farneback.c -- CPU ORACLE (test infrastructure, NOT product code): dense optical flow after G. Farneback,
 * "Two-Frame Motion Estimation Based on Polynomial Expansion" (SCIA 2003), in the form OpenCV ships it
 * (cv::calcOpticalFlowFarneback, modules/video/src/optflowgf.cpp) with the parameters of OpenCV's dense-flow
 * tutorial: pyr_scale 0.5, levels 3, winsize 15, iterations 3, poly_n 5, poly_sigma 1.2, flags 0.
 *
 * Status: the reference never calls a dense-flow routine -- its only flow call is lucas_kanade
 * (/root/reference/generate_illusion.py:549-550, /root/reference/fitness_calculator.py:498); the option exists because
 * BASELINE.json's north_star names "Farneback/Lucas-Kanade flow" (SURVEY.md 8(f) row 4).  PARITY UNPINNED: OpenCV is
 * not installed and the reference holds no dense-flow fixture; what is restated is the published algorithm, and
 * the step from a dense field to the [x, y, dx, dy] vectors the scorers consume is build-defined (the sampling
 * grid of OpenCV's own samples/python/opt_flow.py: draw_flow, every `step` pixels starting at step/2).
 *
 * Canonical arithmetic (so that the HIP kernels reproduce it bit for bit): every step below is fp32 with one
 * rounding per written operation (compile with -ffp-contract=off), sums in the written order; the 2x2 solve per
 * pixel is double precision as in OpenCV; Gaussian coefficients come from double-precision exp() and are rounded to
 * float once.  OpenCV's own running-sum box filter and SIMD paths accumulate in other orders; that difference is of the
 * summation-order kind.
 *
 * Only tests/ may load this code (through oracle/libeig_oracle.so).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FB_MAX_LEVELS 8
#define FB_MAX_POLY_N 7
#define FB_MAX_BLUR_R 32

typedef struct {
    int levels;        /* 3: pyramid levels ON TOP of the full resolution (fewer when a level would be < 32 px) */
    int winsize;       /* 15 */
    int iterations;    /* 3 */
    int poly_n;        /* 5 */
    double poly_sigma; /* 1.2 */
    int step;          /* sampling step of the vector grid (16) */
    int max_vectors;   /* capacity of the output */
} fb_params_t;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int reflect101_(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
static inline int cv_round_d(double v) { return (int)lrint(v); } /* cvRound: half to even */

/* getGaussianKernel(ksize, sigma, CV_32F): the symmetric half k[0..r]; ksize 3 with sigma <= 0 is the fixed {0.25, 0.5, 0.25} */
static void gaussian_half_kernel(int ksize, double sigma, float* k /* [r+1], k[0] = centre */)
{
    const int r = ksize / 2;
    if (sigma <= 0 && ksize == 3) { k[0] = 0.5f; k[1] = 0.25f; return; }
    const double sx = sigma > 0 ? sigma : ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double s2 = -0.5 / (sx * sx);
    double c[2 * FB_MAX_BLUR_R + 1], sum = 0;
    for (int i = 0; i < ksize; i++) { const double x = i - (ksize - 1) * 0.5; c[i] = exp(s2 * x * x); sum += c[i]; }
    sum = 1.0 / sum;
    for (int j = 0; j <= r; j++) k[j] = (float)(c[r + j] * sum);
}

/* FarnebackPrepareGaussian: g, xg, xxg [0..n] (x >= 0 half; xg is odd, g and xxg even) and the four entries of inv(G) */
static void prepare_poly(int n, double sigma, float* g, float* xg, float* xxg, float* ig /* 11, 03, 33, 55 */)
{
    if (sigma < 1.1920928955078125e-07) sigma = n * 0.3;
    double gd[2 * FB_MAX_POLY_N + 1], s = 0;
    for (int x = -n; x <= n; x++) { gd[x + n] = exp(-x * x / (2 * sigma * sigma)); s += gd[x + n]; }
    s = 1.0 / s;
    float gf[2 * FB_MAX_POLY_N + 1];
    for (int x = -n; x <= n; x++) gf[x + n] = (float)(gd[x + n] * s);
    for (int x = 0; x <= n; x++) { g[x] = gf[x + n]; xg[x] = (float)(x * gf[x + n]); xxg[x] = (float)(x * x * gf[x + n]); }
    double G[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            const double w = (double)gf[y + n] * (double)gf[x + n];
            G[0][0] += w;
            G[1][1] += w * x * x;
            G[3][3] += w * x * x * x * x;
            G[5][5] += w * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    /* inverse by Gauss-Jordan with partial pivoting (G is SPD and tiny) */
    double A[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 12; j++) A[i][j] = j < 6 ? G[i][j] : (j - 6 == i ? 1.0 : 0.0);
    for (int c = 0; c < 6; c++) {
        int p = c;
        for (int r = c + 1; r < 6; r++) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (p != c) for (int j = 0; j < 12; j++) { const double t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
        const double d = 1.0 / A[c][c];
        for (int j = 0; j < 12; j++) A[c][j] *= d;
        for (int r = 0; r < 6; r++) {
            if (r == c) continue;
            const double f = A[r][c];
            if (f != 0.0) for (int j = 0; j < 12; j++) A[r][j] -= f * A[c][j];
        }
    }
    ig[0] = (float)A[1][7]; ig[1] = (float)A[0][9]; ig[2] = (float)A[3][9]; ig[3] = (float)A[5][11];
}

/* exported so that the HIP engine's host code can be checked against the same constants */
void eig_oracle_fb_constants(int poly_n, double poly_sigma, float* g, float* xg, float* xxg, float* ig)
{
    prepare_poly(poly_n, poly_sigma, g, xg, xxg, ig);
}

/* convertTo(CV_32F) -> GaussianBlur(ksize, sigma) (BORDER_REFLECT_101, separable: rows then columns, symmetric form)
 * -> resize(INTER_LINEAR) by the exact factor 2^k: the centre 2x2 of every 2^k block, weights 0.5 / 0.5 */
static void blur_down(const uint8_t* img, int H, int W, int k, float* out, int Hk, int Wk)
{
    const double scale = 1.0 / (double)(1 << k);
    const double sigma = (1.0 / scale - 1) * 0.5;
    int ksize = cv_round_d(sigma * 5) | 1;
    if (ksize < 3) ksize = 3;
    const int r = ksize / 2;
    float kk[FB_MAX_BLUR_R + 1];
    gaussian_half_kernel(ksize, sigma, kk);
    float* T = (float*)malloc(sizeof(float) * (size_t)H * W);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float a = kk[0] * (float)img[(size_t)y * W + x];
            for (int j = 1; j <= r; j++) {
                const float s = (float)img[(size_t)y * W + reflect101_(x - j, W)] + (float)img[(size_t)y * W + reflect101_(x + j, W)];
                a = a + kk[j] * s;
            }
            T[(size_t)y * W + x] = a;
        }
    float* Bl = (float*)malloc(sizeof(float) * (size_t)H * W);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float a = kk[0] * T[(size_t)y * W + x];
            for (int j = 1; j <= r; j++) {
                const float s = T[(size_t)reflect101_(y - j, H) * W + x] + T[(size_t)reflect101_(y + j, H) * W + x];
                a = a + kk[j] * s;
            }
            Bl[(size_t)y * W + x] = a;
        }
    if (k == 0) memcpy(out, Bl, sizeof(float) * (size_t)H * W);
    else {
        const int f = 1 << k, o = f / 2 - 1;
        for (int y = 0; y < Hk; y++)
            for (int x = 0; x < Wk; x++) {
                const float* p0 = Bl + (size_t)(y * f + o) * W + x * f + o;
                const float* p1 = p0 + W;
                const float h0 = p0[0] * 0.5f + p0[1] * 0.5f;
                const float h1 = p1[0] * 0.5f + p1[1] * 0.5f;
                out[(size_t)y * Wk + x] = h0 * 0.5f + h1 * 0.5f;
            }
    }
    free(T); free(Bl);
}

/* FarnebackPolyExp: R[y][x][5] = (r3', r2', r5', r4', r6') in OpenCV's storage order (x-gradient first) */
static void poly_exp(const float* I, int H, int W, int n, const float* g, const float* xg, const float* xxg, const float* ig, float* R)
{
    float* row = (float*)malloc(sizeof(float) * (size_t)W * 3);
    for (int y = 0; y < H; y++) {
        /* vertical part (rows clamped) */
        for (int x = 0; x < W; x++) { row[x * 3] = I[(size_t)y * W + x] * g[0]; row[x * 3 + 1] = 0.0f; row[x * 3 + 2] = 0.0f; }
        for (int k = 1; k <= n; k++) {
            const float* s0 = I + (size_t)clampi(y - k, 0, H - 1) * W;
            const float* s1 = I + (size_t)clampi(y + k, 0, H - 1) * W;
            for (int x = 0; x < W; x++) {
                const float p = s0[x] + s1[x];
                const float d = s1[x] - s0[x];
                row[x * 3] = row[x * 3] + g[k] * p;
                row[x * 3 + 1] = row[x * 3 + 1] + xg[k] * d;
                row[x * 3 + 2] = row[x * 3 + 2] + xxg[k] * p;
            }
        }
        /* horizontal part (columns replicated) */
        for (int x = 0; x < W; x++) {
            float b1 = row[x * 3] * g[0], b2 = 0.0f, b3 = row[x * 3 + 1] * g[0], b4 = 0.0f, b5 = row[x * 3 + 2] * g[0], b6 = 0.0f;
            for (int k = 1; k <= n; k++) {
                const float* rp = row + clampi(x + k, 0, W - 1) * 3;
                const float* rm = row + clampi(x - k, 0, W - 1) * 3;
                const float tg = rp[0] + rm[0];
                b1 = b1 + tg * g[k];
                b4 = b4 + tg * xxg[k];
                b2 = b2 + (rp[0] - rm[0]) * xg[k];
                b3 = b3 + (rp[1] + rm[1]) * g[k];
                b6 = b6 + (rp[1] - rm[1]) * xg[k];
                b5 = b5 + (rp[2] + rm[2]) * g[k];
            }
            float* d = R + ((size_t)y * W + x) * 5;
            d[1] = b2 * ig[0];
            d[0] = b3 * ig[0];
            d[3] = b1 * ig[1] + b4 * ig[2];
            d[2] = b1 * ig[1] + b5 * ig[2];
            d[4] = b6 * ig[3];
        }
    }
    free(row);
}

static const float FB_BORDER[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};

/* FarnebackUpdateMatrices over all rows */
static void update_matrices(const float* R0, const float* R1, const float* flow, int H, int W, float* M)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float* a = R0 + ((size_t)y * W + x) * 5;
            const float dx = flow[((size_t)y * W + x) * 2], dy = flow[((size_t)y * W + x) * 2 + 1];
            float fx = (float)x + dx, fy = (float)y + dy;
            const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
            float r2, r3, r4, r5, r6;
            fx = fx - (float)x1; fy = fy - (float)y1;
            if ((unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1)) {
                const float a00 = (1.0f - fx) * (1.0f - fy), a01 = fx * (1.0f - fy), a10 = (1.0f - fx) * fy, a11 = fx * fy;
                const float* p = R1 + ((size_t)y1 * W + x1) * 5;
                const float* q = p + (size_t)W * 5;
                r2 = a00 * p[0] + a01 * p[5] + a10 * q[0] + a11 * q[5];
                r3 = a00 * p[1] + a01 * p[6] + a10 * q[1] + a11 * q[6];
                r4 = a00 * p[2] + a01 * p[7] + a10 * q[2] + a11 * q[7];
                r5 = a00 * p[3] + a01 * p[8] + a10 * q[3] + a11 * q[8];
                r6 = a00 * p[4] + a01 * p[9] + a10 * q[4] + a11 * q[9];
                r4 = (a[2] + r4) * 0.5f;
                r5 = (a[3] + r5) * 0.5f;
                r6 = (a[4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.0f;
                r4 = a[2]; r5 = a[3]; r6 = a[4] * 0.5f;
            }
            r2 = (a[0] - r2) * 0.5f;
            r3 = (a[1] - r3) * 0.5f;
            r2 = r2 + (r4 * dy + r6 * dx);
            r3 = r3 + (r6 * dy + r5 * dx);
            if ((unsigned)(x - 5) >= (unsigned)(W - 10) || (unsigned)(y - 5) >= (unsigned)(H - 10)) {
                const float sc = (x < 5 ? FB_BORDER[x] : 1.0f) * (x >= W - 5 ? FB_BORDER[W - x - 1] : 1.0f) *
                                 (y < 5 ? FB_BORDER[y] : 1.0f) * (y >= H - 5 ? FB_BORDER[H - y - 1] : 1.0f);
                r2 = r2 * sc; r3 = r3 * sc; r4 = r4 * sc; r5 = r5 * sc; r6 = r6 * sc;
            }
            float* m = M + ((size_t)y * W + x) * 5;
            m[0] = r4 * r4 + r6 * r6;
            m[1] = (r4 + r5) * r6;
            m[2] = r5 * r5 + r6 * r6;
            m[3] = r4 * r2 + r6 * r3;
            m[4] = r6 * r2 + r5 * r3;
        }
}

/* FarnebackUpdateFlow_Blur: winsize x winsize box sums of M (rows clamped, columns replicated; direct sums, first term first),
 * then the 2x2 solve in double */
static void update_flow(const float* M, int H, int W, int win, float* flow, float* V /* scratch [H][W][5] */)
{
    const int m = win / 2;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 5; c++) {
                float s = M[((size_t)clampi(y - m, 0, H - 1) * W + x) * 5 + c];
                for (int j = -m + 1; j <= m; j++) s = s + M[((size_t)clampi(y + j, 0, H - 1) * W + x) * 5 + c];
                V[((size_t)y * W + x) * 5 + c] = s;
            }
    const double scale = 1.0 / (double)(win * win);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float h[5];
            for (int c = 0; c < 5; c++) {
                float s = V[((size_t)y * W + clampi(x - m, 0, W - 1)) * 5 + c];
                for (int i = -m + 1; i <= m; i++) s = s + V[((size_t)y * W + clampi(x + i, 0, W - 1)) * 5 + c];
                h[c] = s;
            }
            const double g11 = h[0] * scale, g12 = h[1] * scale, g22 = h[2] * scale, h1 = h[3] * scale, h2 = h[4] * scale;
            const double idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3);
            flow[((size_t)y * W + x) * 2] = (float)((g11 * h2 - g12 * h1) * idet);
            flow[((size_t)y * W + x) * 2 + 1] = (float)((g22 * h1 - g12 * h2) * idet);
        }
}

/* resize(prevFlow, INTER_LINEAR) to twice the size, then * (1 / pyr_scale) = 2: source coordinate d/2 - 0.25 -> taps
 * (m-1, m) with weights (0.25, 0.75) for even d = 2m and (m, m+1) with (0.75, 0.25) for odd d; outside the source the
 * border pixel gets weight 1 (cv::resize: fx = 0) */
static inline void up_taps(int d, int n, int* i0, int* i1, float* w0, float* w1)
{
    int a = (d >> 1) - 1 + (d & 1);
    float wb = (d & 1) ? 0.25f : 0.75f;
    if (a < 0) { a = 0; wb = 0.0f; }
    if (a >= n - 1) { a = n - 1; wb = 0.0f; }
    *i0 = a; *i1 = (a + 1 < n) ? a + 1 : a; *w1 = wb; *w0 = 1.0f - wb;
}
static void upsample_flow(const float* src, int Hs, int Ws, float* dst, int H, int W)
{
    for (int y = 0; y < H; y++) {
        int y0, y1; float wy0, wy1;
        up_taps(y, Hs, &y0, &y1, &wy0, &wy1);
        for (int x = 0; x < W; x++) {
            int x0, x1; float wx0, wx1;
            up_taps(x, Ws, &x0, &x1, &wx0, &wx1);
            for (int c = 0; c < 2; c++) {
                const float t0 = src[((size_t)y0 * Ws + x0) * 2 + c] * wx0 + src[((size_t)y0 * Ws + x1) * 2 + c] * wx1;
                const float t1 = src[((size_t)y1 * Ws + x0) * 2 + c] * wx0 + src[((size_t)y1 * Ws + x1) * 2 + c] * wx1;
                dst[((size_t)y * W + x) * 2 + c] = (t0 * wy0 + t1 * wy1) * 2.0f;
            }
        }
    }
}

/* number of pyramid levels actually used on top of level 0 (calcOpticalFlowFarneback: min_size = 32) */
int eig_oracle_fb_levels(int H, int W, int levels)
{
    int k = 0;
    double scale = 1;
    for (; k < levels; k++) {
        scale *= 0.5;
        if (W * scale < 32 || H * scale < 32) break;
    }
    return k;
}

/* Dense flow g0 -> g1 (uint8 gray [H][W], H and W multiples of 2^levels_used).  flow: float [H][W][2] (dx, dy). */
int eig_oracle_farneback(const uint8_t* g0, const uint8_t* g1, int H, int W, const fb_params_t* p, float* flow)
{
    if (p->poly_n < 1 || p->poly_n > FB_MAX_POLY_N || p->winsize < 1 || !(p->winsize & 1)) return -1;
    const int levels = eig_oracle_fb_levels(H, W, p->levels);
    if (levels >= FB_MAX_LEVELS || (H % (1 << levels)) || (W % (1 << levels))) return -1;
    float g[FB_MAX_POLY_N + 1], xg[FB_MAX_POLY_N + 1], xxg[FB_MAX_POLY_N + 1], ig[4];
    prepare_poly(p->poly_n, p->poly_sigma, g, xg, xxg, ig);
    const size_t hw = (size_t)H * W;
    float* I = (float*)malloc(sizeof(float) * hw);
    float* R0 = (float*)malloc(sizeof(float) * hw * 5);
    float* R1 = (float*)malloc(sizeof(float) * hw * 5);
    float* M = (float*)malloc(sizeof(float) * hw * 5);
    float* V = (float*)malloc(sizeof(float) * hw * 5);
    float* fl = (float*)malloc(sizeof(float) * hw * 2);
    float* prev = (float*)malloc(sizeof(float) * hw * 2);
    int Hp = 0, Wp = 0;
    for (int k = levels; k >= 0; k--) {
        const int Hk = H >> k, Wk = W >> k;
        if (k == levels) memset(fl, 0, sizeof(float) * (size_t)Hk * Wk * 2);
        else upsample_flow(prev, Hp, Wp, fl, Hk, Wk);
        blur_down(g0, H, W, k, I, Hk, Wk);
        poly_exp(I, Hk, Wk, p->poly_n, g, xg, xxg, ig, R0);
        blur_down(g1, H, W, k, I, Hk, Wk);
        poly_exp(I, Hk, Wk, p->poly_n, g, xg, xxg, ig, R1);
        update_matrices(R0, R1, fl, Hk, Wk, M);
        for (int i = 0; i < p->iterations; i++) {
            update_flow(M, Hk, Wk, p->winsize, fl, V);
            if (i < p->iterations - 1) update_matrices(R0, R1, fl, Hk, Wk, M);
        }
        memcpy(prev, fl, sizeof(float) * (size_t)Hk * Wk * 2);
        Hp = Hk; Wp = Wk;
    }
    memcpy(flow, fl, sizeof(float) * hw * 2);
    free(I); free(R0); free(R1); free(M); free(V); free(fl); free(prev);
    return 0;
}

/* Sampling step of the vector grid: the smallest multiple of p->step whose grid fits max_vectors */
int eig_oracle_fb_grid_step(int H, int W, int step, int max_vectors)
{
    int s = step;
    while ((H / s) * (W / s) > max_vectors) s += step;
    return s;
}

/* vectors [n][4] = [x, y, dx, dy] at (s/2 + i*s, s/2 + j*s), row-major, without the grid points whose flow is exactly (0, 0); returns n */
int eig_oracle_fb_vectors(const float* flow, int H, int W, const fb_params_t* p, float* vec)
{
    const int s = eig_oracle_fb_grid_step(H, W, p->step, p->max_vectors);
    int n = 0;
    for (int y = s / 2; y < H; y += s)
        for (int x = s / 2; x < W; x += s) {
            if ((y - s / 2) / s >= H / s || (x - s / 2) / s >= W / s) continue;
            const float dx = flow[((size_t)y * W + x) * 2], dy = flow[((size_t)y * W + x) * 2 + 1];
            if (dx == 0.0f && dy == 0.0f) continue; /* flat, identical neighbourhoods: no motion estimate (the scorers divide by the length) */
            vec[n * 4] = (float)x; vec[n * 4 + 1] = (float)y;
            vec[n * 4 + 2] = dx; vec[n * 4 + 3] = dy;
            n++;
        }
    return n;
}
