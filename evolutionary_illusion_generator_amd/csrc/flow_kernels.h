// flow_kernels.h -- sparse Lucas-Kanade optical flow for a batch of image pairs, on device.
//
// Replaces lucas_kanade(img0, img1, ...) of Optical_Flow_Analyzer as called at
// /root/reference/generate_illusion.py:549-550 and /root/reference/fitness_calculator.py:498, i.e. OpenCV's
// cvtColor(BGR2GRAY) -> goodFeaturesToTrack(maxCorners, qualityLevel, minDistance, blockSize) ->
// calcOpticalFlowPyrLK(winSize, maxLevel, criteria) -> [x0, y0, x1-x0, y1-y0] for status == 1.
// Same canonical arithmetic as oracle/eig_oracle.c: all pixel / window sums are exact
// integers, the fp32 steps are spelled without contraction in OpenCV's order.
//
// All of it is HBM/L2-bound integer stencil work over uint8 images (a 256x256 gray image is 64 KB and stays in
// L2): corner response uses an LDS-staged derivative tile, corner selection is one workgroup per image with a
// wavefront-reduced arg-max, the tracker is one wavefront per feature with the 15x15 window spread over lanes
// and int64 wave reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eig {

constexpr int FLOW_MAX_LEVELS = 4;

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * n - 2 - i;
    return i;
}

// ---- cvtColor(BGR2GRAY), 8u, OpenCV 4.x 15-bit fixed point ----
__global__ void gray_kernel(const uint8_t* img, long long bstride, int C, int HW, uint8_t* gray, int B)
{
    const int b = blockIdx.y;
    const uint8_t* src = img + (size_t)b * bstride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        uint8_t g;
        if (C == 1) g = src[i];
        else g = (uint8_t)((src[i] * 9798 + src[HW + i] * 19235 + src[2 * HW + i] * 3735 + (1 << 14)) >> 15);
        gray[(size_t)b * HW + i] = g;
    }
}

// ---- pyrDown, 8u: [1 4 6 4 1]^2 / 256, (s + 128) >> 8, REFLECT_101 ----
__global__ void pyrdown_kernel(const uint8_t* src, int H, int W, uint8_t* dst, int Hd, int Wd)
{
    const int b = blockIdx.y;
    const uint8_t* s = src + (size_t)b * H * W;
    uint8_t* d = dst + (size_t)b * Hd * Wd;
    const int k[5] = {1, 4, 6, 4, 1};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hd * Wd; i += gridDim.x * blockDim.x) {
        const int y = i / Wd, x = i - y * Wd;
        int xs[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) xs[t] = reflect101(2 * x + t - 2, W);
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint8_t* row = s + (size_t)reflect101(2 * y + j - 2, H) * W;
            int rs = 0;
#pragma unroll
            for (int t = 0; t < 5; ++t) rs += k[t] * row[xs[t]];
            acc += k[j] * rs;
        }
        d[i] = (uint8_t)((acc + 128) >> 8);
    }
}

// ---- cornerMinEigenVal(gray, blockSize, ksize = 3) ----
// 16x16 outputs per block; Sobel derivatives of the (16+block-1)^2 neighbourhood staged in LDS as ints.
constexpr int EIG_T = 16;
constexpr int EIG_MAXB = 9;  // largest supported blockSize
__global__ void __launch_bounds__(256) mineig_kernel(const uint8_t* gray, int H, int W, int block, float SC, float* eig)
{
    __shared__ int sdx[(EIG_T + EIG_MAXB - 1) * (EIG_T + EIG_MAXB - 1)];
    __shared__ int sdy[(EIG_T + EIG_MAXB - 1) * (EIG_T + EIG_MAXB - 1)];
    const int b = blockIdx.z;
    const uint8_t* g = gray + (size_t)b * H * W;
    const int r0 = block / 2;
    const int TS = EIG_T + block - 1;
    const int y0 = blockIdx.y * EIG_T - r0, x0 = blockIdx.x * EIG_T - r0;
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int ty = i / TS, tx = i - ty * TS;
        // box filter border: the covariance image is reflected, i.e. take the derivative AT the reflected pixel
        const int y = reflect101(y0 + ty, H), x = reflect101(x0 + tx, W);
        const int ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
        const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W);
        const int p00 = g[ym * W + xm], p01 = g[ym * W + x], p02 = g[ym * W + xp];
        const int p10 = g[y * W + xm], p12 = g[y * W + xp];
        const int p20 = g[yp * W + xm], p21 = g[yp * W + x], p22 = g[yp * W + xp];
        sdx[i] = (p02 - p00) + 2 * (p12 - p10) + (p22 - p20);
        sdy[i] = (p20 - p00) + 2 * (p21 - p01) + (p22 - p02);
    }
    __syncthreads();
    const int ly = threadIdx.x / EIG_T, lx = threadIdx.x % EIG_T;
    const int y = blockIdx.y * EIG_T + ly, x = blockIdx.x * EIG_T + lx;
    if (y >= H || x >= W) return;
    int sxx = 0, sxy = 0, syy = 0;
    for (int j = 0; j < block; ++j)
        for (int i = 0; i < block; ++i) {
            const int a = sdx[(ly + j) * TS + lx + i], c = sdy[(ly + j) * TS + lx + i];
            sxx += a * a; sxy += a * c; syy += c * c;
        }
    const float fa = (float)sxx * SC, fb = (float)sxy * SC, fc = (float)syy * SC;
    const float a = fa * 0.5f, bb = fb, c = fc * 0.5f;
    const float d = a - c;
    const float t = d * d;
    const float u = bb * bb;
    eig[(size_t)b * H * W + y * W + x] = (a + c) - sqrtf(t + u);
}

// ---- goodFeaturesToTrack: threshold, 3x3 local maxima, greedy min-distance selection ----
// One 1024-thread block per image.  key = (float bits << 32) | pixel index: max key == OpenCV's sort order
// (value descending, ties by higher address first).
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

__global__ void __launch_bounds__(1024) corner_select_kernel(const float* eig, int H, int W, double quality, float min_dist, int max_corners,
                                                             unsigned long long* cand /*[B][H*W]*/, float* corners /*[B][K][2]*/, int* ncorners)
{
    __shared__ unsigned long long red[16];
    __shared__ int s_count;
    __shared__ unsigned long long s_best;
    const int b = blockIdx.x;
    const int HW = H * W;
    const float* e = eig + (size_t)b * HW;
    unsigned long long* cd = cand + (size_t)b * HW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // max over the image (floats may be slightly negative: compare as floats)
    float m = -3.4e38f;
    for (int i = tid; i < HW; i += 1024) m = fmaxf(m, e[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wv] = (unsigned long long)__float_as_uint(m);
    if (tid == 0) s_count = 0;
    __syncthreads();
    float maxv = __uint_as_float((unsigned)red[0]);
    for (int i = 1; i < 16; ++i) maxv = fmaxf(maxv, __uint_as_float((unsigned)red[i]));
    const float thr = (float)((double)maxv * quality);
    __syncthreads();
    // candidates: v > thr (THRESH_TOZERO) and v == 3x3 dilate, interior pixels only
    for (int i = tid; i < HW; i += 1024) {
        const int y = i / W, x = i - y * W;
        if (y < 1 || y >= H - 1 || x < 1 || x >= W - 1) continue;
        const float v = e[i];
        if (!(v > thr) || v == 0.0f) continue;
        bool ismax = true;
#pragma unroll
        for (int j = -1; j <= 1; ++j)
#pragma unroll
            for (int k = -1; k <= 1; ++k) ismax = ismax && !(e[i + j * W + k] > v);
        if (ismax) {
            const int slot = atomicAdd(&s_count, 1);
            cd[slot] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)i;
        }
    }
    __syncthreads();
    const int nc = s_count;
    const float md2 = min_dist * min_dist;
    int n = 0;
    while (n < max_corners) {
        unsigned long long best = 0;
        for (int i = tid; i < nc; i += 1024) {
            const unsigned long long k = cd[i];
            best = k > best ? k : best;
        }
        best = wave_max_u64(best);
        if (lane == 0) red[wv] = best;
        __syncthreads();
        if (tid == 0) {
            unsigned long long bb = red[0];
            for (int i = 1; i < 16; ++i) bb = red[i] > bb ? red[i] : bb;
            s_best = bb;
        }
        __syncthreads();
        const unsigned long long win = s_best;
        if (win == 0) break;
        const int idx = (int)(win & 0xFFFFFFFFull);
        const int wy = idx / W, wx = idx - wy * W;
        if (tid == 0) {
            corners[((size_t)b * max_corners + n) * 2] = (float)wx;
            corners[((size_t)b * max_corners + n) * 2 + 1] = (float)wy;
        }
        for (int i = tid; i < nc; i += 1024) {
            const unsigned long long k = cd[i];
            if (k == 0) continue;
            const int ci = (int)(k & 0xFFFFFFFFull);
            const int cy = ci / W, cx = ci - cy * W;
            const float dx = (float)cx - (float)wx, dy = (float)cy - (float)wy;
            const bool kill = (min_dist >= 1.0f) ? (dx * dx + dy * dy < md2) : (ci == idx);
            if (kill) cd[i] = 0;
        }
        ++n;
        __syncthreads();
    }
    if (tid == 0) ncorners[b] = n;
}

// ---- calcSharrDeriv: int16 (dx, dy), REFLECT_101 ----
__global__ void scharr_kernel(const uint8_t* gray, int H, int W, short2* deriv)
{
    const int b = blockIdx.y;
    const uint8_t* g = gray + (size_t)b * H * W;
    short2* d = deriv + (size_t)b * H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const int y0 = reflect101(y - 1, H), y2 = reflect101(y + 1, H);
        const int x0 = reflect101(x - 1, W), x2 = reflect101(x + 1, W);
        const int a00 = g[y0 * W + x0], a01 = g[y0 * W + x], a02 = g[y0 * W + x2];
        const int a10 = g[y * W + x0], a12 = g[y * W + x2];
        const int a20 = g[y2 * W + x0], a21 = g[y2 * W + x], a22 = g[y2 * W + x2];
        short2 v;
        v.x = (short)(3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20));
        v.y = (short)(3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02));
        d[i] = v;
    }
}

struct LKArgs {
    const uint8_t* I[FLOW_MAX_LEVELS];
    const uint8_t* J[FLOW_MAX_LEVELS];
    const short2* dI[FLOW_MAX_LEVELS];
    int Hs[FLOW_MAX_LEVELS], Ws[FLOW_MAX_LEVELS];
    int max_level;  // effective (levels that exist)
    int win, max_iter, K;
    double eps_sq, min_eig_thr;
    const float* corners;  // [B][K][2]
    const int* ncorners;   // [B]
    float* next_pts;       // [B][K][2]
    uint8_t* status;       // [B][K]
};

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define EIG_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

// LKTrackerInvoker: one wavefront per feature, window positions spread over lanes (<= 4 per lane, win <= 16)
__global__ void __launch_bounds__(64) lk_track_kernel(const LKArgs a)
{
    const int f = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (f >= a.ncorners[b]) return;
    const int win = a.win, nwin = win * win;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.0f / (float)(1 << 20);
    const float c_x = a.corners[((size_t)b * a.K + f) * 2], c_y = a.corners[((size_t)b * a.K + f) * 2 + 1];
    float out_x = 0.f, out_y = 0.f;
    bool st = true;
    int wy[4], wx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        wy[i] = idx / win;
        wx[i] = idx - wy[i] * win;
    }
    for (int level = a.max_level; level >= 0; --level) {
        const int h = a.Hs[level], w = a.Ws[level];
        const uint8_t* I = a.I[level] + (size_t)b * h * w;
        const uint8_t* J = a.J[level] + (size_t)b * h * w;
        const short2* D = a.dI[level] + (size_t)b * h * w;
        const float lscale = (float)(1.0 / (double)(1 << level));
        float px = c_x * lscale, py = c_y * lscale;
        float nx, ny;
        if (level == a.max_level) { nx = px; ny = py; }
        else { nx = out_x * 2.0f; ny = out_y * 2.0f; }
        out_x = nx; out_y = ny;
        px -= half; py -= half;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
            if (level == 0) st = false;
            continue;
        }
        float fa = px - (float)ipx, fb = py - (float)ipy;
        int iw00 = __float2int_rn((1.0f - fa) * (1.0f - fb) * 16384.0f);
        int iw01 = __float2int_rn(fa * (1.0f - fb) * 16384.0f);
        int iw10 = __float2int_rn((1.0f - fa) * fb * 16384.0f);
        int iw11 = 16384 - iw00 - iw01 - iw10;
        int Iw[4], dxw[4], dyw[4];
        long long sA11 = 0, sA12 = 0, sA22 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Iw[i] = 0; dxw[i] = 0; dyw[i] = 0;
            if (lane + 64 * i < nwin) {
                const int yy = wy[i] + ipy, xx = wx[i] + ipx;
                const int ry0 = reflect101(yy, h), ry1 = reflect101(yy + 1, h);
                const int rx0 = reflect101(xx, w), rx1 = reflect101(xx + 1, w);
                const int ival = EIG_DESCALE(I[ry0 * w + rx0] * iw00 + I[ry0 * w + rx1] * iw01 + I[ry1 * w + rx0] * iw10 + I[ry1 * w + rx1] * iw11, 14 - 5);
                // derivative border is constant 0 (copyMakeBorder BORDER_CONSTANT)
                const bool y0in = yy >= 0 && yy < h, y1in = yy + 1 >= 0 && yy + 1 < h;
                const bool x0in = xx >= 0 && xx < w, x1in = xx + 1 >= 0 && xx + 1 < w;
                short2 z; z.x = 0; z.y = 0;
                const short2 d00 = (y0in && x0in) ? D[yy * w + xx] : z;
                const short2 d01 = (y0in && x1in) ? D[yy * w + xx + 1] : z;
                const short2 d10 = (y1in && x0in) ? D[(yy + 1) * w + xx] : z;
                const short2 d11 = (y1in && x1in) ? D[(yy + 1) * w + xx + 1] : z;
                const int ixval = EIG_DESCALE(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, 14);
                const int iyval = EIG_DESCALE(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, 14);
                Iw[i] = (short)ival; dxw[i] = (short)ixval; dyw[i] = (short)iyval;
                sA11 += (long long)dxw[i] * dxw[i]; sA12 += (long long)dxw[i] * dyw[i]; sA22 += (long long)dyw[i] * dyw[i];
            }
        }
        sA11 = wave_sum_i64(sA11); sA12 = wave_sum_i64(sA12); sA22 = wave_sum_i64(sA22);
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        const float m0 = A11 * A22, m1 = A12 * A12;
        float Dm = m0 - m1;
        const float dd = A11 - A22;
        const float mn = A22 + A11;
        const float qq = dd * dd;
        const float r4 = 4.0f * A12;
        const float rr = r4 * A12;
        const float minEig = (mn - sqrtf(qq + rr)) / (float)(2 * win * win);
        if ((double)minEig < a.min_eig_thr || Dm < 1.1920928955078125e-7f) {
            if (level == 0) st = false;
            continue;
        }
        Dm = 1.0f / Dm;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_iter; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= w || iny < -win || iny >= h) {
                if (level == 0) st = false;
                break;
            }
            fa = nx - (float)inx; fb = ny - (float)iny;
            iw00 = __float2int_rn((1.0f - fa) * (1.0f - fb) * 16384.0f);
            iw01 = __float2int_rn(fa * (1.0f - fb) * 16384.0f);
            iw10 = __float2int_rn((1.0f - fa) * fb * 16384.0f);
            iw11 = 16384 - iw00 - iw01 - iw10;
            long long sb1 = 0, sb2 = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (lane + 64 * i < nwin) {
                    const int yy = wy[i] + iny, xx = wx[i] + inx;
                    const int ry0 = reflect101(yy, h), ry1 = reflect101(yy + 1, h);
                    const int rx0 = reflect101(xx, w), rx1 = reflect101(xx + 1, w);
                    const int jv = EIG_DESCALE(J[ry0 * w + rx0] * iw00 + J[ry0 * w + rx1] * iw01 + J[ry1 * w + rx0] * iw10 + J[ry1 * w + rx1] * iw11, 14 - 5);
                    const int diff = jv - Iw[i];
                    sb1 += (long long)diff * dxw[i];
                    sb2 += (long long)diff * dyw[i];
                }
            }
            sb1 = wave_sum_i64(sb1); sb2 = wave_sum_i64(sb2);
            const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            const float t0 = A12 * b2, t1 = A22 * b1, t2 = A12 * b1, t3 = A11 * b2;
            const float dxv = (t0 - t1) * Dm;
            const float dyv = (t2 - t3) * Dm;
            nx += dxv; ny += dyv;
            out_x = nx + half; out_y = ny + half;
            if ((double)dxv * (double)dxv + (double)dyv * (double)dyv <= a.eps_sq) break;
            if (j > 0 && (double)fabsf(dxv + pdx) < 0.01 && (double)fabsf(dyv + pdy) < 0.01) {
                out_x -= dxv * 0.5f; out_y -= dyv * 0.5f;
                break;
            }
            pdx = dxv; pdy = dyv;
        }
        if (st && level == 0) {
            const float fx = out_x - half, fy = out_y - half;
            const int ix = (int)floorf(fx), iy = (int)floorf(fy);
            if (ix < -win || ix >= w || iy < -win || iy >= h) st = false;
        }
    }
    if (lane == 0) {
        a.next_pts[((size_t)b * a.K + f) * 2] = out_x;
        a.next_pts[((size_t)b * a.K + f) * 2 + 1] = out_y;
        a.status[(size_t)b * a.K + f] = st ? 1 : 0;
    }
}

// vectors [x0, y0, x1-x0, y1-y0] for status == 1, in feature order; one 128-thread block per image (K <= 128)
__global__ void __launch_bounds__(128) compact_vectors_kernel(const float* corners, const float* next_pts, const uint8_t* status,
                                                              const int* ncorners, int K, float* vectors, int* counts)
{
    __shared__ int s_pos[128];
    const int b = blockIdx.x, t = threadIdx.x;
    const int n = ncorners[b];
    const int ok = (t < n && t < K) ? (int)status[(size_t)b * K + t] : 0;
    s_pos[t] = ok;
    __syncthreads();
    for (int o = 1; o < 128; o <<= 1) {  // inclusive scan
        const int v = (t >= o) ? s_pos[t - o] : 0;
        __syncthreads();
        s_pos[t] += v;
        __syncthreads();
    }
    if (ok) {
        const int slot = s_pos[t] - 1;
        const float x0 = corners[((size_t)b * K + t) * 2], y0 = corners[((size_t)b * K + t) * 2 + 1];
        float* v = vectors + ((size_t)b * K + slot) * 4;
        v[0] = x0; v[1] = y0;
        v[2] = next_pts[((size_t)b * K + t) * 2] - x0;
        v[3] = next_pts[((size_t)b * K + t) * 2 + 1] - y0;
    }
    if (t == 127) counts[b] = s_pos[127];
}

}  // namespace eig
