// score_kernels.h -- motion scores of the flow vectors, one 128-thread workgroup per genome, float64.
//
// Replaces the scoring block of get_fitnesses_neat (/root/reference/generate_illusion.py:559-616) and the scorers
// it calls: plausibility_ratio (fitness_calculator.py:18-27), strength_number (:32-41),
// horizontal_symmetry_score (:81-120), swarm_score (:124-159), rotation_symmetry_score (:166-215).
// The reference's quirks are kept: (nx, nx) for the upper half in the Bands score,
// `% 2 * pi` precedence and arccos(x) in the swarm score, |dx| only in the strength, population variance,
// "more than 24 vectors" for circles, sentinel [[0,0,-1000,0]] when LK found nothing -> fitness 0.
// Sums are sequential in vector order (numpy uses pairwise sums; the difference is ~1e-16 relative).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eig {

constexpr int SCORE_T = 128;  // >= lk_max_corners

struct ScoreArgs {
    const float* vectors;  // [B][K][4]
    const int* counts;     // [B]
    int K;
    int structure;         // 0 Bands, 1 Circles, 2 Free, 3 CirclesFree (score_kernel); 4 inside_outside_score (inside_outside_kernel)
    int w, h;
    double* fitness;       // [B]
};

__device__ __forceinline__ double py_min1(double v) { return (1.0 < v) ? 1.0 : v; }  // Python min(v, 1)

// np.mean / np.var (ddof 0) of s[0..n)
__device__ double seq_mean(const double* s, int n)
{
    double t = 0.0;
    for (int i = 0; i < n; ++i) t += s[i];
    return t / (double)n;
}
__device__ double seq_var(const double* s, int n)
{
    const double m = seq_mean(s, n);
    double t = 0.0;
    for (int i = 0; i < n; ++i) { const double d = s[i] - m; t += fabs(d) * fabs(d); }
    return t / (double)n;
}

__global__ void __launch_bounds__(SCORE_T) score_kernel(const ScoreArgs a)
{
    __shared__ double vx[SCORE_T], vy[SCORE_T], vdx[SCORE_T], vdy[SCORE_T];
    __shared__ double t0[SCORE_T], t1[SCORE_T], t2[SCORE_T];
    __shared__ int s_keep[SCORE_T];
    __shared__ int s_n;
    const int b = blockIdx.x, t = threadIdx.x;
    int n = a.counts[b];
    if (n > a.K) n = a.K;
    const float* v = a.vectors + (size_t)b * a.K * 4;
    double x = 0, y = 0, dx = 0, dy = 0;
    if (n == 0) {  // generate_illusion.py:551-554
        n = 1;
        if (t == 0) { x = 0.0; y = 0.0; dx = -1000.0; dy = 0.0; }
    } else if (t < n) {
        x = (double)v[t * 4]; y = (double)v[t * 4 + 1]; dx = (double)v[t * 4 + 2]; dy = (double)v[t * 4 + 3];
    }
    const double limit = (a.structure == 0) ? 0.15 : (a.structure == 2) ? 0.4 : 0.3;
    const double xx = dx * dx, yy = dy * dy;
    const double norm = sqrt(xx + yy);
    s_keep[t] = (t < n && !(norm > limit)) ? 1 : 0;
    __syncthreads();
    if (t == 0) {  // order-preserving compaction (n <= 128)
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (s_keep[i]) s_keep[i] = ++m;
        s_n = m;
    }
    __syncthreads();
    if (t < n && s_keep[t]) {
        const int k = s_keep[t] - 1;
        vx[k] = x; vy[k] = y; vdx[k] = dx; vdy[k] = dy;
    }
    __syncthreads();
    const int m = s_n;
    double score = 0.0;

    if (a.structure == 1 || a.structure == 3) {
        if (m > 24) {
            // rotation_symmetry_score(good, w, h, [0, h/2])
            const double cxc = (double)a.w / 2.0, cyc = (double)a.h / 2.0, lim1 = (double)a.h / 2.0;
            double cx = 0, cy = 0, dist = 0;
            int keep = 0;
            if (t < m) {
                cx = vx[t] - cxc; cy = vy[t] - cyc;
                const double c2 = cx * cx, d2 = cy * cy;
                dist = sqrt(c2 + d2);
                keep = !((dist < 0.0) || (dist > lim1) || dist == 0.0);
            }
            s_keep[t] = keep;
            __syncthreads();
            if (t == 0) {
                int c = 0;
                for (int i = 0; i < m; ++i)
                    if (s_keep[i]) s_keep[i] = ++c;
                s_n = c;
            }
            __syncthreads();
            const int cnt = s_n;
            if (t < m && s_keep[t]) {
                const int k = s_keep[t] - 1;
                const double nn = sqrt(vdx[t] * vdx[t] + vdy[t] * vdy[t]);
                const double ndx = vdx[t] / nn, ndy = vdy[t] / nn;
                const double x1 = cx + ndx, y1 = cy + ndy;
                const double rx = (x1 * cx + y1 * cy) / dist;
                const double ry = (-x1 * cy + y1 * cx) / dist;
                t0[k] = rx - dist;
                t1[k] = ry;
            }
            __syncthreads();
            if (t == 0) {
                double rot = 0.0;
                if (cnt >= 2) {
                    const double var_x = seq_var(t0, cnt), var_y = seq_var(t1, cnt);
                    rot = ((1 - var_x) * (1 - var_x) + (1 - var_y) * (1 - var_y)) / 2;
                }
                // strength_number(good, 0.3)
                for (int i = 0; i < m; ++i) { t0[i] = fabs(vdx[i]); t1[i] = sqrt(vdx[i] * vdx[i] + vdy[i] * vdy[i]); }
                const double mx = seq_mean(t0, m);
                const double str = (mx / 0.3) * (1 - py_min1(seq_var(t1, m)));
                score = 0.7 * rot + 0.3 * str;
            }
        }
    } else if (a.structure == 2) {
        if (m > 0) {
            // swarm_score(good)
            double nx = 0, ny = 0;
            if (t < m) {
                const double nn = sqrt(vdx[t] * vdx[t] + vdy[t] * vdy[t]);
                nx = vdx[t] / nn; ny = vdy[t] / nn;
                t0[t] = acos(nx);  // angles
                t2[t] = nx;
            }
            __syncthreads();
            if (t < m) {
                const double PI = 3.141592653589793;
                const double va = acos(t2[t]);
                double total = 0.0;
                for (int j = 0; j < m; ++j) {
                    const double ddx = vx[j] - vx[t], ddy = vy[j] - vy[t];
                    double f = (ddx * ddx + ddy * ddy) / (100.0 * 100.0);
                    f = (f > 1.0) ? 1.0 : f;
                    const double close = 1.0 - ((f < 1.0) ? 0.0 : f);
                    double opt = va + f * PI;
                    opt = fmod(opt, 2.0);   // python %: operands are >= 0 here (NaN stays NaN)
                    opt = opt * PI;
                    total = total + close * fabs(t0[j] - opt);
                }
                t1[t] = (PI - total / (double)m) / PI;
            }
            __syncthreads();
            if (t == 0) {
                double s = 0.0;
                for (int i = 0; i < m; ++i) s = s + t1[i];
                const double swarm = s / (double)m;
                for (int i = 0; i < m; ++i) { t0[i] = fabs(vdx[i]); t1[i] = sqrt(vdx[i] * vdx[i] + vdy[i] * vdy[i]); }
                const double mx = seq_mean(t0, m);
                const double str = (mx / 0.4) * (1 - py_min1(seq_var(t1, m)));
                const double num = (double)(m < 15 ? m : 15) / 15.0;
                score = 0.5 * swarm + 0.1 * str + 0.4 * num;
            }
        }
    } else {  // Bands
        if (m > 0) {
            // horizontal_symmetry_score(good, [0, (h/4)*2])
            const double lim1 = ((double)a.h / 4.0) * 2.0;
            const int middle = (int)(lim1 / 2.0);
            int keep = 0;
            if (t < m) keep = !((vy[t] < 0.0) || (vy[t] > lim1));
            s_keep[t] = keep;
            __syncthreads();
            if (t == 0) {
                int c = 0;
                for (int i = 0; i < m; ++i)
                    if (s_keep[i]) s_keep[i] = ++c;
                s_n = c;
            }
            __syncthreads();
            const int cnt = s_n;
            if (t < m && s_keep[t]) {
                const int k = s_keep[t] - 1;
                const double nn = sqrt(vdx[t] * vdx[t] + vdy[t] * vdy[t]);
                const double nx = vdx[t] / nn, ny = vdy[t] / nn;
                const bool upper = vy[t] < (double)middle;
                t0[k] = upper ? nx : -nx;
                t1[k] = upper ? nx : ny;
            }
            __syncthreads();
            if (t == 0 && cnt > 0) {
                const double var_x = seq_var(t0, cnt);
                const double mean_x = fabs(seq_mean(t0, cnt)), mean_y = fabs(seq_mean(t1, cnt));
                score = ((1 - var_x) + mean_x + (1 - mean_y)) / 3;
            }
        }
    }
    if (t == 0) a.fitness[b] = score;
}

// inside_outside_score (fitness_calculator.py:219-304; a12): agreement of the vectors inside 5 x n cells plus disagreement
// with the neighbouring cells.  Unreachable at the reference's call sites (the else branch of the scoring block names an
// undefined `good_vectors`, generate_illusion.py:606-607), so it is a scorer of its own (eigen_score structure 4): ALL the
// given vectors, no plausibility filter, no sentinel.  Quirks kept: counts start at 1, neighbourhood x in [i-1, i],
// y in [j-1, min(h, i+1)) (`i + 1` for `j + 1`, :272), the cell itself excluded.  Vector sums run sequentially in vector
// order like the reference's loops; the two np.mean reductions are sequential here (pairwise in numpy: ~1e-16 relative).
constexpr int IO_MAX_CELLS = 6 * 64;

__global__ void __launch_bounds__(64) inside_outside_kernel(const ScoreArgs a)
{
    __shared__ double fx[IO_MAX_CELLS], fy[IO_MAX_CELLS], cnt[IO_MAX_CELLS], ax[IO_MAX_CELLS], ay[IO_MAX_CELLS], ns[IO_MAX_CELLS];
    __shared__ double sd[IO_MAX_CELLS];
    const int b = blockIdx.x, t = threadIdx.x;
    int n = a.counts[b];
    if (n > a.K) n = a.K;
    const float* v = a.vectors + (size_t)b * a.K * 4;
    const double step = (double)a.w / 5.0;
    const int w = (int)((double)a.w / step) + 1, h = (int)((double)a.h / step) + 1;
    const int cells = w * h;  // <= IO_MAX_CELLS, checked by the host
    for (int c = t; c < cells; c += 64) { fx[c] = 0; fy[c] = 0; cnt[c] = 1; ax[c] = 0; ay[c] = 0; ns[c] = 0; }
    __syncthreads();
    auto cell = [&](int k) {
        int i = (int)((double)v[k * 4] / step), j = (int)((double)v[k * 4 + 1] / step);
        i = i < 0 ? 0 : (i >= w ? w - 1 : i);  // positions are inside the image; the reference would wrap / raise otherwise
        j = j < 0 ? 0 : (j >= h ? h - 1 : j);
        return i * h + j;
    };
    if (t == 0) {
        for (int k = 0; k < n; ++k) {
            const int c = cell(k);
            const double dx = (double)v[k * 4 + 2], dy = (double)v[k * 4 + 3];
            fx[c] += dx; fy[c] += dy; cnt[c] += 1;
            ns[c] += sqrt(dx * dx + dy * dy);
        }
        for (int c = 0; c < cells; ++c) { fx[c] = fx[c] / cnt[c]; fy[c] = fy[c] / cnt[c]; ns[c] = ns[c] / cnt[c]; }
        for (int k = 0; k < n; ++k) {
            const int c = cell(k);
            const double dx = (double)v[k * 4 + 2], dy = (double)v[k * 4 + 3];
            ax[c] += (fx[c] - dx) * (fx[c] - dx);
            ay[c] += (fy[c] - dy) * (fy[c] - dy);
        }
        for (int c = 0; c < cells; ++c) { ax[c] = ax[c] / cnt[c]; ay[c] = ay[c] / cnt[c]; }
    }
    __syncthreads();
    for (int c = t; c < cells; c += 64) {
        const int i = c / h, j = c % h;
        double vx = fx[c], vy = fy[c];
        if (vx != 0 || vy != 0) { const double nv = sqrt(vx * vx + vy * vy); vx = vx / nv; vy = vy / nv; }
        const int min_i = i - 1 < 0 ? 0 : i - 1, max_i = i + 1 < w ? i + 1 : w;
        const int min_j = j - 1 < 0 ? 0 : j - 1, max_j = i + 1 < h ? i + 1 : h;
        int plus = 0, minus = 0;
        for (int x = min_i; x < max_i; ++x)
            for (int y = min_j; y < max_j; ++y) {
                if (x == i && y == j) continue;
                double wx = fx[x * h + y], wy = fy[x * h + y];
                if (wx != 0 || wy != 0) {
                    const double nw = sqrt(wx * wx + wy * wy);
                    wx = wx / nw; wy = wy / nw;
                    if (vx * wx + vy * wy > 0) ++plus; else ++minus;
                }
            }
        sd[c] = (double)((plus < 2 ? plus : 2) + (minus < 2 ? minus : 2)) / 4.0;
    }
    __syncthreads();
    if (t == 0) {
        double sa = 0, sn = 0, sum_d = 0;
        for (int c = 0; c < cells; ++c) { sa += ax[c]; sa += ay[c]; }   // np.mean over [w][h][2]: (x, y) interleaved
        for (int c = 0; c < cells; ++c) sn += ns[c];
        for (int c = 0; c < cells; ++c) sum_d += sd[c];
        const double mean_a = sa / (double)(2 * cells), mean_n = sn / (double)cells;
        const double score_agreement = -(mean_a < 10.0 ? mean_a : 10.0);
        const double score_size = (10.0 < mean_n) ? 10.0 : mean_n;   // Python min(10, x)
        sum_d = sum_d / (double)cells;
        sum_d = sum_d * 10;
        a.fitness[b] = (score_agreement + score_size + sum_d) / 30;
    }
}

}  // namespace eig
