/*
 * eig_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the two compiled-library stages of EIGen's fitness path:
 *
 *   (1) PredNet forward roll-out (20 repeats + self-fed extension frames), the algorithm
 *       behind `test_prednet(...)` called at /root/reference/generate_illusion.py:533-537 and
 *       /root/reference/fitness_calculator.py:487-491.
 *   (2) Sparse Lucas-Kanade flow, the algorithm behind `lucas_kanade(...)` called at
 *       /root/reference/generate_illusion.py:549-550 and /root/reference/fitness_calculator.py:498.
 *
 * PARITY UNPINNED: both algorithms live in un-vendored submodules that are EMPTY directories in
 * the reference checkout (chainer_prednet/, optical_flow/; .gitmodules:1-6, pins unknown) on top of
 * chainer and OpenCV, neither installed here.  The reference holds no test or golden vector for
 * them.  What is restated is the PUBLISHED algorithm:
 *   - PredNet: quadjr/PredNet -> LanaSina/chainer_prednet `PredNet/net.py` (PredNet.__call__,
 *     ConvLSTM.__call__, EltFilter) and `PredNet/call_prednet.py` (read_image: uint8/255 float32;
 *     write_image: (P0*255).astype(uint8); extension frames feed the float prediction back).
 *   - LK: OpenCV 4.x imgproc/video: cvtColor BGR2GRAY (8u, 15-bit fixed point), pyrDown (8u),
 *     cornerMinEigenVal + goodFeaturesToTrack, calcSharrDeriv, LKTrackerInvoker
 *     (calcOpticalFlowPyrLK), with the OpenCV tutorial parameters used by Optical_Flow_Analyzer.
 *
 * Floating-point evaluation ORDER is not pinned by the reference (chainer = im2col + BLAS sgemm,
 * cupy = cuDNN, OpenCV = SIMD dependent).  This oracle fixes ONE order, documented in DESIGN.md
 * ("canonical arithmetic"), chosen so that a HIP implementation can reproduce it bit-for-bit:
 *   - every 3x3 convolution output is ONE fp32 fused-multiply-add chain, acc=0, over
 *     k = (source, input channel c, ky, kx) in that nesting order (= OIHW flattening c*9+ky*3+kx),
 *     zero padding included as explicit 0*w terms; bias / peephole are added after the chain;
 *   - a source that is nearest-neighbour unpooled x2 before its 3x3 convolution (R_{l+1} inside
 *     ConvLSTM_l) is a chain of its own, started from 0 and added to the chain of the
 *     full-resolution sources with ONE fp32 addition (chainer's ConvLSTM likewise adds the outputs
 *     of separate convolutions), and it is evaluated in the algebraically identical 2x2
 *     form: the 3x3 window of output pixel (y, x) covers only 2x2 distinct pixels of the
 *     half-resolution source, which ones depends on the parity class (y&1, x&1), so its nine
 *     weights are summed per distinct source pixel beforehand (presum_up_weights below: fp32
 *     additions in (ky, kx) order) and the chain runs over k = (channel c, a, b), a, b in {0, 1},
 *     4 terms per channel instead of 9.  Same function as unpool + conv3x3, 2.25x fewer
 *     multiply-adds; the summation order differs from the 9-tap form like any other order would.
 *   - sigmoid / tanh are the fixed polynomial kernels below (Cephes-style, explicit fmaf);
 *   - LK window sums are exact integers (int64); the 2x2 solve is fp32 with no contraction.
 * Compile with -ffp-contract=off (see oracle/Makefile).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EIG_MAX_LAYERS 8

/* ----------------------------------------------------------------------------------------------
 * Deterministic fp32 transcendental kernels (canonical arithmetic).
 * Only IEEE basic operations with explicit fmaf -> bit-identical on any IEEE machine.
 * ---------------------------------------------------------------------------------------------- */
static inline float det_expf(float x)
{
    if (x > 80.0f) x = 80.0f;
    if (x < -80.0f) x = -80.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int)n + 127) << 23;
    return y * s.f;
}

/* EIG_GATE_ORDER: element-wise order of the canonical ConvLSTM gate epilogue (the HIP side's conv_mfma.h has the same switch and the
 * same default).  1 (round 4): what chainer_prednet's ConvLSTM.__call__ does wherever that is knowable -- the peephole c_g(c) = W * c
 * as a rounded product added last, F.sigmoid = tanh(x * 0.5) * 0.5 + 0.5 (chainer/functions/activation/sigmoid.py forward_cpu; on
 * the deterministic tanh below instead of the host's libm), cc = tanh(cc) * ii; cc += ff * c as two rounded products and one
 * addition.  0: rounds 1-3 (fmaf peepholes / cell update, 1 / (1 + exp(-x))), kept for A/B builds. */
#ifndef EIG_GATE_ORDER
#define EIG_GATE_ORDER 1
#endif
int eig_oracle_gate_order(void) { return EIG_GATE_ORDER; }
/* Threads of the OpenMP loops (oracle/__init__.py caps them: one item per (channel, row) does not feed 256 threads well). */
static int g_threads = 0;  /* 0: the OpenMP default.  A num_threads clause of THIS library's loops only: omp_set_num_threads would also
                              re-size the pools of every other OpenMP user of the process (torch's CPU convolutions in bench.py's CPU leg) */
void eig_oracle_set_threads(int n) { g_threads = n > 0 ? n : 0; }
int eig_oracle_get_threads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }
#define EIG_NT (g_threads > 0 ? g_threads : omp_get_max_threads())
static inline float det_tanhf(float x);
#if EIG_GATE_ORDER
static inline float det_sigmoidf(float x) { return det_tanhf(x * 0.5f) * 0.5f + 0.5f; }
#else
static inline float det_sigmoidf(float x) { return 1.0f / (1.0f + det_expf(-x)); }
#endif

static inline float det_tanhf(float x)
{
    float ax = fabsf(x);
    if (ax < 0.625f) {
        float z = x * x;
        float p = -5.70498872745e-3f;
        p = fmaf(p, z, 2.06390887954e-2f);
        p = fmaf(p, z, -5.37397155531e-2f);
        p = fmaf(p, z, 1.33314422036e-1f);
        p = fmaf(p, z, -3.33332819422e-1f);
        float pz = p * z;
        return fmaf(pz, x, x);
    }
    if (ax > 10.0f) ax = 10.0f;
    float t = det_expf(2.0f * ax);
    float r = 1.0f - 2.0f / (t + 1.0f);
    return x < 0.0f ? -r : r;
}

/* exported for unit tests of the math kernels */
void eig_oracle_det_math(const float* x, int n, float* out_exp, float* out_sig, float* out_tanh)
{
    for (int i = 0; i < n; i++) {
        if (out_exp) out_exp[i] = det_expf(x[i]);
        if (out_sig) out_sig[i] = det_sigmoidf(x[i]);
        if (out_tanh) out_tanh[i] = det_tanhf(x[i]);
    }
}

/* ----------------------------------------------------------------------------------------------
 * PredNet
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int L;
    int ch[EIG_MAX_LAYERS];
    int W[EIG_MAX_LAYERS], H[EIG_MAX_LAYERS];
    const float* convA_w[EIG_MAX_LAYERS]; /* [C_l][2C_{l-1}][3][3], l>=1 */
    const float* convA_b[EIG_MAX_LAYERS];
    const float* convP_w[EIG_MAX_LAYERS]; /* [C_l][C_l][3][3] */
    const float* convP_b[EIG_MAX_LAYERS];
    const float* wx0[EIG_MAX_LAYERS][4];  /* x_{g}0: [C_l][2C_l][3][3]   (E_l source)        */
    const float* wx1[EIG_MAX_LAYERS][4];  /* x_{g}1: [C_l][C_{l+1}][3][3] (unpooled R_{l+1}) */
    const float* wh[EIG_MAX_LAYERS][4];   /* h_{g}:  [C_l][C_l][3][3] + bias                  */
    const float* bh[EIG_MAX_LAYERS][4];
    const float* peep[EIG_MAX_LAYERS][3]; /* c_i, c_f, c_o: [C_l][H_l][W_l] */
    /* state */
    float* h[EIG_MAX_LAYERS];
    float* hn[EIG_MAX_LAYERS];
    float* c[EIG_MAX_LAYERS];
    float* P[EIG_MAX_LAYERS];
    float* E[EIG_MAX_LAYERS];
    float* pad;  /* zero-padded source planes scratch */
    float* gate; /* 4 gate pre-activations scratch */
    float* up;   /* chain of the unpooled source scratch */
    float* tmp;  /* ConvA full-resolution scratch */
    int order;   /* 0: the build's canonical arithmetic; 1: the reference's element-wise order (lstm_reference_order) */
    int wino_mask; /* canonical order only: which operators run as Winograd F(4x4, 3x3) (eig_wino_op) */
} prednet_t;

/* Tensor table order shared with the Python wrapper (oracle/__init__.py: tensor_table()). */
static int bind_tensors(prednet_t* n, const float* const* t)
{
    int k = 0;
    for (int l = 0; l < n->L; l++) {
        if (l > 0) { n->convA_w[l] = t[k++]; n->convA_b[l] = t[k++]; }
        n->convP_w[l] = t[k++]; n->convP_b[l] = t[k++];
        for (int g = 0; g < 4; g++) {
            n->wx0[l][g] = t[k++];
            n->wx1[l][g] = (l < n->L - 1) ? t[k++] : NULL;
            n->wh[l][g] = t[k++];
            n->bh[l][g] = t[k++];
        }
        for (int g = 0; g < 3; g++) n->peep[l][g] = t[k++];
    }
    return k;
}

/* Copy `C` planes of src (Hs x Ws each) into zero-padded planes ((H+2) x (W+2)); when up==1 the
 * source is at half resolution and is nearest-neighbour unpooled x2 on the fly
 * (chainer F.unpooling_2d(R, 2, stride=2, cover_all=False)). */
static void fill_padded(float* pad, const float* src, int C, int H, int W, int up)
{
    const int PW = W + 2, PH = H + 2;
    memset(pad, 0, sizeof(float) * (size_t)C * PW * PH);
    for (int c = 0; c < C; c++) {
        float* pp = pad + (size_t)c * PW * PH;
        if (!up) {
            const float* sp = src + (size_t)c * H * W;
            for (int y = 0; y < H; y++) memcpy(pp + (size_t)(y + 1) * PW + 1, sp + (size_t)y * W, sizeof(float) * W);
        } else {
            const int Hs = H / 2, Ws = W / 2;
            const float* sp = src + (size_t)c * Hs * Ws;
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) pp[(size_t)(y + 1) * PW + 1 + x] = sp[(size_t)(y / 2) * Ws + x / 2];
        }
    }
}

/* acc[o][y][x] = fmaf-chain over (c, ky, kx) of pad[c][y+ky][x+kx] * w[o][c][ky][kx], continuing
 * from the values already in acc (so several sources chain into one accumulator). */
#define XB 32
static void conv3x3_chain(float* acc, const float* pad, const float* w, int Cout, int Cin, int H, int W)
{
    const int PW = W + 2;
    const size_t PP = (size_t)PW * (H + 2);
    /* one work item per (output channel, row): every output pixel is its own chain, so the split changes no bit (a 3-channel
     * layer keeps a many-core host busy too; tests/test_oracle_algorithms.py compares a 1-thread run) */
#pragma omp parallel for schedule(static) num_threads(EIG_NT)
    for (int oy = 0; oy < Cout * H; oy++) {
        const int o = oy / H, y = oy - o * H;
        const float* wo = w + (size_t)o * Cin * 9;
        float* ao = acc + (size_t)o * H * W;
        {
            for (int x0 = 0; x0 < W; x0 += XB) {
                const int nb = (W - x0 < XB) ? (W - x0) : XB;
                float a[XB];
                for (int i = 0; i < nb; i++) a[i] = ao[(size_t)y * W + x0 + i];
                if (nb == XB) {
                    for (int c = 0; c < Cin; c++) {
                        const float* pc = pad + (size_t)c * PP + (size_t)y * PW + x0;
                        const float* wc = wo + c * 9;
                        for (int ky = 0; ky < 3; ky++) {
                            const float* row = pc + (size_t)ky * PW;
                            for (int kx = 0; kx < 3; kx++) {
                                const float wv = wc[ky * 3 + kx];
                                for (int i = 0; i < XB; i++) a[i] = fmaf(row[i + kx], wv, a[i]);
                            }
                        }
                    }
                } else {
                    for (int c = 0; c < Cin; c++) {
                        const float* pc = pad + (size_t)c * PP + (size_t)y * PW + x0;
                        const float* wc = wo + c * 9;
                        for (int ky = 0; ky < 3; ky++) {
                            const float* row = pc + (size_t)ky * PW;
                            for (int kx = 0; kx < 3; kx++) {
                                const float wv = wc[ky * 3 + kx];
                                for (int i = 0; i < nb; i++) a[i] = fmaf(row[i + kx], wv, a[i]);
                            }
                        }
                    }
                }
                for (int i = 0; i < nb; i++) ao[(size_t)y * W + x0 + i] = a[i];
            }
        }
    }
}

/* Weights of the 2x2 form of `unpool x2 -> conv3x3` for parity class (py, px) = (y&1, x&1) of the
 * output pixel.  Output row y = 2Y+py reads unpooled rows y-1, y, y+1 = source rows Y-1, Y, Y (py=0)
 * or Y, Y, Y+1 (py=1): tap a in {0,1} stands for source row Y+a-1+py and collects ky in
 * {0} / {1,2} (py=0) or {0,1} / {2} (py=1); columns likewise.  The collected weights are added in
 * fp32 in (ky, kx) row-major order, starting from the first one.   w9: [3][3] -> w4: [2][2] */
static void presum_up_weights(const float* w9, int py, int px, float* w4)
{
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            const int ky0 = py ? (a ? 2 : 0) : (a ? 1 : 0), ky1 = py ? (a ? 2 : 1) : (a ? 2 : 0);
            const int kx0 = px ? (b ? 2 : 0) : (b ? 1 : 0), kx1 = px ? (b ? 2 : 1) : (b ? 2 : 0);
            float s = 0.0f;
            int first = 1;
            for (int ky = ky0; ky <= ky1; ky++)
                for (int kx = kx0; kx <= kx1; kx++) {
                    if (first) { s = w9[ky * 3 + kx]; first = 0; }
                    else s = s + w9[ky * 3 + kx];
                }
            w4[a * 2 + b] = s;
        }
}

/* acc[o][y][x] = fmaf-chain over (c, a, b) of src[c][Y+a-1+py][X+b-1+px] * w4_{py,px}[o][c][a][b] (zero outside the
 * source), continuing from acc: `unpool x2 -> conv3x3` of a half-resolution source [Cin][H/2][W/2] in its 2x2 form. */
static void conv_up2x2_chain(float* acc, const float* src, const float* w, int Cout, int Cin, int H, int W)
{
    const int Hs = H / 2, Ws = W / 2;
    float* w4 = (float*)malloc(sizeof(float) * (size_t)Cout * Cin * 16); /* [o][c][class][a][b] */
    for (size_t oc = 0; oc < (size_t)Cout * Cin; oc++)
        for (int cls = 0; cls < 4; cls++) presum_up_weights(w + oc * 9, cls >> 1, cls & 1, w4 + oc * 16 + cls * 4);
#pragma omp parallel for schedule(static) num_threads(EIG_NT)
    for (int oy = 0; oy < Cout * H; oy++) {
        const int o = oy / H, y = oy - o * H;
        float* ao = acc + (size_t)o * H * W;
        {
            const int py = y & 1, Y = y >> 1;
            for (int x = 0; x < W; x++) {
                const int px = x & 1, X = x >> 1;
                const int cls = py * 2 + px;
                float a_ = ao[(size_t)y * W + x];
                for (int c = 0; c < Cin; c++) {
                    const float* sc = src + (size_t)c * Hs * Ws;
                    const float* wc = w4 + ((size_t)o * Cin + c) * 16 + cls * 4;
                    for (int a = 0; a < 2; a++) {
                        const int sy = Y + a - 1 + py;
                        for (int b = 0; b < 2; b++) {
                            const int sx = X + b - 1 + px;
                            const float v = (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) ? sc[(size_t)sy * Ws + sx] : 0.0f;
                            a_ = fmaf(v, wc[a * 2 + b], a_);
                        }
                    }
                }
                ao[(size_t)y * W + x] = a_;
            }
        }
    }
    free(w4);
}

/* ---- Winograd F(2x2, 3x3) form of a 3x3 'same' convolution, fp32, ONE fixed order of operations.  Rounds 4-5 shipped a HIP kernel that ran exactly these; round 6
 * removed it (nothing ran it once F(4x4) was the default) and NO operator of the roll-out below takes this form any more -- it stays reachable through
 * eig_oracle_wino_chain_m(m = 2) for tests/studies/winograd_study.py and the "same convolution" test.  Per 2x2 output tile T = (ty, tx), input patch d[4][4] =
 * in[c][2ty-1+i][2tx-1+j] (zeros outside the image):
 *   input transform   t_ij = rows:  t0j = d0j - d2j, t1j = d1j + d2j, t2j = d2j - d1j, t3j = d1j - d3j   (B^T d)
 *                     V_ij = cols:  Vi0 = ti0 - ti2, Vi1 = ti1 + ti2, Vi2 = ti2 - ti1, Vi3 = ti1 - ti3   ((B^T d) B)
 *   weight transform  s_ij = rows:  s0j = g0j, s1j = ((g0j + g1j) + g2j) * 0.5, s2j = ((g0j - g1j) + g2j) * 0.5, s3j = g2j   (G g)
 *                     U_ij = cols:  Ui0 = si0, Ui1 = ((si0 + si1) + si2) * 0.5, Ui2 = ((si0 - si1) + si2) * 0.5, Ui3 = si2   ((G g) G^T)
 *   16 independent chains  M_ij[o][T] = fmaf(V_ij[c][T], U_ij[o][c], M_ij[o][T]) over the sources in list order, channels ascending
 *   output transform  c_i0 = (M_i0 + M_i1) + M_i2,  c_i1 = (M_i1 - M_i2) - M_i3                       (M A)
 *                     y_0b = (c_0b + c_1b) + c_2b,  y_1b = c_1b - (c_2b + c_3b)                        (A^T (M A))
 * 16 multiply-adds per channel and 2x2 outputs instead of 36.  Algebraically the same convolution; numerically one more summation
 * order (profiles/r04_c_winograd_study.json: indistinguishable from any other fp32 re-order of the reference's arithmetic). */
static void wino_weights(const float* g /*[3][3]*/, float* U /*[16]*/)
{
    float s[4][3];
    for (int j = 0; j < 3; j++) {
        s[0][j] = g[j];
        s[1][j] = ((g[j] + g[3 + j]) + g[6 + j]) * 0.5f;
        s[2][j] = ((g[j] - g[3 + j]) + g[6 + j]) * 0.5f;
        s[3][j] = g[6 + j];
    }
    for (int i = 0; i < 4; i++) {
        U[i * 4 + 0] = s[i][0];
        U[i * 4 + 1] = ((s[i][0] + s[i][1]) + s[i][2]) * 0.5f;
        U[i * 4 + 2] = ((s[i][0] - s[i][1]) + s[i][2]) * 0.5f;
        U[i * 4 + 3] = s[i][2];
    }
}

/* ---- Winograd F(4x4, 3x3) (round 5; csrc/conv_wino4.h runs exactly these operations): 36 multiply-adds per channel and 4x4 outputs instead of the 64 of
 * F(2x2) (144 direct).  Interpolation points 0, +-1, +-2, inf (Lavin & Gray); fp32, ONE fixed order.  Whether the larger tile costs parity was measured BEFORE it
 * was built (tests/studies/winograd_study.py --large-tiles, profiles/r05_c_winograd_large_tiles_study.json): on the operators that take it (layers >= 1) F(4x4) is
 * indistinguishable from F(2x2) -- the byte flips against the reference-order implementations are decided in the image layer, which stays direct.
 * 1-D transforms, applied to rows first, then to columns (fmaf = ONE rounding; this file is compiled with -ffp-contract=off):
 *   input  (B^T, six values d0..d5 -> six):   T0 = fmaf(4, d0, fmaf(-5, d2, d4));   p = fmaf(-4, d2, d4), q = fmaf(-4, d1, d3):  T1 = p + q, T2 = p - q;
 *                                             r = d4 - d2, s = d3 - d1:  T3 = fmaf(2, s, r), T4 = fmaf(-2, s, r);   T5 = fmaf(4, d1, fmaf(-5, d3, d5))
 *   weight (G, three values g0..g2 -> six):   W0 = 0.25 g0;   a = g0 + g2:  W1 = (a + g1) * (-1/6), W2 = (a - g1) * (-1/6);
 *                                             b = fmaf(4, g2, g0):  W3 = fmaf(2, g1, b) * (1/24), W4 = fmaf(-2, g1, b) * (1/24);   W5 = g2
 *   output (A^T, six values m0..m5 -> four):  s = m1 + m2, d = m1 - m2, u = m3 + m4, w = m3 - m4:
 *                                             Y0 = (m0 + s) + u, Y1 = fmaf(2, w, d), Y2 = fmaf(4, u, s), Y3 = fmaf(8, w, d) + m5
 * 36 independent chains M_ij[o][T] = fmaf(V_ij[c][T], U_ij[o][c], M_ij[o][T]) over the sources in list order, channels ascending.  The patch of a x2
 * nearest-unpooled map has rows / columns (a, b, b, c, c, d): p and q above are then the SAME operation on the SAME operands, T2 = p - q is an exact zero, so the
 * positions with xi = 2 or nu = 2 are chains of exact zeros (the HIP kernel skips them): 25 of 36. */
#define WINO4_C6 (-1.0f / 6.0f)
#define WINO4_C24 (1.0f / 24.0f)
static inline void wino4_in1d(const float* d, int st, float* T)   /* six values d[0], d[st], .. -> T[0..5] */
{
    const float d0 = d[0], d1 = d[st], d2 = d[2 * st], d3 = d[3 * st], d4 = d[4 * st], d5 = d[5 * st];
    T[0] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
    const float p = fmaf(-4.0f, d2, d4), q = fmaf(-4.0f, d1, d3);
    T[1] = p + q; T[2] = p - q;
    const float r = d4 - d2, s = d3 - d1;
    T[3] = fmaf(2.0f, s, r); T[4] = fmaf(-2.0f, s, r);
    T[5] = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
}
static inline void wino4_w1d(float g0, float g1, float g2, float* W)
{
    W[0] = 0.25f * g0;
    const float a = g0 + g2;
    W[1] = (a + g1) * WINO4_C6; W[2] = (a - g1) * WINO4_C6;
    const float b = fmaf(4.0f, g2, g0);
    W[3] = fmaf(2.0f, g1, b) * WINO4_C24; W[4] = fmaf(-2.0f, g1, b) * WINO4_C24;
    W[5] = g2;
}
static inline void wino4_out1d(const float* m, int st, float* Y)  /* six values -> Y[0..3] */
{
    const float m0 = m[0], m1 = m[st], m2 = m[2 * st], m3 = m[3 * st], m4 = m[4 * st], m5 = m[5 * st];
    const float s = m1 + m2, d = m1 - m2, u = m3 + m4, w = m3 - m4;
    Y[0] = (m0 + s) + u; Y[1] = fmaf(2.0f, w, d); Y[2] = fmaf(4.0f, u, s); Y[3] = fmaf(8.0f, w, d) + m5;
}
static void wino4_weights(const float* g /*[3][3]*/, float* U /*[36]*/)
{
    float s[6][3], W[6];
    for (int j = 0; j < 3; j++) { wino4_w1d(g[j], g[3 + j], g[6 + j], W); for (int i = 0; i < 6; i++) s[i][j] = W[i]; }   /* G g: down the columns */
    for (int i = 0; i < 6; i++) wino4_w1d(s[i][0], s[i][1], s[i][2], U + i * 6);                                            /* (G g) G^T: along the rows */
}
void eig_oracle_wino4_weights(const float* g, float* U) { wino4_weights(g, U); }   /* (tests: the engine's host code packs the same values) */

/* tile geometry of the two forms: m = 2 or 4 output pixels per tile side, a = m + 2 patch side, a * a positions */
#define WINO_NPOS(m) (((m) + 2) * ((m) + 2))
static int wino_th(int H, int m) { return (H + m - 1) / m; }
static int wino_tw(int W, int m) { return (W + m - 1) / m; }

/* M[npos][Cout][TH*TW] += chains over the Cin channels of one source (pad: [Cin][H+2][W+2], zero border).  A map whose height is not a multiple of m (a
 * top-layer map, e.g. 20 x 15): the rows below the map read as zeros and the outputs below it are dropped. */
static float* wino_input(const float* pad, int Cin, int H, int W, int m)   /* -> V[Cin][npos][TH*TW] (malloc'ed): B^T d B of every tile of every channel */
{
    const int TH = wino_th(H, m), TW = wino_tw(W, m), NT = TH * TW, PW = W + 2, PH = H + 2, np = WINO_NPOS(m);
    const size_t PP = (size_t)PW * PH;
    float* V = (float*)malloc(sizeof(float) * (size_t)Cin * np * NT);
#pragma omp parallel for schedule(static) num_threads(EIG_NT)
    for (int c = 0; c < Cin; c++) {
        const float* pc = pad + (size_t)c * PP;
        float* vc = V + (size_t)c * np * NT;
        for (int ty = 0; ty < TH; ty++)
            for (int tx = 0; tx < TW; tx++) {
                const int T = ty * TW + tx;
                if (m == 2) {
                    float d[4][4], t[4][4];
                    for (int i = 0; i < 4; i++)
                        for (int j = 0; j < 4; j++) d[i][j] = (2 * ty + i < PH) ? pc[(size_t)(2 * ty + i) * PW + 2 * tx + j] : 0.0f;  /* pad offset +1 absorbs the -1 */
                    for (int j = 0; j < 4; j++) {
                        t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
                    }
                    for (int i = 0; i < 4; i++) {
                        vc[(size_t)(i * 4 + 0) * NT + T] = t[i][0] - t[i][2];
                        vc[(size_t)(i * 4 + 1) * NT + T] = t[i][1] + t[i][2];
                        vc[(size_t)(i * 4 + 2) * NT + T] = t[i][2] - t[i][1];
                        vc[(size_t)(i * 4 + 3) * NT + T] = t[i][1] - t[i][3];
                    }
                } else {
                    float d[6][6], t[6][6], o6[6];
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 6; j++) d[i][j] = (4 * ty + i < PH && 4 * tx + j < PW) ? pc[(size_t)(4 * ty + i) * PW + 4 * tx + j] : 0.0f;
                    for (int j = 0; j < 6; j++) { wino4_in1d(&d[0][j], 6, o6); for (int i = 0; i < 6; i++) t[i][j] = o6[i]; }   /* rows: B^T d */
                    for (int i = 0; i < 6; i++) { wino4_in1d(&t[i][0], 1, o6); for (int j = 0; j < 6; j++) vc[(size_t)(i * 6 + j) * NT + T] = o6[j]; }   /* columns */
                }
            }
    }
    return V;
}
/* M[npos][Cout][TH*TW]: the chains continue over the Cin channels whose transformed tiles are in V */
static void wino_chains(float* M, const float* V, const float* w, int Cout, int Cin, int H, int W, int m)
{
    const int NT = wino_th(H, m) * wino_tw(W, m), np = WINO_NPOS(m);
    float* U = (float*)malloc(sizeof(float) * (size_t)Cout * Cin * np);
    for (size_t oc = 0; oc < (size_t)Cout * Cin; oc++) { if (m == 2) wino_weights(w + oc * 9, U + oc * np); else wino4_weights(w + oc * 9, U + oc * np); }
#pragma omp parallel for schedule(static) num_threads(EIG_NT)
    for (int op = 0; op < Cout * np; op++) {
        const int o = op / np, pos = op - o * np;
        float* mm = M + ((size_t)pos * Cout + o) * NT;
        for (int c = 0; c < Cin; c++) {
            const float u = U[((size_t)o * Cin + c) * np + pos];
            const float* v = V + ((size_t)c * np + pos) * NT;
            for (int T = 0; T < NT; T++) mm[T] = fmaf(v[T], u, mm[T]);
        }
    }
    free(U);
}
static void wino_accumulate(float* M, const float* pad, const float* w, int Cout, int Cin, int H, int W, int m)
{
    float* V = wino_input(pad, Cin, H, W, m);
    wino_chains(M, V, w, Cout, Cin, H, W, m);
    free(V);
}

/* out[o][m ty + a][m tx + b] = y_ab of the output transform (overwrites out) */
static void wino_finish(float* out, const float* M, int Cout, int H, int W, int m)
{
    const int TH = wino_th(H, m), TW = wino_tw(W, m), NT = TH * TW;
#pragma omp parallel for schedule(static) num_threads(EIG_NT)
    for (int o = 0; o < Cout; o++)
        for (int ty = 0; ty < TH; ty++)
            for (int tx = 0; tx < TW; tx++) {
                const int T = ty * TW + tx;
                float* po = out + (size_t)o * H * W;
                if (m == 2) {
                    float mv[16], c[4][2];
                    for (int p = 0; p < 16; p++) mv[p] = M[((size_t)p * Cout + o) * NT + T];
                    for (int i = 0; i < 4; i++) {
                        c[i][0] = (mv[i * 4 + 0] + mv[i * 4 + 1]) + mv[i * 4 + 2];
                        c[i][1] = (mv[i * 4 + 1] - mv[i * 4 + 2]) - mv[i * 4 + 3];
                    }
                    for (int b = 0; b < 2; b++) {
                        po[(size_t)(2 * ty) * W + 2 * tx + b] = (c[0][b] + c[1][b]) + c[2][b];
                        if (2 * ty + 1 < H) po[(size_t)(2 * ty + 1) * W + 2 * tx + b] = c[1][b] - (c[2][b] + c[3][b]);
                    }
                } else {
                    float mv[36], c[6][4], y4[4];
                    for (int p = 0; p < 36; p++) mv[p] = M[((size_t)p * Cout + o) * NT + T];
                    for (int i = 0; i < 6; i++) wino4_out1d(mv + i * 6, 1, c[i]);      /* columns in each row xi: c_xi,b */
                    for (int b = 0; b < 4; b++) {
                        wino4_out1d(&c[0][b], 4, y4);                                  /* rows: y_a,b */
                        for (int aa = 0; aa < 4; aa++)
                            if (4 * ty + aa < H && 4 * tx + b < W) po[(size_t)(4 * ty + aa) * W + 4 * tx + b] = y4[aa];
                    }
                }
            }
}
static size_t wino_m_floats(int Cout, int H, int W, int m) { return (size_t)WINO_NPOS(m) * Cout * wino_th(H, m) * wino_tw(W, m); }
#define EIG_WINO_M 4   /* tile size of every Winograd operator of the roll-out: F(4x4, 3x3) (csrc/conv_wino4.h) */

/* Which operators take the Winograd form (the HIP engine applies the same rule, eigen_engine.hip: wino_op).  wino_mask: bit l =
 * ConvLSTM_l, bit 8 + l = ConvA_l, bit 16 + l = ConvP_l, AND the class bit 25 / 26 / 27 of the ConvLSTMs / ConvAs / ConvPs (rounds 4-5: the class bit chose F(4x4) over
 * F(2x2); with the F(2x2) kernel gone an operator whose class bit is clear is a direct one).  kind 0 ConvLSTM_l, 1 ConvA_l, 2 ConvP_l; Cin: every full-resolution source
 * has a multiple of 8 channels; Cout (per gate): 16-channel groups, and for the plain convolutions N-blocks of 48 or 64 columns
 * without padding; rows of 16-byte chunks; odd H only for an operator of the TOP layer.  A property of the operator's shape only,
 * never of the batch: results must not depend on how a population is split into device batches. */
static int eig_wino_op(int wino_mask, int kind, int l, int Cin, int Cout, int H, int W, int top)
{
    if (!((wino_mask >> (8 * kind + l)) & 1) || !((wino_mask >> (25 + kind)) & 1) || l < 1) return 0;
    if ((Cin % 8) || (Cout % 16) || (W % 4)) return 0;
    if ((H % 2) && !(top && kind != 1)) return 0;
    if (kind != 0 && (Cout % 48) && (Cout % 64)) return 0;
    return 1;
}

/* Bit 24 of wino_mask: a ConvLSTM in Winograd form takes its unpooled source R_{l+1} INTO the same sixteen chains, between E_l and h_l --
 * x_g0(E) + x_g1(unpool R) + h_g(h), chainer's own left-to-right order -- as the plain Winograd convolution of the unpooled
 * (nearest x2) map, instead of adding a separate 2x2-form chain afterwards.  The 4x4 patch of an unpooled map has only 3x3 distinct
 * values (rows s_-1, s_0, s_0, s_+1), so t_2 = d_2 - d_1 = 0: the positions with xi = 2 or nu = 2 are chains of exact zeros (the HIP
 * kernel skips them; fma(0, u, M) = M and c + 0 = c exactly) -- 9 multiply-adds per channel and tile instead of the 2x2 form's 16.
 * Needs 16-byte rows at the source resolution (W % 8 == 0) and a multiple of 8 source channels; same rule in eigen_engine.hip. */
static int eig_wino_fuse_up(int wino_mask, int l, int L, int W, int Cup) { return ((wino_mask >> 24) & 1) && l < L - 1 && (W % 8) == 0 && (Cup % 8) == 0; }

/* A ConvLSTM takes the Winograd form when its operator shape is eligible AND -- below the top layer -- its unpooled source can ride in the same chains (round 6: the
 * form "Winograd chains + a separate 2x2-form chain" and the eight-wave kernel that ran it are gone; such an operator is a direct one, in the engine and here). */
static int eig_wino_lstm(int wino_mask, int l, int L, int C, int H, int W, int Cup)
{
    return eig_wino_op(wino_mask, 0, l, C, C, H, W, l == L - 1) && (l == L - 1 || eig_wino_fuse_up(wino_mask, l, L, W, Cup));
}

/* exported for kernel-level tests: out[Cout][H][W] = Winograd chain over the listed full-resolution sources (canonical order) */
int eig_oracle_wino_chain_m(int ns, const float* const* src, const int* cin, const float* const* w, int Cout, int H, int W, float* out, int m)
{
    if ((W & 1) || (m != 2 && m != 4) || (m == 4 && (W & 3))) return -1;
    size_t maxc = 0;
    for (int s = 0; s < ns; s++) if ((size_t)cin[s] > maxc) maxc = (size_t)cin[s];
    float* pad = (float*)malloc(sizeof(float) * maxc * (H + 2) * (W + 2));
    float* M = (float*)calloc(wino_m_floats(Cout, H, W, m), sizeof(float));
    for (int s = 0; s < ns; s++) {
        const int PW = W + 2;
        memset(pad, 0, sizeof(float) * (size_t)cin[s] * PW * (H + 2));
        for (int c = 0; c < cin[s]; c++)
            for (int y = 0; y < H; y++) memcpy(pad + ((size_t)c * (H + 2) + y + 1) * PW + 1, src[s] + ((size_t)c * H + y) * W, sizeof(float) * W);
        wino_accumulate(M, pad, w[s], Cout, cin[s], H, W, m);
    }
    wino_finish(out, M, Cout, H, W, m);
    free(pad); free(M);
    return 0;
}
int eig_oracle_wino_chain(int ns, const float* const* src, const int* cin, const float* const* w, int Cout, int H, int W, float* out)
{
    return eig_oracle_wino_chain_m(ns, src, cin, w, Cout, H, W, out, 2);
}

static inline float relu(float v) { return v > 0.0f ? v : 0.0f; }

/* E = concat(relu(A - P), relu(P - A))  -- net.py PredNet.__call__ */
static void error_unit(float* E, const float* A, const float* P, int C, int HW)
{
    for (size_t i = 0; i < (size_t)C * HW; i++) {
        E[i] = relu(A[i] - P[i]);
        E[(size_t)C * HW + i] = relu(P[i] - A[i]);
    }
}

static void prednet_alloc(prednet_t* n)
{
    size_t maxpad = 0, maxgate = 0, maxtmp = 0;
    for (int l = 0; l < n->L; l++) {
        const size_t hw = (size_t)n->H[l] * n->W[l], C = (size_t)n->ch[l];
        n->h[l] = (float*)calloc(C * hw, sizeof(float));
        n->hn[l] = (float*)calloc(C * hw, sizeof(float));
        n->c[l] = (float*)calloc(C * hw, sizeof(float));
        n->P[l] = (float*)calloc(C * hw, sizeof(float));
        n->E[l] = (float*)calloc(2 * C * hw, sizeof(float));
        const size_t phw = (size_t)(n->H[l] + 2) * (n->W[l] + 2);
        size_t cmax = 2 * C;
        if (l < n->L - 1 && (size_t)n->ch[l + 1] > cmax) cmax = (size_t)n->ch[l + 1];
        if (cmax * phw > maxpad) maxpad = cmax * phw;
        if (4 * C * hw > maxgate) maxgate = 4 * C * hw;
        if (l < n->L - 1 && (size_t)n->ch[l + 1] * hw > maxtmp) maxtmp = (size_t)n->ch[l + 1] * hw;
    }
    n->pad = (float*)malloc(sizeof(float) * maxpad);
    n->gate = (float*)malloc(sizeof(float) * maxgate);
    n->up = (float*)malloc(sizeof(float) * maxgate);
    n->tmp = (float*)malloc(sizeof(float) * (maxtmp ? maxtmp : 1));
}

static void prednet_free(prednet_t* n)
{
    for (int l = 0; l < n->L; l++) { free(n->h[l]); free(n->hn[l]); free(n->c[l]); free(n->P[l]); free(n->E[l]); }
    free(n->pad); free(n->gate); free(n->up); free(n->tmp);
}

static void prednet_reset(prednet_t* n)
{
    for (int l = 0; l < n->L; l++) {
        const size_t sz = (size_t)n->ch[l] * n->H[l] * n->W[l];
        memset(n->h[l], 0, sz * sizeof(float));
        memset(n->c[l], 0, sz * sizeof(float));
        memset(n->P[l], 0, sz * sizeof(float));
    }
}

/* ConvLSTM_l in the element-wise order the REFERENCE evaluates it, as far as that order is knowable without its BLAS
 * (chainer_prednet PredNet/net.py ConvLSTM.__call__, quadjr/PredNet lineage -- UPSTREAM-RECALL; the submodule is
 * absent from /root/reference, .gitmodules:1-3):
 *     ii = x_i0(E); ii += x_i1(unpooling_2d(R_{l+1})); ii += h_i(h); ii += c_i(c); ii = F.sigmoid(ii)      (ff likewise)
 *     cc = x_c0(E); cc += x_c1(upR); cc += h_c(h); cc = F.tanh(cc); cc *= ii; cc += ff * c
 *     oo = x_o0(E); oo += x_o1(upR); oo += h_o(h); oo += c_o(c)  [the OLD c]; oo = F.sigmoid(oo);  c = cc; h = oo * F.tanh(c)
 * i.e. every convolution is a tensor of its own started from 0 (x_* without bias; h_* = Convolution2D WITH bias, added to its own
 * output), the unpooled source goes through the PLAIN 9-tap convolution (no 2x2 form), the tensors are added left to right with
 * one fp32 rounding each, the peephole EltFilter is a rounded product added last, the cell update is two rounded products and
 * one addition (no fma; this file is compiled with -ffp-contract=off), and chainer's CPU F.sigmoid is tanh(x * 0.5) * 0.5 + 0.5
 * (chainer/functions/activation/sigmoid.py forward_cpu) on the host's libm tanh.  What stays unknowable: the summation order
 * INSIDE a convolution (chainer: im2col + the host's BLAS) -- the (c, ky, kx) fma chain is kept there.  Used by the tests as a
 * second, torch-free statement of the same order as oracle/prednet_torch.py order="chainer". */
static inline float ref_sigmoidf(float x) { return tanhf(x * 0.5f) * 0.5f + 0.5f; }
static void lstm_reference_order(prednet_t* n, int l)
{
    const int L = n->L, H = n->H[l], W = n->W[l], C = n->ch[l];
    const size_t hw = (size_t)H * W, chw = (size_t)C * hw;
    float* z = n->gate;                 /* [4][C][hw]: running sum of the tensors */
    float* t = n->up;                   /* [C][hw]: one convolution's output */
    for (int g = 0; g < 4; g++) {
        float* zg = z + (size_t)g * chw;
        memset(zg, 0, sizeof(float) * chw);
        fill_padded(n->pad, n->E[l], 2 * C, H, W, 0);
        conv3x3_chain(zg, n->pad, n->wx0[l][g], C, 2 * C, H, W);                       /* x_g0(E) */
        if (l < L - 1) {
            memset(t, 0, sizeof(float) * chw);
            fill_padded(n->pad, n->h[l + 1], n->ch[l + 1], H, W, 1);                    /* unpooling_2d(R_{l+1}, 2) */
            conv3x3_chain(t, n->pad, n->wx1[l][g], C, n->ch[l + 1], H, W);              /* x_g1(upR): plain 9 taps */
            for (size_t i = 0; i < chw; i++) zg[i] = zg[i] + t[i];
        }
        memset(t, 0, sizeof(float) * chw);
        fill_padded(n->pad, n->h[l], C, H, W, 0);
        conv3x3_chain(t, n->pad, n->wh[l][g], C, C, H, W);                              /* h_g(h) ... */
        for (int o = 0; o < C; o++) {
            const float b = n->bh[l][g][o];
            for (size_t p = 0; p < hw; p++) { const size_t i = (size_t)o * hw + p; const float hb = t[i] + b; zg[i] = zg[i] + hb; }  /* ... + its bias, then += */
        }
    }
    for (size_t i = 0; i < chw; i++) {
        const float cold = n->c[l][i];
        const float pi = n->peep[l][0][i] * cold, pf = n->peep[l][1][i] * cold, po = n->peep[l][2][i] * cold;  /* EltFilter: W * c */
        const float ii = ref_sigmoidf(z[i] + pi);
        const float ff = ref_sigmoidf(z[chw + i] + pf);
        float cc = tanhf(z[2 * chw + i]);
        cc = cc * ii;
        const float fc = ff * cold;
        cc = cc + fc;
        const float oo = ref_sigmoidf(z[3 * chw + i] + po);
        n->c[l][i] = cc;
        n->hn[l][i] = oo * tanhf(cc);
    }
    { float* sw = n->h[l]; n->h[l] = n->hn[l]; n->hn[l] = sw; }
}

/* One PredNet.__call__(x): x is [C0][H][W] float32.  Afterwards n->P[0] holds the prediction. */
static void prednet_step(prednet_t* n, const float* x)
{
    const int L = n->L;
    /* ---- bottom-up: error units ---- */
    error_unit(n->E[0], x, n->P[0], n->ch[0], n->H[0] * n->W[0]);
    for (int l = 1; l < L; l++) {
        const int Hi = n->H[l - 1], Wi = n->W[l - 1], Ci = 2 * n->ch[l - 1], Co = n->ch[l];
        const int Ho = n->H[l], Wo = n->W[l];
        fill_padded(n->pad, n->E[l - 1], Ci, Hi, Wi, 0);
        memset(n->tmp, 0, sizeof(float) * (size_t)Co * Hi * Wi);
        if (eig_wino_op(n->wino_mask, 1, l, n->ch[l - 1], Co, Hi, Wi, 0)) {  /* Winograd form: the 2x2 tile is the pooling window */
            const int wm = EIG_WINO_M;
            float* M = (float*)calloc(wino_m_floats(Co, Hi, Wi, wm), sizeof(float));
            wino_accumulate(M, n->pad, n->convA_w[l], Co, Ci, Hi, Wi, wm);
            wino_finish(n->tmp, M, Co, Hi, Wi, wm);
            free(M);
        } else
        conv3x3_chain(n->tmp, n->pad, n->convA_w[l], Co, Ci, Hi, Wi);
        /* A = max_pooling_2d(relu(conv + b), 2, stride=2); E_l = err(A, P_l) */
        float* E = n->E[l];
        const float* P = n->P[l];
        for (int o = 0; o < Co; o++) {
            const float b = n->convA_b[l][o];
            const float* t = n->tmp + (size_t)o * Hi * Wi;
            for (int y = 0; y < Ho; y++)
                for (int xx = 0; xx < Wo; xx++) {
                    float v00 = relu(t[(size_t)(2 * y) * Wi + 2 * xx] + b);
                    float v01 = relu(t[(size_t)(2 * y) * Wi + 2 * xx + 1] + b);
                    float v10 = relu(t[(size_t)(2 * y + 1) * Wi + 2 * xx] + b);
                    float v11 = relu(t[(size_t)(2 * y + 1) * Wi + 2 * xx + 1] + b);
                    float A = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
                    const size_t idx = ((size_t)o * Ho + y) * Wo + xx;
                    const float p = P[idx];
                    E[idx] = relu(A - p);
                    E[(size_t)Co * Ho * Wo + idx] = relu(p - A);
                }
        }
    }
    /* ---- top-down: ConvLSTM + prediction ---- */
    for (int l = L - 1; l >= 0; l--) {
        const int H = n->H[l], W = n->W[l], C = n->ch[l];
        const size_t hw = (size_t)H * W;
        if (n->order == 1) { lstm_reference_order(n, l); goto predict; }
        memset(n->gate, 0, sizeof(float) * 4 * C * hw);
        /* one chain over the full-resolution sources E_l, h_l (ConvLSTM.__call__: x_*0, h_*) ... */
        if (eig_wino_lstm(n->wino_mask, l, L, C, H, W, l < L - 1 ? n->ch[l + 1] : 0)) {  /* ... in its Winograd form: (m + 2)^2 chains per m x m tile */
            const int fuse = l < L - 1;   /* (below the top layer the Winograd form exists only with the unpooled source inside the chains: eig_wino_lstm) */
            const int wm = EIG_WINO_M;
            const size_t mf = wino_m_floats(C, H, W, wm);
            float* M = (float*)calloc(4 * mf, sizeof(float));   /* the chains of each of the four gates */
            float* V;
            fill_padded(n->pad, n->E[l], 2 * C, H, W, 0);
            V = wino_input(n->pad, 2 * C, H, W, wm);
            for (int g = 0; g < 4; g++) wino_chains(M + g * mf, V, n->wx0[l][g], C, 2 * C, H, W, wm);
            free(V);
            if (fuse) {   /* x_g1(unpooling_2d(R_{l+1})) in the same chains */
                fill_padded(n->pad, n->h[l + 1], n->ch[l + 1], H, W, 1);
                V = wino_input(n->pad, n->ch[l + 1], H, W, wm);
                for (int g = 0; g < 4; g++) wino_chains(M + g * mf, V, n->wx1[l][g], C, n->ch[l + 1], H, W, wm);
                free(V);
            }
            fill_padded(n->pad, n->h[l], C, H, W, 0);
            V = wino_input(n->pad, C, H, W, wm);
            for (int g = 0; g < 4; g++) wino_chains(M + g * mf, V, n->wh[l][g], C, C, H, W, wm);
            free(V);
            for (int g = 0; g < 4; g++) wino_finish(n->gate + (size_t)g * C * hw, M + g * mf, C, H, W, wm);
            free(M);
        } else {
        fill_padded(n->pad, n->E[l], 2 * C, H, W, 0);
        for (int g = 0; g < 4; g++) conv3x3_chain(n->gate + (size_t)g * C * hw, n->pad, n->wx0[l][g], C, 2 * C, H, W);
        fill_padded(n->pad, n->h[l], C, H, W, 0);
        for (int g = 0; g < 4; g++) conv3x3_chain(n->gate + (size_t)g * C * hw, n->pad, n->wh[l][g], C, C, H, W);
        }
        /* ... plus the chain of the unpooled R_{l+1} (x_*1; 2x2 form, see the header): one fp32 addition */
        if (l < L - 1 && !eig_wino_lstm(n->wino_mask, l, L, C, H, W, n->ch[l + 1])) { /* (direct form only) h[l+1] already holds R_{l+1} of this step */
            memset(n->up, 0, sizeof(float) * 4 * C * hw);
            for (int g = 0; g < 4; g++) conv_up2x2_chain(n->up + (size_t)g * C * hw, n->h[l + 1], n->wx1[l][g], C, n->ch[l + 1], H, W);
            for (size_t i = 0; i < 4 * C * hw; i++) n->gate[i] = n->gate[i] + n->up[i];
        }
        /* gate epilogue */
        const float* gi = n->gate;
        const float* gf = n->gate + (size_t)C * hw;
        const float* gc = n->gate + (size_t)2 * C * hw;
        const float* go = n->gate + (size_t)3 * C * hw;
        for (int o = 0; o < C; o++) {
            const float bi = n->bh[l][0][o], bf = n->bh[l][1][o], bc = n->bh[l][2][o], bo = n->bh[l][3][o];
            for (size_t p = 0; p < hw; p++) {
                const size_t idx = (size_t)o * hw + p;
                const float cold = n->c[l][idx];
#if EIG_GATE_ORDER
                const float zi = (gi[idx] + bi) + n->peep[l][0][idx] * cold;   /* (-ffp-contract=off: rounded product, then add) */
                const float zf = (gf[idx] + bf) + n->peep[l][1][idx] * cold;
                const float zc = gc[idx] + bc;
                const float zo = (go[idx] + bo) + n->peep[l][2][idx] * cold;
                const float ii = det_sigmoidf(zi);
                const float ff = det_sigmoidf(zf);
                const float oo = det_sigmoidf(zo);
                float cc = det_tanhf(zc);
                cc = cc * ii;
                const float fc = ff * cold;
                const float cnew = cc + fc;
#else
                float zi = gi[idx] + bi; zi = fmaf(n->peep[l][0][idx], cold, zi);
                float zf = gf[idx] + bf; zf = fmaf(n->peep[l][1][idx], cold, zf);
                float zc = gc[idx] + bc;
                float zo = go[idx] + bo; zo = fmaf(n->peep[l][2][idx], cold, zo);
                const float ii = det_sigmoidf(zi);
                const float ff = det_sigmoidf(zf);
                const float gg = det_tanhf(zc);
                const float oo = det_sigmoidf(zo);
                const float gi_ = gg * ii;
                const float cnew = fmaf(ff, cold, gi_);
#endif
                n->c[l][idx] = cnew;
                n->hn[l][idx] = oo * det_tanhf(cnew);
            }
        }
        { float* t = n->h[l]; n->h[l] = n->hn[l]; n->hn[l] = t; }
    predict:
        /* P_l = act(ConvP_l(R_l)) */
        fill_padded(n->pad, n->h[l], C, H, W, 0);
        memset(n->gate, 0, sizeof(float) * C * hw);
        if (eig_wino_op(n->wino_mask, 2, l, C, C, H, W, l == L - 1)) {
            const int wm = EIG_WINO_M;
            float* M = (float*)calloc(wino_m_floats(C, H, W, wm), sizeof(float));
            wino_accumulate(M, n->pad, n->convP_w[l], C, C, H, W, wm);
            wino_finish(n->gate, M, C, H, W, wm);
            free(M);
        } else
        conv3x3_chain(n->gate, n->pad, n->convP_w[l], C, C, H, W);
        for (int o = 0; o < C; o++) {
            const float b = n->convP_b[l][o];
            for (size_t p = 0; p < hw; p++) {
                float v = n->gate[(size_t)o * hw + p] + b;
                v = relu(v);
                if (l == 0 && v > 1.0f) v = 1.0f; /* clipped_relu(., 1.0) */
                n->P[l][(size_t)o * hw + p] = v;
            }
        }
    }
}

/*
 * Roll one image through PredNet the way test_prednet is called by the reference:
 * n_repeat steps on the constant frame, then n_ext steps feeding the prediction back,
 * state reset before (reset_at = n_repeat + n_ext).
 *   img        : uint8 [C0][H][W] (planar)
 *   out_frames : uint8 [n_repeat + n_ext][C0][H][W] -- every P0 quantised as write_image does
 *   out_p0     : optional float [n_repeat + n_ext][C0][H][W] (pre-quantisation P0)
 *   requant    : 0 = extension frames feed back the float prediction (upstream behaviour),
 *                1 = feed back the uint8-quantised prediction / 255
 * returns 0, or -1 on bad arguments.
 */
int eig_oracle_prednet_rollout_order(int L, const int* channels, int W, int H, const float* const* tensors,
                                     const uint8_t* img, int n_repeat, int n_ext, int requant,
                                     uint8_t* out_frames, float* out_p0, int order);
int eig_oracle_prednet_rollout(int L, const int* channels, int W, int H, const float* const* tensors,
                               const uint8_t* img, int n_repeat, int n_ext, int requant,
                               uint8_t* out_frames, float* out_p0)
{
    return eig_oracle_prednet_rollout_order(L, channels, W, H, tensors, img, n_repeat, n_ext, requant, out_frames, out_p0, 0);
}

int eig_oracle_prednet_rollout_wino(int L, const int* channels, int W, int H, const float* const* tensors,
                                    const uint8_t* img, int n_repeat, int n_ext, int requant,
                                    uint8_t* out_frames, float* out_p0, int order, int wino_mask);
/* order: 0 = the build's canonical arithmetic with direct convolutions, 1 = the reference's element-wise order (lstm_reference_order) */
int eig_oracle_prednet_rollout_order(int L, const int* channels, int W, int H, const float* const* tensors,
                                     const uint8_t* img, int n_repeat, int n_ext, int requant,
                                     uint8_t* out_frames, float* out_p0, int order)
{
    return eig_oracle_prednet_rollout_wino(L, channels, W, H, tensors, img, n_repeat, n_ext, requant, out_frames, out_p0, order, 0);
}

/* wino_mask (canonical order only): which operators run in their Winograd form -- eig_wino_op; bit 24: eig_wino_fuse_up */
int eig_oracle_prednet_rollout_wino(int L, const int* channels, int W, int H, const float* const* tensors,
                                    const uint8_t* img, int n_repeat, int n_ext, int requant,
                                    uint8_t* out_frames, float* out_p0, int order, int wino_mask)
{
    if (L < 1 || L > EIG_MAX_LAYERS || order < 0 || order > 1) return -1;
    if ((W % (1 << (L - 1))) || (H % (1 << (L - 1)))) return -1;
    prednet_t n;
    memset(&n, 0, sizeof(n));
    n.L = L;
    n.order = order;
    n.wino_mask = wino_mask;
    for (int l = 0; l < L; l++) { n.ch[l] = channels[l]; n.W[l] = W >> l; n.H[l] = H >> l; }
    bind_tensors(&n, tensors);
    prednet_alloc(&n);
    prednet_reset(&n);
    const size_t fsz = (size_t)channels[0] * H * W;
    float* x = (float*)malloc(sizeof(float) * fsz);
    for (size_t i = 0; i < fsz; i++) x[i] = (float)img[i] / 255.0f; /* read_image: astype(float32) / 255 */
    for (int t = 0; t < n_repeat + n_ext; t++) {
        if (t >= n_repeat) { /* feed the previous prediction back */
            for (size_t i = 0; i < fsz; i++) {
                if (requant) x[i] = (float)(uint8_t)(int)(n.P[0][i] * 255.0f) / 255.0f;
                else x[i] = n.P[0][i];
            }
        }
        prednet_step(&n, x);
        for (size_t i = 0; i < fsz; i++) out_frames[(size_t)t * fsz + i] = (uint8_t)(int)(n.P[0][i] * 255.0f);
        if (out_p0) memcpy(out_p0 + (size_t)t * fsz, n.P[0], sizeof(float) * fsz);
    }
    free(x);
    prednet_free(&n);
    return 0;
}

/* Single conv chain exposed for kernel-level parity tests:
 * out[o][y][x] = chain over the full-resolution sources in list order + (one fp32 addition) chain over the
 * unpooled sources (up[s] = 1: [Cin_s][H/2][W/2], 2x2 form), if there are any. */
int eig_oracle_conv_chain(int ns, const float* const* src, const int* cin, const int* up,
                          const float* const* w, int Cout, int H, int W, float* out)
{
    size_t maxc = 0;
    for (int s = 0; s < ns; s++) if ((size_t)cin[s] > maxc) maxc = (size_t)cin[s];
    float* pad = (float*)malloc(sizeof(float) * maxc * (H + 2) * (W + 2));
    memset(out, 0, sizeof(float) * (size_t)Cout * H * W);
    for (int s = 0; s < ns; s++) {
        if (up[s]) continue;
        fill_padded(pad, src[s], cin[s], H, W, 0);
        conv3x3_chain(out, pad, w[s], Cout, cin[s], H, W);
    }
    int n_up = 0;
    for (int s = 0; s < ns; s++) n_up += up[s] != 0;
    if (n_up) {
        float* u = (float*)calloc((size_t)Cout * H * W, sizeof(float));
        for (int s = 0; s < ns; s++)
            if (up[s]) conv_up2x2_chain(u, src[s], w[s], Cout, cin[s], H, W);
        for (size_t i = 0; i < (size_t)Cout * H * W; i++) out[i] = out[i] + u[i];
        free(u);
    }
    free(pad);
    return 0;
}

/* ----------------------------------------------------------------------------------------------
 * Optical flow (OpenCV restatement)
 * ---------------------------------------------------------------------------------------------- */
static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

/* cv::cvtColor(BGR2GRAY) on 8u, OpenCV 4.x: (R*9798 + G*19235 + B*3735 + 2^14) >> 15.
 * rgb is planar [3][H][W] in R,G,B order; for c_dim == 1 the PNG is gray and imread replicates
 * it to 3 channels, for which the formula is the identity. */
void eig_oracle_gray(const uint8_t* img, int c_dim, int H, int W, uint8_t* gray)
{
    const size_t hw = (size_t)H * W;
    if (c_dim == 1) { memcpy(gray, img, hw); return; }
    for (size_t i = 0; i < hw; i++) {
        const int r = img[i], g = img[hw + i], b = img[2 * hw + i];
        gray[i] = (uint8_t)((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15);
    }
}

/* cv::pyrDown on 8u: 5x5 Gaussian [1 4 6 4 1]^2 / 256, rounding (s+128)>>8, BORDER_REFLECT_101 */
static void pyr_down(const uint8_t* src, int H, int W, uint8_t* dst, int Hd, int Wd)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < Hd; y++)
        for (int x = 0; x < Wd; x++) {
            int s = 0;
            for (int j = 0; j < 5; j++) {
                const int sy = reflect101(2 * y + j - 2, H);
                int rs = 0;
                for (int i = 0; i < 5; i++) rs += k[i] * src[(size_t)sy * W + reflect101(2 * x + i - 2, W)];
                s += k[j] * rs;
            }
            dst[(size_t)y * Wd + x] = (uint8_t)((s + 128) >> 8);
        }
}

typedef struct {
    int max_corners;      /* 100  */
    double quality_level; /* 0.3  */
    double min_distance;  /* 7    */
    int block_size;       /* 7    */
    int win;              /* 15   */
    int max_level;        /* 2    */
    int max_iter;         /* 10   */
    double epsilon;       /* 0.03 */
    double min_eig_thr;   /* 1e-4 */
} lk_params_t;

/* cv::cornerMinEigenVal(gray, eig, blockSize, ksize=3), canonical arithmetic: exact integer Sobel
 * and box sums, then fp32 scaling and eigenvalue exactly as calcMinEigenVal orders it. */
void eig_oracle_min_eig(const uint8_t* g, int H, int W, int block, float* eig)
{
    int* dx = (int*)malloc(sizeof(int) * (size_t)H * W);
    int* dy = (int*)malloc(sizeof(int) * (size_t)H * W);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int p[3][3];
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 3; i++) p[j][i] = g[(size_t)reflect101(y + j - 1, H) * W + reflect101(x + i - 1, W)];
            dx[(size_t)y * W + x] = (p[0][2] - p[0][0]) + 2 * (p[1][2] - p[1][0]) + (p[2][2] - p[2][0]);
            dy[(size_t)y * W + x] = (p[2][0] - p[0][0]) + 2 * (p[2][1] - p[0][1]) + (p[2][2] - p[0][2]);
        }
    /* scale = 1 / (2^(ksize-1) * blockSize * 255); applied to both derivatives -> scale^2 on products */
    const double sc = 1.0 / ((double)(1 << 2) * block * 255.0);
    const float SC = (float)(sc * sc);
    const int r0 = block / 2; /* anchor = centre; window [-r0, block-1-r0] */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int sxx = 0, sxy = 0, syy = 0;
            for (int j = 0; j < block; j++) {
                const int yy = reflect101(y + j - r0, H);
                for (int i = 0; i < block; i++) {
                    const int xx = reflect101(x + i - r0, W);
                    const int a = dx[(size_t)yy * W + xx], b = dy[(size_t)yy * W + xx];
                    sxx += a * a; sxy += a * b; syy += b * b;
                }
            }
            const float fa = (float)sxx * SC, fb = (float)sxy * SC, fc = (float)syy * SC;
            const float a = fa * 0.5f, b = fb, c = fc * 0.5f;
            const float d = a - c;
            const float t = d * d;
            const float u = b * b;
            eig[(size_t)y * W + x] = (a + c) - sqrtf(t + u);
        }
    free(dx); free(dy);
}

typedef struct { float v; int idx; } cand_t;
static int cand_cmp(const void* a, const void* b)
{
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return (x->idx > y->idx) ? -1 : (x->idx < y->idx) ? 1 : 0; /* greaterThanPtr: ties -> higher address first */
}

/* cv::goodFeaturesToTrack(gray, maxCorners, qualityLevel, minDistance, mask=None, blockSize)
 * corners: float [max_corners][2] (x, y); returns the number found. */
int eig_oracle_good_features(const uint8_t* gray, int H, int W, const lk_params_t* p, float* corners)
{
    const size_t hw = (size_t)H * W;
    float* eig = (float*)malloc(sizeof(float) * hw);
    eig_oracle_min_eig(gray, H, W, p->block_size, eig);
    float maxv = eig[0];
    for (size_t i = 1; i < hw; i++) if (eig[i] > maxv) maxv = eig[i];
    const float thr = (float)((double)maxv * p->quality_level);
    for (size_t i = 0; i < hw; i++) if (!(eig[i] > thr)) eig[i] = 0.0f; /* THRESH_TOZERO */
    cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * hw);
    int nc = 0;
    for (int y = 1; y < H - 1; y++)
        for (int x = 1; x < W - 1; x++) {
            const float v = eig[(size_t)y * W + x];
            if (v == 0.0f) continue;
            float m = v; /* 3x3 dilate; out-of-image never reached since 1 <= y,x <= n-2 */
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    const float q = eig[(size_t)(y + j) * W + x + i];
                    if (q > m) m = q;
                }
            if (v == m) { cand[nc].v = v; cand[nc].idx = y * W + x; nc++; }
        }
    qsort(cand, nc, sizeof(cand_t), cand_cmp);
    int n = 0;
    const float md2 = (float)(p->min_distance * p->min_distance);
    for (int i = 0; i < nc && n < p->max_corners; i++) {
        const int x = cand[i].idx % W, y = cand[i].idx / W;
        int good = 1;
        if (p->min_distance >= 1) {
            for (int j = 0; j < n; j++) {
                const float ddx = (float)x - corners[2 * j], ddy = (float)y - corners[2 * j + 1];
                if (ddx * ddx + ddy * ddy < md2) { good = 0; break; }
            }
        }
        if (good) { corners[2 * n] = (float)x; corners[2 * n + 1] = (float)y; n++; }
    }
    free(cand); free(eig);
    return n;
}

/* cv::detail::calcSharrDeriv: int16 (dx, dy) Scharr derivative, REFLECT_101 at the image border */
static void scharr_deriv(const uint8_t* g, int H, int W, int16_t* d /* [H][W][2] */)
{
    for (int y = 0; y < H; y++) {
        const int y0 = reflect101(y - 1, H), y2 = reflect101(y + 1, H);
        for (int x = 0; x < W; x++) {
            const int x0 = reflect101(x - 1, W), x2 = reflect101(x + 1, W);
            const int a00 = g[(size_t)y0 * W + x0], a01 = g[(size_t)y0 * W + x], a02 = g[(size_t)y0 * W + x2];
            const int a10 = g[(size_t)y * W + x0], a12 = g[(size_t)y * W + x2];
            const int a20 = g[(size_t)y2 * W + x0], a21 = g[(size_t)y2 * W + x], a22 = g[(size_t)y2 * W + x2];
            d[((size_t)y * W + x) * 2] = (int16_t)(3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20));
            d[((size_t)y * W + x) * 2 + 1] = (int16_t)(3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02));
        }
    }
}

static inline int img_at(const uint8_t* g, int H, int W, int y, int x) /* pyramid border: REFLECT_101 */
{
    return g[(size_t)reflect101(y, H) * W + reflect101(x, W)];
}
static inline int der_at(const int16_t* d, int H, int W, int y, int x, int k) /* deriv border: constant 0 */
{
    if (y < 0 || y >= H || x < 0 || x >= W) return 0;
    return d[((size_t)y * W + x) * 2 + k];
}
static inline int cv_round(float v) { return (int)lrintf(v); } /* cvRound: round-half-even */
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/*
 * cv::calcOpticalFlowPyrLK(prev, next, prevPts, None, winSize, maxLevel, criteria) -- LKTrackerInvoker.
 * next_pts [n][2], status [n].
 */
void eig_oracle_pyr_lk(const uint8_t* g0, const uint8_t* g1, int H, int W, const lk_params_t* p,
                       const float* prev_pts, int n, float* next_pts, uint8_t* status)
{
    const int win = p->win;
    /* buildOpticalFlowPyramid: a level exists only while both dims stay > winSize */
    const uint8_t* I[8]; const uint8_t* J[8]; int Hs[8], Ws[8];
    uint8_t* own[16]; int nown = 0;
    I[0] = g0; J[0] = g1; Hs[0] = H; Ws[0] = W;
    int max_level = 0;
    for (int l = 1; l <= p->max_level && l < 8; l++) {
        const int hd = (Hs[l - 1] + 1) / 2, wd = (Ws[l - 1] + 1) / 2;
        if (wd <= win || hd <= win) break;
        uint8_t* a = (uint8_t*)malloc((size_t)hd * wd);
        uint8_t* b = (uint8_t*)malloc((size_t)hd * wd);
        pyr_down(I[l - 1], Hs[l - 1], Ws[l - 1], a, hd, wd);
        pyr_down(J[l - 1], Hs[l - 1], Ws[l - 1], b, hd, wd);
        I[l] = a; J[l] = b; Hs[l] = hd; Ws[l] = wd; own[nown++] = a; own[nown++] = b;
        max_level = l;
    }
    int max_count = p->max_iter < 0 ? 0 : (p->max_iter > 100 ? 100 : p->max_iter);
    double eps = p->epsilon < 0 ? 0 : (p->epsilon > 10 ? 10 : p->epsilon);
    eps *= eps;
    for (int i = 0; i < n; i++) status[i] = 1;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.0f / (float)(1 << 20);
    int16_t* Iw = (int16_t*)malloc(sizeof(int16_t) * win * win);
    int16_t* dIw = (int16_t*)malloc(sizeof(int16_t) * win * win * 2);
    for (int level = max_level; level >= 0; level--) {
        const int h = Hs[level], w = Ws[level];
        int16_t* deriv = (int16_t*)malloc(sizeof(int16_t) * (size_t)h * w * 2);
        scharr_deriv(I[level], h, w, deriv);
        const float lscale = (float)(1.0 / (double)(1 << level));
        for (int pt = 0; pt < n; pt++) {
            float px = prev_pts[2 * pt] * lscale, py = prev_pts[2 * pt + 1] * lscale;
            float nx, ny;
            if (level == max_level) { nx = px; ny = py; }
            else { nx = next_pts[2 * pt] * 2.0f; ny = next_pts[2 * pt + 1] * 2.0f; }
            next_pts[2 * pt] = nx; next_pts[2 * pt + 1] = ny;
            px -= half; py -= half;
            const int ipx = (int)floorf(px), ipy = (int)floorf(py);
            if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            float a = px - (float)ipx, b = py - (float)ipy;
            int iw00 = cv_round((1.0f - a) * (1.0f - b) * 16384.0f);
            int iw01 = cv_round(a * (1.0f - b) * 16384.0f);
            int iw10 = cv_round((1.0f - a) * b * 16384.0f);
            int iw11 = 16384 - iw00 - iw01 - iw10;
            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    const int yy = y + ipy, xx = x + ipx;
                    const int ival = DESCALE(img_at(I[level], h, w, yy, xx) * iw00 + img_at(I[level], h, w, yy, xx + 1) * iw01 +
                                             img_at(I[level], h, w, yy + 1, xx) * iw10 + img_at(I[level], h, w, yy + 1, xx + 1) * iw11, 14 - 5);
                    const int ixval = DESCALE(der_at(deriv, h, w, yy, xx, 0) * iw00 + der_at(deriv, h, w, yy, xx + 1, 0) * iw01 +
                                              der_at(deriv, h, w, yy + 1, xx, 0) * iw10 + der_at(deriv, h, w, yy + 1, xx + 1, 0) * iw11, 14);
                    const int iyval = DESCALE(der_at(deriv, h, w, yy, xx, 1) * iw00 + der_at(deriv, h, w, yy, xx + 1, 1) * iw01 +
                                              der_at(deriv, h, w, yy + 1, xx, 1) * iw10 + der_at(deriv, h, w, yy + 1, xx + 1, 1) * iw11, 14);
                    Iw[y * win + x] = (int16_t)ival;
                    dIw[(y * win + x) * 2] = (int16_t)ixval;
                    dIw[(y * win + x) * 2 + 1] = (int16_t)iyval;
                    sA11 += (int64_t)ixval * ixval; sA12 += (int64_t)ixval * iyval; sA22 += (int64_t)iyval * iyval;
                }
            const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float dd = A11 - A22;
            const float mn = A22 + A11;
            const float q = dd * dd;
            const float r4 = 4.0f * A12;
            const float r = r4 * A12;
            const float minEig = (mn - sqrtf(q + r)) / (float)(2 * win * win);
            if ((double)minEig < p->min_eig_thr || D < 1.1920928955078125e-7f) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.0f / D;
            nx -= half; ny -= half;
            float pdx = 0.0f, pdy = 0.0f;
            for (int j = 0; j < max_count; j++) {
                const int inx = (int)floorf(nx), iny = (int)floorf(ny);
                if (inx < -win || inx >= w || iny < -win || iny >= h) {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                a = nx - (float)inx; b = ny - (float)iny;
                iw00 = cv_round((1.0f - a) * (1.0f - b) * 16384.0f);
                iw01 = cv_round(a * (1.0f - b) * 16384.0f);
                iw10 = cv_round((1.0f - a) * b * 16384.0f);
                iw11 = 16384 - iw00 - iw01 - iw10;
                int64_t sb1 = 0, sb2 = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        const int yy = y + iny, xx = x + inx;
                        const int jv = DESCALE(img_at(J[level], h, w, yy, xx) * iw00 + img_at(J[level], h, w, yy, xx + 1) * iw01 +
                                               img_at(J[level], h, w, yy + 1, xx) * iw10 + img_at(J[level], h, w, yy + 1, xx + 1) * iw11, 14 - 5);
                        const int diff = jv - Iw[y * win + x];
                        sb1 += (int64_t)diff * dIw[(y * win + x) * 2];
                        sb2 += (int64_t)diff * dIw[(y * win + x) * 2 + 1];
                    }
                const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
                const float t0 = A12 * b2, t1 = A22 * b1, t2 = A12 * b1, t3 = A11 * b2;
                const float dxv = (t0 - t1) * D;
                const float dyv = (t2 - t3) * D;
                nx += dxv; ny += dyv;
                next_pts[2 * pt] = nx + half; next_pts[2 * pt + 1] = ny + half;
                /* delta.ddot(delta) <= epsilon  (double) */
                if ((double)dxv * (double)dxv + (double)dyv * (double)dyv <= eps) break;
                if (j > 0 && (double)fabsf(dxv + pdx) < 0.01 && (double)fabsf(dyv + pdy) < 0.01) {
                    next_pts[2 * pt] -= dxv * 0.5f; next_pts[2 * pt + 1] -= dyv * 0.5f;
                    break;
                }
                pdx = dxv; pdy = dyv;
            }
            if (status[pt] && level == 0) { /* bounds check of the error block (err requested, flags=0) */
                const float fx = next_pts[2 * pt] - half, fy = next_pts[2 * pt + 1] - half;
                const int ix = (int)floorf(fx), iy = (int)floorf(fy);
                if (ix < -win || ix >= w || iy < -win || iy >= h) status[pt] = 0;
            }
        }
        free(deriv);
    }
    free(Iw); free(dIw);
    for (int i = 0; i < nown; i++) free(own[i]);
}

/*
 * lucas_kanade(img0, img1) of Optical_Flow_Analyzer: gray -> goodFeaturesToTrack(img0) ->
 * calcOpticalFlowPyrLK -> vectors [x0, y0, x1-x0, y1-y0] for status==1 (float32 arithmetic).
 * img0/img1: uint8 planar [c_dim][H][W].  vectors: float [max_corners][4].  Returns the count.
 */
int eig_oracle_lucas_kanade(const uint8_t* img0, const uint8_t* img1, int c_dim, int H, int W,
                            const lk_params_t* p, float* vectors)
{
    const size_t hw = (size_t)H * W;
    uint8_t* g0 = (uint8_t*)malloc(hw);
    uint8_t* g1 = (uint8_t*)malloc(hw);
    eig_oracle_gray(img0, c_dim, H, W, g0);
    eig_oracle_gray(img1, c_dim, H, W, g1);
    float* pts = (float*)malloc(sizeof(float) * 2 * (p->max_corners > 0 ? p->max_corners : 1));
    float* nxt = (float*)malloc(sizeof(float) * 2 * (p->max_corners > 0 ? p->max_corners : 1));
    uint8_t* st = (uint8_t*)malloc(p->max_corners > 0 ? p->max_corners : 1);
    const int n = eig_oracle_good_features(g0, H, W, p, pts);
    int nv = 0;
    if (n > 0) {
        eig_oracle_pyr_lk(g0, g1, H, W, p, pts, n, nxt, st);
        for (int i = 0; i < n; i++)
            if (st[i]) {
                vectors[4 * nv] = pts[2 * i];
                vectors[4 * nv + 1] = pts[2 * i + 1];
                vectors[4 * nv + 2] = nxt[2 * i] - pts[2 * i];
                vectors[4 * nv + 3] = nxt[2 * i + 1] - pts[2 * i + 1];
                nv++;
            }
    }
    free(g0); free(g1); free(pts); free(nxt); free(st);
    return nv;
}

/* pyrDown exported for tests */
void eig_oracle_pyr_down(const uint8_t* src, int H, int W, uint8_t* dst)
{
    pyr_down(src, H, W, dst, (H + 1) / 2, (W + 1) / 2);
}
