// eigen_engine.hip -- C ABI (include/eigen_engine.h) of the MI355X fitness engine: handle, device workspaces,
// weight packing, kernel launch sequencing.  All compute is in the HIP kernels of conv_mfma.h / cppn_kernel.h /
// flow_kernels.h / score_kernels.h; there is no CPU fallback anywhere in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/eigen_engine.h"
#define EIG_ENGINE_UNIT 1   // (conv_mfma.h: the non-template kernels are defined in this unit)
#include "conv_mfma.h"
#include "wino_launch.h"   // the Winograd kernels live in wino4_kernels.hip / wino4t_kernels.hip / wino4h_kernels.hip / wino4h_kernels.hip
#include "cppn_kernel.h"
#include "farneback_kernels.h"
#include "flow_kernels.h"
#include "score_kernels.h"

using namespace eig;

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x)                                                                                      \
    do {                                                                                               \
        hipError_t _e = (x);                                                                           \
        if (_e != hipSuccess) return fail(EIGEN_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(_e)); \
    } while (0)

namespace {

struct ConvOp {
    int tl_seen = 0;  // EIG_TIMING builds: launches seen (timeline dump)
    int last_grid = 0, last_waves = 4;  // geometry of the last launch (launch_conv), for the EIG_TIMING read-back
    int epi = 0, NI = 4, TW = 16, layer = 0;
    int nsrc = 0;
    int src_C[3] = {0, 0, 0};
    int src_Ct[3] = {0, 0, 0};  // channels of the source TENSOR when only its first src_C channels are read (0: = src_C)
    int H = 0, W = 0, Cout = 0, n_nblk = 0, krows = 0;
    float* d_wpk = nullptr;
    float* d_wraw = nullptr;  // image-layer ConvP only: the unpacked OIHW weights for convp0_direct_kernel
    double macs = 0;  // algorithmic multiply-accumulates per image (real channels only)
    double ms = 0;    // profiling accumulator
    int launches = 0;
    // Winograd ConvLSTM below the top layer: its unpooled source R_{l+1} rides inside the same chains (conv_wino4.h: up_fused)
    bool fused = false;
    int up_C = 0, up_kb = 0;
    bool wino = false;  // Winograd F(4x4, 3x3) form (conv_wino4.h); epi stays the operator's epilogue
};

struct Layer {
    int C = 0, H = 0, W = 0;
    float* h[2] = {nullptr, nullptr};
    float *c = nullptr, *P = nullptr, *E = nullptr;
    float *bias_lstm = nullptr, *peep = nullptr, *biasA = nullptr, *biasP = nullptr;
    ConvOp convA, lstm, convP;
    // Step-0 operators: after reset_state() h_l = 0 and P_l = 0, hence the second half of every E_l (relu(P - A), A >= 0)
    // is 0 as well.  Their terms fma(0, w, acc) leave the chain untouched, so the first step runs the same chains over the
    // non-zero sources only: ConvA reads the first half of E_{l-1}, the ConvLSTM the first half of E_l and R_{l+1}.
    ConvOp convA_t0, lstm_t0;
    // The unpooled source R_{l+1} of the ConvLSTM in its 2x2 form (conv_mfma.h: EPI_UP4), launched at the resolution of
    // layer l+1 ahead of the ConvLSTM launch, which adds the result to its own chain (eigen_engine::d_raw4).
    ConvOp up4;
};

template <typename T> struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc((void**)&p, n * sizeof(T)) != hipSuccess) return -1;
        cap = n;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct eigen_engine {
    eigen_config cfg;
    int L = 0, B = 0, C0 = 0, H = 0, W = 0, K = 0;
    Layer layer[EIGEN_MAX_LAYERS];
    bool have_weights = false, have_grid = false;
    int n_planes = 0;
    double* d_planes = nullptr;
    // genome staging
    DevBuf<int32_t> g_node_off, g_edge_off, g_edge_src, g_out_node;
    DevBuf<uint8_t> g_node_act;
    DevBuf<double> g_node_bias, g_node_resp, g_edge_w;
    // images / frames
    uint8_t* d_images = nullptr;  // [B][C0][H][W]
    uint8_t* d_frames = nullptr;  // [B][3][C0][H][W]
    // flow
    int n_levels = 1;             // pyramid levels that exist (max_level + 1)
    int lvH[FLOW_MAX_LEVELS], lvW[FLOW_MAX_LEVELS];
    uint8_t* d_gray[2][FLOW_MAX_LEVELS] = {{nullptr}};
    short2* d_deriv[FLOW_MAX_LEVELS] = {nullptr};
    float* d_eig = nullptr;
    unsigned long long* d_cand = nullptr;
    float *d_corners = nullptr, *d_next = nullptr, *d_vectors = nullptr;
    uint8_t* d_status = nullptr;
    int *d_ncorners = nullptr, *d_counts = nullptr;
    double* d_fitness = nullptr;
    float* d_zeros = nullptr;  // DMA source for zero fill (conv_mfma.h)
    // Farneback dense flow (cfg.flow_method == EIGEN_FLOW_FARNEBACK): allocated on first use
    float *fb_I = nullptr, *fb_R0 = nullptr, *fb_R1 = nullptr, *fb_M = nullptr, *fb_V = nullptr, *fb_flow[2] = {nullptr, nullptr};
    float* d_raw4 = nullptr;   // partial chains of the unpooled source, [B][4 classes][n_nblk*NB][H/2][W/2], reused by all layers
    size_t raw4_floats = 0;
    // timing
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pev0 = nullptr, pev1 = nullptr;
    int n_cu = 256;  // compute units of the device (launch-shape heuristics)
    bool profile_convs = false;
    double ms[6] = {0, 0, 0, 0, 0, 0};
    int hflip = 0;  // which h buffer holds the current R
    // ConvP_l (l > 0) is read by nobody until ConvA_l of the NEXT step: forked onto a side stream it can fill the CUs the last round of the ConvLSTM below it leaves idle
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join[EIGEN_MAX_LAYERS] = {nullptr};
};

// ------------------------------------------------------------------------------------------------ helpers
static int pad4(int c) { return (c + 3) & ~3; }

// ---- Farneback constants (host; the same double-precision recipe as oracle/farneback.c, checked by tests/test_gpu_parity.py) ----
static int fb_levels_used(int H, int W, int levels)  // calcOpticalFlowFarneback: no level below 32 pixels
{
    int k = 0;
    double scale = 1;
    for (; k < levels; ++k) {
        scale *= 0.5;
        if (W * scale < 32 || H * scale < 32) break;
    }
    return k;
}
static int fb_grid_step(int H, int W, int step, int max_vectors)
{
    int s = step;
    while ((H / s) * (W / s) > max_vectors) s += step;
    return s;
}
static FbBlur fb_blur_kernel(int k)  // GaussianBlur of pyramid level k: sigma = (2^k - 1) / 2, ksize = max(cvRound(5 sigma) | 1, 3)
{
    FbBlur b;
    memset(&b, 0, sizeof(b));
    const double sigma = ((double)(1 << k) - 1) * 0.5;
    int ksize = (int)lrint(sigma * 5) | 1;
    if (ksize < 3) ksize = 3;
    b.r = ksize / 2;
    if (sigma <= 0 && ksize == 3) { b.k[0] = 0.5f; b.k[1] = 0.25f; return b; }
    const double sx = sigma > 0 ? sigma : ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double s2 = -0.5 / (sx * sx);
    std::vector<double> c(ksize);
    double sum = 0;
    for (int i = 0; i < ksize; ++i) { const double x = i - (ksize - 1) * 0.5; c[i] = exp(s2 * x * x); sum += c[i]; }
    sum = 1.0 / sum;
    for (int j = 0; j <= b.r; ++j) b.k[j] = (float)(c[b.r + j] * sum);
    return b;
}
static FbConst fb_poly_constants(int n, double sigma)  // FarnebackPrepareGaussian
{
    FbConst c;
    memset(&c, 0, sizeof(c));
    c.poly_n = n;
    if (sigma < 1.1920928955078125e-07) sigma = n * 0.3;
    std::vector<double> gd(2 * n + 1);
    std::vector<float> gf(2 * n + 1);
    double s = 0;
    for (int x = -n; x <= n; ++x) { gd[x + n] = exp(-x * x / (2 * sigma * sigma)); s += gd[x + n]; }
    s = 1.0 / s;
    for (int x = -n; x <= n; ++x) gf[x + n] = (float)(gd[x + n] * s);
    for (int x = 0; x <= n; ++x) { c.g[x] = gf[x + n]; c.xg[x] = (float)(x * gf[x + n]); c.xxg[x] = (float)(x * x * gf[x + n]); }
    double G[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; ++y)
        for (int x = -n; x <= n; ++x) {
            const double w = (double)gf[y + n] * (double)gf[x + n];
            G[0][0] += w;
            G[1][1] += w * x * x;
            G[3][3] += w * x * x * x * x;
            G[5][5] += w * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    double A[6][12];  // Gauss-Jordan with partial pivoting
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 12; ++j) A[i][j] = j < 6 ? G[i][j] : (j - 6 == i ? 1.0 : 0.0);
    for (int col = 0; col < 6; ++col) {
        int p = col;
        for (int r = col + 1; r < 6; ++r) if (fabs(A[r][col]) > fabs(A[p][col])) p = r;
        if (p != col) for (int j = 0; j < 12; ++j) std::swap(A[col][j], A[p][j]);
        const double d = 1.0 / A[col][col];
        for (int j = 0; j < 12; ++j) A[col][j] *= d;
        for (int r = 0; r < 6; ++r) {
            if (r == col) continue;
            const double f = A[r][col];
            if (f != 0.0) for (int j = 0; j < 12; ++j) A[r][j] -= f * A[col][j];
        }
    }
    c.ig[0] = (float)A[1][7]; c.ig[1] = (float)A[0][9]; c.ig[2] = (float)A[3][9]; c.ig[3] = (float)A[5][11];
    return c;
}

static void choose_ni(int Cout, bool lstm, int* NI, int* n_nblk)
{
    if (lstm) { *NI = 4; *n_nblk = (Cout + 15) / 16; return; }
    int best = 1; double beste = -1;
    for (int ni = 1; ni <= 4; ++ni) {
        const int nb = (Cout + 16 * ni - 1) / (16 * ni);
        const double eff = (double)Cout / (nb * 16.0 * ni) + 1e-3 * ni;  // ties -> larger tile
        if (eff > beste) { beste = eff; best = ni; }
    }
    *NI = best; *n_nblk = (Cout + 16 * best - 1) / (16 * best);
}

// Tile shape of an operator: 16 x 16, or 8 x 8 where that covers the map at least 15 % better.
static int choose_tw(int H, int W)
{
    auto util = [&](int tw) {
        const int th = (tw == 8) ? 8 : 16;
        const int ty = (H + th - 1) / th, tx = (W + tw - 1) / tw;
        return (double)H * W / ((double)ty * th * tx * tw);
    };
    // 16 x 16 tiles unless 8 x 8 tiles cover the map at least 15 % better: the 8-wide instantiations have no branch-free staging path
    // and a conflicted LDS row stride (conv_mfma.h) -- measured at 160 x 120 maps (640x480 colour, layer 2: profiles/r04_c_perop_shapes.txt):
    // 8 x 8 tiles at 100 % cover ran at 0.78 of peak, 16 x 16 tiles at 93.75 % cover at 0.86.  EIGEN_TW8_FACTOR for A/Bs.
    static const double tw8_factor = getenv("EIGEN_TW8_FACTOR") ? atof(getenv("EIGEN_TW8_FACTOR")) : 1.15;
    const int sq = (util(16) * tw8_factor + 1e-9 >= util(8)) ? 16 : 8;
    return sq;
}

// Pack OIHW weights of one fused conv into [n_nblk][krows][NB]; row = (source, channel (padded to 4), tap).
// srcw[s][g] points at [Cout][Cin_s][3][3]; g = 0 for plain convs, 0..3 (i,f,c,o) for the LSTM.
// lstm: 0 plain conv, 1 gates as four 16-channel tiles (column = gate*16 + channel), 2 packed for C <= 4
// (ONE 16-column tile, column = gate*4 + channel).
static std::vector<float> pack_weights(const ConvOp& op, const float* const srcw[3][4], int lstm)
{
    const int NB = op.NI * 16;
    std::vector<float> out((size_t)op.n_nblk * op.krows * NB, 0.0f);
    for (int nb = 0; nb < op.n_nblk; ++nb) {
        size_t row = 0;
        for (int s = 0; s < op.nsrc; ++s) {
            const int Cin = op.src_C[s], Cp = pad4(Cin);
            const int Cw = op.src_Ct[s] ? op.src_Ct[s] : Cin;  // input channels of the weight tensor
            for (int c = 0; c < Cp; ++c)
                for (int tap = 0; tap < 9; ++tap, ++row) {
                    if (c >= Cin) continue;
                    float* dst = &out[((size_t)nb * op.krows + row) * NB];
                    for (int n = 0; n < NB; ++n) {
                        int g = 0, o;
                        if (lstm == 1) { g = n / 16; o = nb * 16 + (n % 16); }
                        else if (lstm == 2) { g = n / 4; o = n % 4; }
                        else o = nb * NB + n;
                        if (o >= op.Cout) continue;
                        // LDS/slab column order: [16 lanes (n % 16)][NI tiles (n / 16)] so that a lane reads its NI values
                        // of a row with one ds_read_b128 (conv_mfma.h: boff)
                        dst[(n % 16) * op.NI + (n / 16)] = srcw[s][g][((size_t)o * Cw + c) * 9 + tap];
                    }
                }
        }
    }
    return out;
}

// Weights of the 2x2 form of `unpool x2 -> conv3x3` for parity class (py, px) of the output pixel (oracle/eig_oracle.c:
// presum_up_weights states the same rule).  Output row 2Y+py reads source rows Y-1, Y, Y (py = 0) or Y, Y, Y+1 (py = 1): tap a
// stands for source row Y+a-1+py and collects ky in {0} / {1,2} (py = 0) or {0,1} / {2} (py = 1); columns likewise.  The
// collected weights are added in fp32 in (ky, kx) row-major order starting from the first one.
static float presum_up_weight(const float* w9, int py, int px, int a, int b)
{
    const int ky0 = py ? (a ? 2 : 0) : (a ? 1 : 0), ky1 = py ? (a ? 2 : 1) : (a ? 2 : 0);
    const int kx0 = px ? (b ? 2 : 0) : (b ? 1 : 0), kx1 = px ? (b ? 2 : 1) : (b ? 2 : 0);
    volatile float s = 0.0f;  // volatile: one fp32 rounding per addition whatever the host compiler's flags
    bool first = true;
    for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) {
            if (first) { s = w9[ky * 3 + kx]; first = false; }
            else s = s + w9[ky * 3 + kx];
        }
    return s;
}

// Pack the 2x2-form weights of ONE unpooled source into [4 classes][n_nblk][krows = Cpad*4][NB]; row = (channel, a, b);
// column order as pack_weights (lstm: 0 plain, 1 four 16-channel gate tiles, 2 packed gates for C <= 4).
static std::vector<float> pack_weights_up4(const ConvOp& op, const float* const srcw[4], int lstm)
{
    const int NB = op.NI * 16;
    const int Cin = op.src_C[0], Cp = pad4(Cin);
    std::vector<float> out((size_t)4 * op.n_nblk * op.krows * NB, 0.0f);
    for (int cls = 0; cls < 4; ++cls)
        for (int nb = 0; nb < op.n_nblk; ++nb)
            for (int c = 0; c < Cin; ++c)
                for (int tap = 0; tap < 4; ++tap) {
                    float* dst = &out[(((size_t)cls * op.n_nblk + nb) * op.krows + (size_t)c * 4 + tap) * NB];
                    for (int n = 0; n < NB; ++n) {
                        int g = 0, o;
                        if (lstm == 1) { g = n / 16; o = nb * 16 + (n % 16); }
                        else if (lstm == 2) { g = n / 4; o = n % 4; }
                        else o = nb * NB + n;
                        if (o >= op.Cout) continue;
                        dst[(n % 16) * op.NI + (n / 16)] = presum_up_weight(srcw[g] + ((size_t)o * Cin + c) * 9, cls >> 1, cls & 1, tap >> 1, tap & 1);
                    }
                }
    (void)Cp;
    return out;
}

// EPI_UP4C (conv_mfma.h): the four classes are the four N-tiles of ONE block: [n_nblk][krows][16 columns][4 classes]
static std::vector<float> pack_weights_up4c(const ConvOp& op, const float* const srcw[4], int lstm)
{
    const int Cin = op.src_C[0];
    std::vector<float> out((size_t)op.n_nblk * op.krows * 64, 0.0f);
    for (int nb = 0; nb < op.n_nblk; ++nb)
        for (int c = 0; c < Cin; ++c)
            for (int tap = 0; tap < 4; ++tap)
                for (int n = 0; n < 16; ++n) {
                    int g = 0, o;
                    if (lstm == 2) { g = n / 4; o = n % 4; }
                    else o = nb * 16 + n;
                    if (o >= op.Cout) continue;
                    for (int cls = 0; cls < 4; ++cls)
                        out[(((size_t)nb * op.krows + (size_t)c * 4 + tap) * 16 + n) * 4 + cls] =
                            presum_up_weight(srcw[g] + ((size_t)o * Cin + c) * 9, cls >> 1, cls & 1, tap >> 1, tap & 1);
                }
    return out;
}

// ---- Winograd F(4x4, 3x3) form of the 3x3 convolutions of layers >= 1 (conv_wino4.h; oracle/eig_oracle.c: wino4_* state the same rule)
// EIGEN_WINOGRAD: bit l = ConvLSTM_l, bit 8 + l = ConvA_l, bit 16 + l = ConvP_l may take the Winograd form (if eligible) AND bit 25 / 26 / 27 enables it for the
// ConvLSTMs / ConvAs / ConvPs as a class (rounds 4-5 had an F(2x2, 3x3) kernel behind the per-operator bits and F(4x4) behind the class bits; round 6 removed the
// F(2x2) kernel -- nothing ran it -- and an operator whose class bit is clear now runs direct).  Default: all of them -- measured faster at every shape tried,
// 256^2 / 512^2 / 640x480 / 160x120, colour and gray.  Eligibility (the same rule in oracle/eig_oracle.c: eig_wino_op) is a property
// of the operator's shape only, never of the batch: results must not depend on the device batch a genome lands in.
//   kind 0 ConvLSTM_l, 1 ConvA_l, 2 ConvP_l; Cin = channels of the full-resolution sources (multiples of 8 each), Cout per gate;
//   H x W = the resolution the convolution runs at; odd H only for an operator of the TOP layer (nothing is pooled / unpooled from it)
#ifndef EIGEN_WINO_DEFAULT
#define EIGEN_WINO_DEFAULT 0x0FFFFFFE   // every eligible operator in Winograd form, the unpooled source inside the ConvLSTM chains (bit 24), F(4x4, 3x3) tiles (bits 25-27)
#endif
// the effective mask of this process: EIGEN_WINOGRAD (default EIGEN_WINO_DEFAULT), bit 24 cleared by EIGEN_WINO_FUSEUP=0 (eigen_winograd_mask; oracle.wino_mask_default)
static int wino_mask_env()
{
    static const int mask = [] {
        int m = getenv("EIGEN_WINOGRAD") && *getenv("EIGEN_WINOGRAD") ? (int)strtol(getenv("EIGEN_WINOGRAD"), nullptr, 0) : EIGEN_WINO_DEFAULT;
        if (getenv("EIGEN_WINO_FUSEUP") && !atoi(getenv("EIGEN_WINO_FUSEUP"))) m &= ~(1 << 24);
        return m;
    }();
    return mask;
}
static bool wino_op(int mask, int kind, int l, int Cin, int Cout, int H, int W, bool top)
{
    if (!((mask >> (8 * kind + l)) & 1) || !((mask >> (25 + kind)) & 1) || l < 1) return false;
    if ((Cin % 8) || (Cout % 16) || (W % 4)) return false;
    if ((H % 2) && !(top && kind != 1)) return false;
    if (kind != 0 && (Cout % 48) && (Cout % 64)) return false;  // plain convolutions: N-blocks of 48 or 64 columns without padding
    return true;
}
// F(4x4, 3x3): U = G g G^T, 6 x 6 (oracle/eig_oracle.c: wino4_w1d / wino4_weights state the same operations in the same order; fmaf = one rounding, this file
// is compiled with -ffp-contract=off)
static void wino4_w1d(float g0, float g1, float g2, float* W)
{
    const float c6 = -1.0f / 6.0f, c24 = 1.0f / 24.0f;
    W[0] = 0.25f * g0;
    const float a = g0 + g2;
    W[1] = (a + g1) * c6; W[2] = (a - g1) * c6;
    const float b = fmaf(4.0f, g2, g0);
    W[3] = fmaf(2.0f, g1, b) * c24; W[4] = fmaf(-2.0f, g1, b) * c24;
    W[5] = g2;
}
static void wino4_weight(const float* g, float* U)
{
    float s[6][3], W[6];
    for (int j = 0; j < 3; ++j) { wino4_w1d(g[j], g[3 + j], g[6 + j], W); for (int i = 0; i < 6; ++i) s[i][j] = W[i]; }
    for (int i = 0; i < 6; ++i) wino4_w1d(s[i][0], s[i][1], s[i][2], U + i * 6);
}
// [n_nblk][K-blocks: 4 channels of one source, sources in order][36 positions][4 channels][16 columns][NI N-tiles]
// lstm: N-tile = gate, output channel = 16 nb + column (srcw[s][gate]); plain convolution: output channel = 16 (NI nb + N-tile) + column (srcw[s][0])
static std::vector<float> pack_weights_wino(int C, int NI, int n_nblk, bool lstm, int nsrc, const int* src_C, const int* src_Cw, const float* const srcw[3][4])
{
    const int kc = W4_KC;   // channels of a packed K-block (conv_wino4.h streams them with a running offset, and one K-block past the end: padding)
    int nkb = 0;
    for (int s = 0; s < nsrc; ++s) nkb += src_C[s] / kc;
    const int npos = W4_NPOS;
    const int uf = wino4_u_floats(NI);
    std::vector<float> out((size_t)n_nblk * nkb * uf + uf, 0.0f);
    float U[36];
    for (int nb = 0; nb < n_nblk; ++nb) {
        int kb0 = 0;
        for (int s = 0; s < nsrc; ++s) {
            for (int c = 0; c < src_C[s]; ++c)
                for (int ni = 0; ni < NI; ++ni)
                    for (int n = 0; n < 16; ++n) {
                        const int o = lstm ? nb * 16 + n : (nb * NI + ni) * 16 + n;
                        if (o >= C) continue;
                        wino4_weight(srcw[s][lstm ? ni : 0] + ((size_t)o * src_Cw[s] + c) * 9, U);
                        float* dst = &out[((size_t)nb * nkb + kb0 + c / kc) * uf];
                        for (int pos = 0; pos < npos; ++pos) dst[((pos * kc + (c % kc)) * 16 + n) * NI + ni] = U[pos];
                    }
            kb0 += src_C[s] / kc;
        }
    }
    return out;
}

template <int NI, int TW, int EPI, bool VEC, bool ONEKB = false, int SPLIT = 0> static hipError_t launch_inst2(const ConvArgs& a, int grid, hipStream_t st)
{
    constexpr int NT = SPLIT == 1 ? 512 : CONV_THREADS;
    constexpr int lds = conv_lds_bytes<NI, TW, VEC, epi_taps(EPI), ONEKB, NT, false>();
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_mfma<NI, TW, EPI, VEC, ONEKB, false, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3x3_mfma<NI, TW, EPI, VEC, ONEKB, false, SPLIT>), dim3(grid), dim3(NT), lds, st, a);
    return hipGetLastError();
}

// w8: the eight-wave instantiation (conv_mfma.h: W8) -- ConvA only (the image layer's ConvA at small launches; every other direct operator class lost its
// eight-wave / half-block / strip instantiations in round 6: with the Winograd forms as the default nothing at any BASELINE shape ran them)
template <int NI, int TW, int EPI> static hipError_t launch_inst(const ConvArgs& a, int grid, hipStream_t st, bool vec, int w8 = 0)
{
    if constexpr (EPI == EPI_CONVA && TW == 16 && NI < 4) {   // (16-wide tiles only: the image layer's ConvA never runs on 8 x 8 tiles at a reference shape; a 64-column ConvA is a Winograd operator)
        if (w8 && vec) return launch_inst2<NI, TW, EPI, true, false, 1>(a, grid, st);
    }
    return vec ? launch_inst2<NI, TW, EPI, true>(a, grid, st) : launch_inst2<NI, TW, EPI, false>(a, grid, st);
}

template <int EPI> static hipError_t launch_epi(int NI, int TW, const ConvArgs& a, int grid, hipStream_t st, bool vec, int w8 = 0)
{
    if (TW == 16) {
        switch (NI) {
            case 1: return launch_inst<1, 16, EPI>(a, grid, st, vec, w8);
            case 2: return launch_inst<2, 16, EPI>(a, grid, st, vec, w8);
            case 3: return launch_inst<3, 16, EPI>(a, grid, st, vec, w8);
            default: return launch_inst<4, 16, EPI>(a, grid, st, vec, w8);
        }
    }
    switch (NI) {
        case 1: return launch_inst<1, 8, EPI>(a, grid, st, vec, w8);
        case 2: return launch_inst<2, 8, EPI>(a, grid, st, vec, w8);
        case 3: return launch_inst<3, 8, EPI>(a, grid, st, vec, w8);
        default: return launch_inst<4, 8, EPI>(a, grid, st, vec, w8);
    }
}

static hipError_t launch_conv(eigen_engine* e, ConvOp& op, ConvArgs& a, int batch, hipStream_t st)
{
    const int TH = (op.TW == 8) ? 8 : 16;
    const int NIMG = 256 / (TH * op.TW);
    a.H = op.H; a.W = op.W; a.B = batch;
    a.tilesX = (op.W + op.TW - 1) / op.TW;
    a.tilesY = (op.H + TH - 1) / TH;
    a.n_nblk = op.n_nblk; a.krows = op.krows; a.wpk = op.d_wpk; a.Cout = op.Cout; a.zeros = e->d_zeros;
    a.nsrc = op.nsrc;
    for (int s = 0; s < op.nsrc; ++s) { a.src[s].C = op.src_C[s]; a.src[s].Cpad = pad4(op.src_C[s]); a.src[s]._reserved = 0; a.src[s].Ct = op.src_Ct[s] ? op.src_Ct[s] : op.src_C[s]; }
    int ntile = ((batch + NIMG - 1) / NIMG) * a.tilesX * a.tilesY;
    const int per_tile = op.n_nblk * (op.epi == EPI_UP4 ? 4 : 1);
    int grid = per_tile * ((ntile + 7) / 8) * 8;  // XCD-aware tile map (conv_mfma.h): tiles padded to a multiple of 8
    // 16-byte DMA staging needs chunk-aligned rows: W % 4 == 0
    const bool vec = (op.W % 4) == 0;
#if EIG_TIMING
    unsigned long long* tl_dbg = nullptr;
    if (getenv("EIGEN_TIMELINE") && (op.epi == EPI_LSTM || (op.epi == EPI_UP4 && op.NI == 4) || (op.wino && getenv("EIGEN_TIMELINE_ALL"))) && ++op.tl_seen == 6) {  // a steady-state launch of every ConvLSTM op and 2x2-form pass
        (void)hipMalloc((void**)&tl_dbg, (size_t)grid * 2 * 64 * 8);  // half blocks double the grid, W8 blocks have 8 waves
        (void)hipMemset(tl_dbg, 0, (size_t)grid * 2 * 64 * 8);
        a.dbg = tl_dbg;
    }
#endif
    {
        // 0 only for A/B measurements: tiles interleaved over the XCDs instead of a contiguous tile range per XCD
        static const int tile_map = getenv("EIGEN_TILE_MAP") ? atoi(getenv("EIGEN_TILE_MAP")) : 1;
        a.tile_map = tile_map != 0;
    }
    // Eight-wave instantiation of the direct ConvA (conv_mfma.h: W8): launches of at most four rounds of the device's block slots gain 4-5 % from twice as many waves out
    // of the same few blocks (profiles/r03_b_ab_w8.txt); EIGEN_W8 = 0 / 1 forces it off / on (A/B measurements and the parity tests).
    static const int w8_env = getenv("EIGEN_W8") ? atoi(getenv("EIGEN_W8")) : -1;
    const int w8 = (vec && op.epi == EPI_CONVA && op.TW == 16 && op.NI < 4 && (w8_env >= 0 ? w8_env != 0 : grid <= 8 * e->n_cu)) ? 1 : 0;
    op.last_grid = grid; op.last_waves = (w8 == 1) ? 8 : 4;
    if (e->profile_convs) (void)hipEventRecord(e->pev0, st);
    hipError_t r;
    if (op.wino) {  // Winograd form: F(4x4, 3x3), conv_wino4.h
        if (op.epi == EPI_LSTM && a.acc_init != nullptr) return hipErrorInvalidConfiguration;   // (set_weights never pairs F(4x4) with a separate unpooled chain)
        // Block shape (conv_wino4.h): 16 rows x 32 columns, or 32 x 16 ("tall") where that covers the MAP with fewer blocks -- 80 x 60: 10 instead of 12, 40 x 30: 3
        // instead of 4 (the reference's 160 x 120); a function of the operator's map size alone, and the chains do not depend on it.  EIGEN_W4_TALL = 0 / 1 forces it (A/B, tests).
        static const int tall_env = getenv("EIGEN_W4_TALL") ? atoi(getenv("EIGEN_W4_TALL")) : -1;
        const bool tall = tall_env >= 0 ? tall_env != 0 : ((op.W + 15) / 16) * ((op.H + 31) / 32) < ((op.W + 31) / 32) * ((op.H + 15) / 16);
        a.tilesX = tall ? (op.W + 15) / 16 : (op.W + 31) / 32; a.tilesY = tall ? (op.H + 31) / 32 : (op.H + 15) / 16;
        // Half blocks (conv_wino4.h: HALF, 8 x 32 pixels, six or twelve waves) while even THEY are at most one block per CU: the launch's time is then ONE block's time, and a half
        // block has the CU's matrix pipe to itself for half the multiply-adds (c1: +15 %; with more half blocks than CUs the second round costs more than the halving gains --
        // c2's 20 x 15 top layer, 200 full blocks: -7 %).  A choice by launch size, like the walk.  EIGEN_W4_HALF = 0 / 1 forces it (A/B, tests).
        static const int half_env = getenv("EIGEN_W4_HALF") ? atoi(getenv("EIGEN_W4_HALF")) : -1;
        const bool half = !tall && (half_env >= 0 ? half_env != 0 : (long long)op.n_nblk * batch * a.tilesX * ((op.H + 7) / 8) <= e->n_cu);
        if (half) a.tilesY = (op.H + 7) / 8;
        // Packed tiles (conv_wino4.h: PACK): maps of 4 x 4 or 5 x 4 tiles -- the 20 x 15 top layer of the reference's 160 x 120 fills 62 % of a wide block -- on half blocks
        // whose sixteen MFMA rows are all real tiles: tile columns 0-3 of one image, or tile column 4 of four images (five blocks per four images).  For ConvLSTMs without
        // an unpooled source and ConvPs.  Taken when its rounds of half blocks (a half block takes about two thirds of a full one's time) cost less than the rounds of
        // full blocks: ref160's ConvLSTM_3, 600 blocks = 3 rounds -> 756 half blocks = 3 rounds of two thirds; configs[1]'s, 200 blocks -> 252 half blocks, one round each.  A choice
        // by map and launch size; the chains do not depend on it.  EIGEN_W4_PACK = 0 / 1 forbids / forces it for every operator it can run.
        static const int pack_env = getenv("EIGEN_W4_PACK") ? atoi(getenv("EIGEN_W4_PACK")) : -1;
        const int ptx = (op.W + 3) / 4, pty = (op.H + 3) / 4;
        const bool pack_can = op.epi != EPI_CONVA && a.up_src == nullptr && (ptx == 4 || ptx == 5) && pty == 4 && tall_env < 0 && half_env < 0;
        bool pack = false;
        if (pack_can) {
            const long long nhalf = (long long)op.n_nblk * (batch + (ptx == 5 ? (batch + 3) / 4 : 0)), nfull = (long long)op.n_nblk * batch;
            pack = pack_env >= 0 ? pack_env != 0 : 2 * ((nhalf + e->n_cu - 1) / e->n_cu) < 3 * ((nfull + e->n_cu - 1) / e->n_cu);
        }
        if (pack) { a.tilesX = ptx; a.tilesY = pty; }
        const int ntile4 = pack ? batch + (ptx == 5 ? (batch + 3) / 4 : 0) : batch * a.tilesX * a.tilesY;
        // WALK (conv_wino4.h): nparts blocks per tile, each computing nwalk = n_nblk / nparts consecutive N-blocks of it: walks of three N-blocks where n_nblk allows, of
        // two otherwise (the blocks of a tile share its planes through the XCD's L2), no walk while the launch would not give every CU four blocks.  A property of the
        // launch only -- the bits do not depend on it.  EIGEN_W4_PARTS = n forces min(n, n_nblk) rounded down to a divisor (n >= n_nblk: one N-block per block), for A/B
        // measurements and the parity tests.
        static const int parts_env = getenv("EIGEN_W4_PARTS") ? atoi(getenv("EIGEN_W4_PARTS")) : 0;
        int nparts;
        if (parts_env > 0) { nparts = std::min(parts_env, op.n_nblk); while (op.n_nblk % nparts) --nparts; }
        else {   // walks of three N-blocks where n_nblk allows (two otherwise), shorter while the launch would not give every CU four blocks
            int nwalk = (op.n_nblk % 3 == 0) ? 3 : ((op.n_nblk % 2 == 0) ? 2 : 1);
            if ((long long)(op.n_nblk / nwalk) * ntile4 < 4ll * e->n_cu) nwalk = 1;
            nparts = op.n_nblk / nwalk;
        }
        if (tall || half || pack) nparts = op.n_nblk;   // (tall and half blocks do not walk)
        a.nparts = nparts; a.nwalk = op.n_nblk / nparts;
        const int g4 = nparts * ((ntile4 + 7) / 8) * 8;
        {   // q = umulhi(x, ceil(2^32 / d)) = x / d for every x with x * d < 2^32 (x < number of blocks here)
            auto magic = [&](long long d) -> unsigned { return (d > 1 && (long long)g4 * 16 * d < (1ll << 32)) ? (unsigned)(((1ll << 32) + d - 1) / d) : 0u; };   // (x 16: a packed block divides its first TILE's index)
            a.mg[0] = magic(nparts); a.mg[1] = magic((long long)a.tilesX * a.tilesY); a.mg[2] = magic(a.tilesX);
        }
        op.last_grid = g4 * a.nwalk; op.last_waves = ((half || pack) && !(op.NI == 4 && op.epi != EPI_CONVA)) ? W4_WAVES / 2 : W4_WAVES;   // (64-column ConvLSTM / ConvP half blocks: twelve waves, conv_wino4.h: NSPLIT)   // (timeline records: one per block and N-block of its walk)
#if EIG_TIMING
        if (tl_dbg) {   // sized from THIS launch's records (the buffer above was sized for the four-wave grid)
            (void)hipFree(tl_dbg);
            (void)hipMalloc((void**)&tl_dbg, (size_t)op.last_grid * op.last_waves * 64);
            (void)hipMemset(tl_dbg, 0, (size_t)op.last_grid * op.last_waves * 64);
            a.dbg = tl_dbg;
        }
#endif
        r = launch_wino4(op.NI, op.epi, pack ? W4_PACK : (tall ? W4_TALL : (half ? W4_HALF : W4_WIDE)), a, g4, st);
    } else {
    static const bool direct_p0 = !(getenv("EIGEN_CONVP0_MFMA") && atoi(getenv("EIGEN_CONVP0_MFMA")));  // A/B measurements only
    if (op.epi == EPI_CONVP && op.d_wraw && direct_p0) {  // image layer: HBM-bound, one thread per pixel (conv_mfma.h)
        const dim3 g((op.W + P0_TX - 1) / P0_TX, (op.H + P0_TY - 1) / P0_TY, batch);
        if (op.Cout == 3) hipLaunchKernelGGL(convp0_direct_kernel<3>, g, dim3(P0_TX * P0_TY), 0, st, a.src[0].ptr, op.d_wraw, a);
        else hipLaunchKernelGGL(convp0_direct_kernel<1>, g, dim3(P0_TX * P0_TY), 0, st, a.src[0].ptr, op.d_wraw, a);
        r = hipGetLastError();
    } else if (static const bool direct_l0 = !(getenv("EIGEN_LSTM0_MFMA") && atoi(getenv("EIGEN_LSTM0_MFMA")));  // A/B measurements only
               op.epi == EPI_LSTM_PACKED && op.d_wraw && direct_l0 && (op.Cout == 1 || op.Cout == 3)) {
        // image layer: one thread per pixel (conv_mfma.h: lstm0_direct_kernel); the step-0 operator has one source
        const dim3 g((op.W + L0_TX - 1) / L0_TX, (op.H + L0_TY - 1) / L0_TY, batch);
        const dim3 blk(L0_TX * L0_TY);
        const float *sE = a.src[0].ptr, *sH = a.src[1].ptr;
        if (op.nsrc == 1) {
            if (op.Cout == 3) hipLaunchKernelGGL((lstm0_direct_kernel<3, true>), g, blk, 0, st, sE, sH, op.d_wraw, a);
            else hipLaunchKernelGGL((lstm0_direct_kernel<1, true>), g, blk, 0, st, sE, sH, op.d_wraw, a);
        } else {
            if (op.Cout == 3) hipLaunchKernelGGL((lstm0_direct_kernel<3, false>), g, blk, 0, st, sE, sH, op.d_wraw, a);
            else hipLaunchKernelGGL((lstm0_direct_kernel<1, false>), g, blk, 0, st, sE, sH, op.d_wraw, a);
        }
        r = hipGetLastError();
    } else
    switch (op.epi) {
        case EPI_LSTM: r = (op.TW == 16) ? launch_inst<4, 16, EPI_LSTM>(a, grid, st, vec) : launch_inst<4, 8, EPI_LSTM>(a, grid, st, vec); break;
        case EPI_LSTM_PACKED: r = (op.TW == 16) ? launch_inst<1, 16, EPI_LSTM_PACKED>(a, grid, st, vec) : launch_inst<1, 8, EPI_LSTM_PACKED>(a, grid, st, vec); break;
        case EPI_CONVA: {
            // the image layer's ConvA (K = 9 x 6 channels): one K-block, its own instantiation (conv_mfma.h: ONEKB)
            static const bool onekb = !(getenv("EIGEN_NO_ONEKB") && atoi(getenv("EIGEN_NO_ONEKB")));  // A/B measurements only
            if (onekb && op.NI == 3 && op.TW == 16 && vec && op.nsrc == 1 && pad4(op.src_C[0]) <= KC) r = launch_inst2<3, 16, EPI_CONVA, true, true>(a, grid, st);
            else r = launch_epi<EPI_CONVA>(op.NI, op.TW, a, grid, st, vec, w8);
            break;
        }
        case EPI_CONVP: r = launch_epi<EPI_CONVP>(op.NI, op.TW, a, grid, st, vec, w8); break;
        case EPI_UP4: r = launch_epi<EPI_UP4>(op.NI, op.TW, a, grid, st, vec, w8); break;
        case EPI_UP4C: r = launch_inst2<4, 16, EPI_UP4C, true>(a, grid, st); break;  // chosen only for 16-wide tiles and 16-byte staging
        default: r = launch_epi<EPI_RAW>(op.NI, op.TW, a, grid, st, vec, w8); break;  // (eigen_test_conv)
    }
    }
#if EIG_TIMING
    if (tl_dbg) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)op.last_grid * op.last_waves * 8);
        (void)hipMemcpy(h.data(), tl_dbg, h.size() * 8, hipMemcpyDeviceToHost);
        char name[256];
        snprintf(name, sizeof(name), "%s/timeline_H%d_C%d%s.bin", getenv("EIGEN_TIMELINE"), op.H, op.Cout, op.epi == EPI_UP4 ? "_up4" : op.epi == EPI_CONVA ? "_convA" : op.epi == EPI_CONVP ? "_convP" : "");
        if (FILE* f = fopen(name, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        (void)hipFree(tl_dbg);
        a.dbg = nullptr;
    }
#endif
    if (e->profile_convs && r == hipSuccess) {
        (void)hipEventRecord(e->pev1, st);
        (void)hipEventSynchronize(e->pev1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e->pev0, e->pev1);
        op.ms += ms; op.launches++;
    }
    return r;
}


// ------------------------------------------------------------------------------------------------ ABI
extern "C" {

int eigen_abi_version(void) { return EIGEN_ABI_VERSION; }
int eigen_gate_order(void) { return EIG_GATE_ORDER; }
int eigen_winograd_mask(void) { return wino_mask_env(); }
const char* eigen_last_error(void) { return g_err.c_str(); }

void eigen_config_defaults(eigen_config* c)
{
    c->n_repeat = 20; c->n_ext = 2; c->requant_feedback = 0;
    c->lk_max_corners = 100; c->lk_block_size = 7; c->lk_win = 15; c->lk_max_level = 2; c->lk_max_iter = 10;
    c->flow_method = EIGEN_FLOW_LK;
    c->fb_levels = 3; c->fb_winsize = 15; c->fb_iterations = 3; c->fb_poly_n = 5; c->fb_step = 16; c->reserved1 = 0; c->fb_poly_sigma = 1.2;
    c->lk_quality_level = 0.3; c->lk_min_distance = 7.0; c->lk_epsilon = 0.03; c->lk_min_eig_thr = 1e-4;
}

int eigen_destroy(eigen_engine* e)
{
    if (!e) return EIGEN_OK;
    (void)hipSetDevice(e->cfg.device);
    for (int l = 0; l < e->L; ++l) {
        Layer& y = e->layer[l];
        float* ptrs[] = {y.h[0], y.h[1], y.c, y.P, y.E, y.bias_lstm, y.peep, y.biasA, y.biasP, y.convA.d_wpk, y.lstm.d_wpk, y.convP.d_wpk, y.convP.d_wraw, y.lstm.d_wraw, y.convA_t0.d_wpk, y.lstm_t0.d_wpk, y.up4.d_wpk};
        for (float* p : ptrs) if (p) (void)hipFree(p);
    }
    if (e->d_planes) (void)hipFree(e->d_planes);
    e->g_node_off.release(); e->g_edge_off.release(); e->g_edge_src.release(); e->g_out_node.release();
    e->g_node_act.release(); e->g_node_bias.release(); e->g_node_resp.release(); e->g_edge_w.release();
    void* misc[] = {e->d_images, e->d_frames, e->d_eig, e->d_cand, e->d_corners, e->d_next, e->d_vectors, e->d_status,
                    e->d_ncorners, e->d_counts, e->d_fitness, e->d_zeros, e->d_raw4,
                    e->fb_I, e->fb_R0, e->fb_R1, e->fb_M, e->fb_V, e->fb_flow[0], e->fb_flow[1]};
    for (void* p : misc) if (p) (void)hipFree(p);
    for (int i = 0; i < 2; ++i)
        for (int l = 0; l < FLOW_MAX_LEVELS; ++l) if (e->d_gray[i][l]) (void)hipFree(e->d_gray[i][l]);
    for (int l = 0; l < FLOW_MAX_LEVELS; ++l) if (e->d_deriv[l]) (void)hipFree(e->d_deriv[l]);
    if (e->side) (void)hipStreamDestroy(e->side);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (auto& ev : e->ev_join) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    if (e->pev0) (void)hipEventDestroy(e->pev0);
    if (e->pev1) (void)hipEventDestroy(e->pev1);
    delete e;
    return EIGEN_OK;
}

int eigen_create(const eigen_config* cfg, eigen_engine** out)
{
    if (!cfg || !out) return fail(EIGEN_ERR_INVALID, "null argument");
    *out = nullptr;
    const int L = cfg->n_layers;
    if (L < 1 || L > EIGEN_MAX_LAYERS) return fail(EIGEN_ERR_INVALID, "n_layers %d out of range", L);
    if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width % (1 << (L - 1))) || (cfg->height % (1 << (L - 1))))
        return fail(EIGEN_ERR_INVALID, "image %dx%d must be divisible by 2^(layers-1)=%d (2x2 pooling per PredNet layer)",
                    cfg->width, cfg->height, 1 << (L - 1));
    if (cfg->channels[0] != 1 && cfg->channels[0] != 3) return fail(EIGEN_ERR_INVALID, "channels[0] (c_dim) must be 1 or 3");
    if (cfg->max_batch < 1) return fail(EIGEN_ERR_INVALID, "max_batch must be >= 1");
    if (cfg->lk_max_corners < 1 || cfg->lk_max_corners > SCORE_T) return fail(EIGEN_ERR_INVALID, "lk_max_corners must be in 1..%d", SCORE_T);
    if (cfg->lk_win < 3 || cfg->lk_win > 16) return fail(EIGEN_ERR_INVALID, "lk_win must be in 3..16");
    if (cfg->lk_block_size < 1 || cfg->lk_block_size > EIG_MAXB) return fail(EIGEN_ERR_INVALID, "lk_block_size must be in 1..%d", EIG_MAXB);
    if (cfg->lk_max_level < 0 || cfg->lk_max_level >= FLOW_MAX_LEVELS) return fail(EIGEN_ERR_INVALID, "lk_max_level must be in 0..%d", FLOW_MAX_LEVELS - 1);
    if (cfg->n_repeat < 1 || cfg->n_ext < 0) return fail(EIGEN_ERR_INVALID, "n_repeat >= 1 and n_ext >= 0 required");
    if (cfg->flow_method != EIGEN_FLOW_LK && cfg->flow_method != EIGEN_FLOW_FARNEBACK) return fail(EIGEN_ERR_INVALID, "unknown flow_method %d", cfg->flow_method);
    if (cfg->flow_method == EIGEN_FLOW_FARNEBACK) {
        if (cfg->fb_poly_n < 1 || cfg->fb_poly_n > FB_MAX_POLY_N) return fail(EIGEN_ERR_INVALID, "fb_poly_n must be in 1..%d", FB_MAX_POLY_N);
        if (cfg->fb_winsize < 1 || !(cfg->fb_winsize & 1) || cfg->fb_winsize / 2 > FB_MAX_WIN_R) return fail(EIGEN_ERR_INVALID, "fb_winsize must be odd and <= %d", 2 * FB_MAX_WIN_R + 1);
        if (cfg->fb_levels < 0 || cfg->fb_levels > 4) return fail(EIGEN_ERR_INVALID, "fb_levels must be in 0..4 (Gaussian radius <= %d)", FB_MAX_BLUR_R);
        if (cfg->fb_iterations < 1 || cfg->fb_step < 1) return fail(EIGEN_ERR_INVALID, "fb_iterations >= 1 and fb_step >= 1 required");
        const int lv = fb_levels_used(cfg->height, cfg->width, cfg->fb_levels);
        if ((cfg->height % (1 << lv)) || (cfg->width % (1 << lv))) return fail(EIGEN_ERR_INVALID, "Farneback flow with %d pyramid levels needs an image divisible by %d", lv, 1 << lv);
    }
    for (int l = 0; l < L; ++l) if (cfg->channels[l] < 1) return fail(EIGEN_ERR_INVALID, "channels[%d] < 1", l);
    HIPCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(EIGEN_ERR_INVALID, "device %d is %s; this library is built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);

    eigen_engine* e = new eigen_engine();
    e->cfg = *cfg;
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e->L = L; e->B = cfg->max_batch; e->C0 = cfg->channels[0]; e->H = cfg->height; e->W = cfg->width; e->K = cfg->lk_max_corners;
    const size_t B = (size_t)e->B;
#define ALLOC(ptr, n)                                                                              \
    do {                                                                                           \
        hipError_t _e = hipMalloc((void**)&(ptr), (n));                                            \
        if (_e != hipSuccess) { eigen_destroy(e); return fail(EIGEN_ERR_HIP, "hipMalloc(%zu) for %s: %s", (size_t)(n), #ptr, hipGetErrorString(_e)); } \
    } while (0)
    for (int l = 0; l < L; ++l) {
        Layer& y = e->layer[l];
        y.C = cfg->channels[l]; y.H = e->H >> l; y.W = e->W >> l;
        const size_t n = B * y.C * y.H * y.W * sizeof(float);
        ALLOC(y.h[0], n); ALLOC(y.h[1], n); ALLOC(y.c, n); ALLOC(y.P, n); ALLOC(y.E, 2 * n);
    }
    const size_t HW = (size_t)e->H * e->W;
    ALLOC(e->d_images, B * e->C0 * HW);
    ALLOC(e->d_frames, B * 3 * e->C0 * HW);
    // flow pyramids: a level exists only while both dims stay > winSize (buildOpticalFlowPyramid)
    e->lvH[0] = e->H; e->lvW[0] = e->W; e->n_levels = 1;
    for (int l = 1; l <= cfg->lk_max_level; ++l) {
        const int hd = (e->lvH[l - 1] + 1) / 2, wd = (e->lvW[l - 1] + 1) / 2;
        if (wd <= cfg->lk_win || hd <= cfg->lk_win) break;
        e->lvH[l] = hd; e->lvW[l] = wd; e->n_levels = l + 1;
    }
    for (int l = 0; l < e->n_levels; ++l) {
        const size_t n = B * e->lvH[l] * e->lvW[l];
        ALLOC(e->d_gray[0][l], n); ALLOC(e->d_gray[1][l], n);
        ALLOC(e->d_deriv[l], n * sizeof(short2));
    }
    ALLOC(e->d_eig, B * HW * sizeof(float));
    ALLOC(e->d_cand, B * HW * sizeof(unsigned long long));
    ALLOC(e->d_corners, B * e->K * 2 * sizeof(float));
    ALLOC(e->d_next, B * e->K * 2 * sizeof(float));
    ALLOC(e->d_vectors, B * e->K * 4 * sizeof(float));
    ALLOC(e->d_status, B * e->K);
    ALLOC(e->d_ncorners, B * sizeof(int));
    ALLOC(e->d_counts, B * sizeof(int));
    ALLOC(e->d_fitness, B * sizeof(double));
    ALLOC(e->d_zeros, 256);
    (void)hipMemset(e->d_zeros, 0, 256);
#undef ALLOC
    for (auto& ev : e->ev) HIPCHK(hipEventCreate(&ev));
    HIPCHK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    for (int l = 0; l < L; ++l) HIPCHK(hipEventCreateWithFlags(&e->ev_join[l], hipEventDisableTiming));
    HIPCHK(hipEventCreate(&e->pev0));
    HIPCHK(hipEventCreate(&e->pev1));
    *out = e;
    return EIGEN_OK;
}

int eigen_set_prednet_weights(eigen_engine* e, const float* const* t, int32_t n_tensors)
{
    if (!e || !t) return fail(EIGEN_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    const int L = e->L;
    int expect = 0;
    for (int l = 0; l < L; ++l) expect += (l > 0 ? 2 : 0) + 2 + 4 * (l < L - 1 ? 4 : 3) + 3;
    if (n_tensors != expect) return fail(EIGEN_ERR_INVALID, "expected %d weight tensors for %d layers, got %d", expect, L, n_tensors);
    int k = 0;
    const int wino_env = wino_mask_env();
    auto upload = [&](float** dst, const float* src, size_t n) -> int {
        if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
        if (hipMalloc((void**)dst, n * sizeof(float)) != hipSuccess) return -1;
        return hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
    };
    for (int l = 0; l < L; ++l) {
        Layer& y = e->layer[l];
        const int C = y.C;
        const float *convA_w = nullptr, *convA_b = nullptr;
        if (l > 0) { convA_w = t[k++]; convA_b = t[k++]; }
        const float* convP_w = t[k++];
        const float* convP_b = t[k++];
        const float *wx0[4], *wx1[4] = {nullptr, nullptr, nullptr, nullptr}, *wh[4], *bh[4];
        for (int g = 0; g < 4; ++g) {
            wx0[g] = t[k++];
            if (l < L - 1) wx1[g] = t[k++];
            wh[g] = t[k++];
            bh[g] = t[k++];
        }
        const float* peep[3] = {t[k], t[k + 1], t[k + 2]};
        k += 3;
        for (int i = 0; i < k; ++i) if (!t[i]) return fail(EIGEN_ERR_INVALID, "weight tensor %d is NULL", i);

        // ---- ConvA_l: E_{l-1} (2 C_{l-1} ch at the finer resolution) -> C_l, fused relu / 2x2 max-pool / error unit
        if (l > 0) {
            ConvOp& op = y.convA;
            { float *k0 = op.d_wpk, *k1 = op.d_wraw; op = ConvOp(); op.d_wpk = k0; op.d_wraw = k1; }  // keep the allocations: upload() frees them
            op.epi = EPI_CONVA; op.layer = l; op.nsrc = 1; op.src_C[0] = 2 * e->layer[l - 1].C;
            op.H = e->layer[l - 1].H; op.W = e->layer[l - 1].W; op.Cout = C;
            choose_ni(C, false, &op.NI, &op.n_nblk);
            op.TW = choose_tw(op.H, op.W);
            op.krows = pad4(op.src_C[0]) * 9;
            op.macs = (double)op.H * op.W * C * op.src_C[0] * 9;
            const float* sw[3][4] = {{convA_w, nullptr, nullptr, nullptr}, {nullptr}, {nullptr}};
            std::vector<float> pk = pack_weights(op, sw, 0);
            if (upload(&op.d_wpk, pk.data(), pk.size()) || upload(&y.biasA, convA_b, C)) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvA%d)", l);
            ConvOp& t0 = y.convA_t0;  // step 0: first half of E_{l-1} only (Layer::convA_t0)
            { float* k0 = t0.d_wpk; t0 = op; t0.d_wpk = k0; t0.d_wraw = nullptr; }
            t0.src_C[0] = e->layer[l - 1].C; t0.src_Ct[0] = 2 * e->layer[l - 1].C;
            t0.krows = pad4(t0.src_C[0]) * 9;
            t0.macs = (double)op.H * op.W * C * t0.src_C[0] * 9;
            std::vector<float> pk0 = pack_weights(t0, sw, 0);
            if (upload(&t0.d_wpk, pk0.data(), pk0.size())) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvA%d, step 0)", l);
            if (wino_op(wino_env, 1, l, e->layer[l - 1].C, C, op.H, op.W, false)) {  // (the step-0 operator reads C_{l-1} channels: multiples of 8 too)
                const int ni = (C % 64) ? 3 : 4, nb = C / (16 * ni);
                const int sc[1] = {2 * e->layer[l - 1].C}, scw[1] = {2 * e->layer[l - 1].C}, sc0[1] = {e->layer[l - 1].C};
                const int wt = 4;   // F(4x4, 3x3) tiles
                std::vector<float> pw = pack_weights_wino(C, ni, nb, false, 1, sc, scw, sw);
                std::vector<float> pw0 = pack_weights_wino(C, ni, nb, false, 1, sc0, scw, sw);
                if (upload(&op.d_wpk, pw.data(), pw.size()) || upload(&t0.d_wpk, pw0.data(), pw0.size())) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvA%d, Winograd form)", l);
                for (ConvOp* f : {&op, &t0}) { f->wino = true; f->TW = 16; f->NI = ni; f->n_nblk = nb; }
                const double tl = (double)((op.H + wt - 1) / wt) * ((op.W + wt - 1) / wt) * (wt + 2) * (wt + 2);   // tiles x positions
                op.macs = tl * C * sc[0];
                t0.macs = tl * C * sc0[0];
            }
        }
        // ---- ConvLSTM_l: 4 gates fused on N.  Chain over E_l, h_l + chain of the unpooled R_{l+1} in its 2x2 form (own launch, Layer::up4)
        {
            ConvOp& op = y.lstm;
            { float *k0 = op.d_wpk, *k1 = op.d_wraw; op = ConvOp(); op.d_wpk = k0; op.d_wraw = k1; }  // keep the allocations: upload() frees them
            op.epi = EPI_LSTM; op.layer = l; op.H = y.H; op.W = y.W; op.Cout = C;
            op.nsrc = 2;
            op.src_C[0] = 2 * C; op.src_C[1] = C;
            choose_ni(C, true, &op.NI, &op.n_nblk);
            if (C <= 4) { op.epi = EPI_LSTM_PACKED; op.NI = 1; op.n_nblk = 1; }  // 4 gates x <=4 channels in one MFMA tile
            const int lstm_mode = (op.epi == EPI_LSTM_PACKED) ? 2 : 1;
            op.TW = choose_tw(op.H, op.W);
            op.krows = 0; op.macs = 0;
            for (int s = 0; s < op.nsrc; ++s) { op.krows += pad4(op.src_C[s]) * 9; op.macs += (double)y.H * y.W * 4 * C * op.src_C[s] * 9; }
            const float* sw[3][4];
            for (int g = 0; g < 4; ++g) { sw[0][g] = wx0[g]; sw[1][g] = wh[g]; sw[2][g] = nullptr; }
            std::vector<float> pk = pack_weights(op, sw, lstm_mode);
            std::vector<float> bias(4 * (size_t)C);
            for (int g = 0; g < 4; ++g) memcpy(&bias[(size_t)g * C], bh[g], sizeof(float) * C);
            const size_t chw = (size_t)C * y.H * y.W;
            std::vector<float> pp(3 * chw);
            for (int g = 0; g < 3; ++g) memcpy(&pp[g * chw], peep[g], sizeof(float) * chw);
            if (upload(&op.d_wpk, pk.data(), pk.size()) || upload(&y.bias_lstm, bias.data(), bias.size()) || upload(&y.peep, pp.data(), pp.size()))
                return fail(EIGEN_ERR_HIP, "weight upload failed (ConvLSTM%d)", l);
            if (op.epi == EPI_LSTM_PACKED && (C == 1 || C == 3)) {  // image layer, lstm0_direct_kernel: [C outputs][K taps = (channel, ky, kx) over E_0 then h_0][4 gates], then the step-0 table (first half of E_0 only)
                const int K = 3 * C * 9, K0 = C * 9;
                std::vector<float> raw((size_t)C * K * 4 + (size_t)C * K0 * 4);
                for (int o = 0; o < C; ++o)
                    for (int g = 0; g < 4; ++g) {
                        for (int c = 0; c < 2 * C; ++c)
                            for (int t9 = 0; t9 < 9; ++t9) {
                                const float wv = wx0[g][((size_t)o * 2 * C + c) * 9 + t9];
                                raw[((size_t)o * K + c * 9 + t9) * 4 + g] = wv;
                                if (c < C) raw[(size_t)C * K * 4 + ((size_t)o * K0 + c * 9 + t9) * 4 + g] = wv;
                            }
                        for (int c = 0; c < C; ++c)
                            for (int t9 = 0; t9 < 9; ++t9) raw[((size_t)o * K + (2 * C + c) * 9 + t9) * 4 + g] = wh[g][((size_t)o * C + c) * 9 + t9];
                    }
                if (upload(&op.d_wraw, raw.data(), raw.size())) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvLSTM%d direct)", l);
            }
            ConvOp& t0 = y.lstm_t0;  // step 0: first half of E_l only; h_l = 0 is not read (Layer::lstm_t0)
            { float* k0 = t0.d_wpk; t0 = op; t0.d_wpk = k0; }  // (d_wraw is shared with the full operator, which owns it)
            t0.nsrc = 1;
            t0.src_C[0] = C; t0.src_Ct[0] = 2 * C; t0.src_C[1] = 0;
            t0.krows = pad4(C) * 9; t0.macs = (double)y.H * y.W * 4 * C * C * 9;
            std::vector<float> pk0 = pack_weights(t0, sw, lstm_mode);
            if (upload(&t0.d_wpk, pk0.data(), pk0.size())) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvLSTM%d, step 0)", l);
            // The chain of the unpooled source R_{l+1}.  Direct ConvLSTM (the image layer, ineligible shapes, EIGEN_WINOGRAD=0): a pass of its own at the source resolution
            // in 2x2 form (EPI_UP4 / EPI_UP4C), added to the ConvLSTM's chain with one fp32 addition.  Winograd ConvLSTM: INSIDE the same chains, between E_l and h_l
            // (conv_wino4.h: up_fused; oracle/eig_oracle.c: eig_wino_lstm) -- below the top layer the Winograd form exists only that way (16-byte rows at the
            // source resolution: W % 8 == 0, 8-channel K-blocks: C_{l+1} % 8 == 0, bit 24 of the mask); an operator that cannot is a direct one.
            // EIGEN_WINOGRAD = bit mask of the operators that run in Winograd form (bit l ConvLSTM_l, 8 + l ConvA_l, 16 + l ConvP_l; bit 24: see above; bits 25-27: F(4x4, 3x3)
            // tiles) -- ANOTHER canonical summation order per setting, which the oracle follows through the same variable.  DEFAULT ON for every eligible operator
            // (EIGEN_WINO_DEFAULT); EIGEN_WINOGRAD=0 = the direct chains of rounds 1-3.  Every rank of a multi-GPU run must use the same value (eigen_winograd_mask).
            const bool wino_fuse = ((wino_env >> 24) & 1) && l < L - 1 && (y.W % 8) == 0 && (e->layer[l + 1].C % 8) == 0;
            const bool wino = op.epi == EPI_LSTM && wino_op(wino_env, 0, l, 3 * C, C, y.H, y.W, l == L - 1) && (l == L - 1 || wino_fuse);
            if (wino) {
                const int Cu = wino_fuse ? e->layer[l + 1].C : 0;
                const float* w3[3][4];
                for (int g = 0; g < 4; ++g) { w3[0][g] = wx0[g]; w3[1][g] = wino_fuse ? wx1[g] : wh[g]; w3[2][g] = wino_fuse ? wh[g] : nullptr; }
                const int sc[3] = {2 * C, wino_fuse ? Cu : C, C}, sw[3] = {2 * C, wino_fuse ? Cu : C, C};
                const int sc0[2] = {C, Cu}, sw0[2] = {2 * C, Cu};
                const int wt = 4;   // F(4x4, 3x3) tiles
                std::vector<float> pw = pack_weights_wino(C, 4, op.n_nblk, true, wino_fuse ? 3 : 2, sc, sw, w3);
                std::vector<float> pw0 = pack_weights_wino(C, 4, op.n_nblk, true, wino_fuse ? 2 : 1, sc0, sw0, w3);
                if (upload(&op.d_wpk, pw.data(), pw.size()) || upload(&t0.d_wpk, pw0.data(), pw0.size())) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvLSTM%d, Winograd form)", l);
                op.wino = t0.wino = true; op.TW = t0.TW = 16;
                const double tiles = (double)((y.H + wt - 1) / wt) * ((y.W + wt - 1) / wt);
                const double pf = 36, pu = 25;   // positions of a tile: full-resolution sources, the unpooled one
                op.macs = tiles * pf * 4 * C * (3.0 * C) + tiles * pu * 4 * C * Cu;   // executed: 16 / 36 (unpooled source: 9 / 25) multiply-adds per channel and tile
                t0.macs = tiles * pf * 4 * C * (1.0 * C) + tiles * pu * 4 * C * Cu;
                if (wino_fuse)
                    for (ConvOp* f : {&op, &t0}) { f->fused = true; f->up_C = Cu; f->up_kb = Cu / KC; }
            }
            ConvOp& u = y.up4;
            { float* k0 = u.d_wpk; u = ConvOp(); u.d_wpk = k0; }
            if (l < L - 1 && !wino) {  // R_{l+1}, at ITS resolution; columns = the ConvLSTM's
                const Layer& yu = e->layer[l + 1];
                u.epi = EPI_UP4; u.layer = l; u.nsrc = 1; u.src_C[0] = e->layer[l + 1].C; u.H = yu.H; u.W = yu.W; u.Cout = C;
                u.NI = op.NI; u.n_nblk = op.n_nblk; u.TW = choose_tw(u.H, u.W);
                u.krows = pad4(u.src_C[0]) * 4;
                u.macs = (double)y.H * y.W * 4 * C * u.src_C[0] * 4;  // 4 taps per output pixel and channel instead of 9
                const size_t need = (size_t)e->B * 4 * u.n_nblk * u.NI * 16 * u.H * u.W;
                // <= 16 columns (the packed image-layer ConvLSTM): all four classes in one block (EPI_UP4C) where the wide staging path exists
                static const bool up4c = !(getenv("EIGEN_NO_UP4C") && atoi(getenv("EIGEN_NO_UP4C")));  // A/B measurements only
                std::vector<float> pku;
                if (up4c && op.NI == 1 && lstm_mode == 2 && u.TW == 16 && (u.W % 4) == 0) {
                    u.epi = EPI_UP4C; u.NI = 4;
                    pku = pack_weights_up4c(u, wx1, lstm_mode);
                } else pku = pack_weights_up4(u, wx1, lstm_mode);
                if (upload(&u.d_wpk, pku.data(), pku.size())) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvLSTM%d, unpooled source)", l);
                if (need > e->raw4_floats) {
                    if (e->d_raw4) { (void)hipFree(e->d_raw4); e->d_raw4 = nullptr; e->raw4_floats = 0; }
                    if (hipMalloc((void**)&e->d_raw4, need * sizeof(float)) != hipSuccess) return fail(EIGEN_ERR_HIP, "hipMalloc(%zu) for the unpooled-source partial chains", need * sizeof(float));
                    e->raw4_floats = need;
                }
            }
        }
        // ---- ConvP_l
        {
            ConvOp& op = y.convP;
            { float *k0 = op.d_wpk, *k1 = op.d_wraw; op = ConvOp(); op.d_wpk = k0; op.d_wraw = k1; }  // keep the allocations: upload() frees them
            op.epi = EPI_CONVP; op.layer = l; op.nsrc = 1; op.src_C[0] = C;
            op.H = y.H; op.W = y.W; op.Cout = C;
            choose_ni(C, false, &op.NI, &op.n_nblk);
            op.TW = choose_tw(op.H, op.W);
            op.krows = pad4(C) * 9;
            op.macs = (double)y.H * y.W * C * C * 9;
            const float* sw[3][4] = {{convP_w, nullptr, nullptr, nullptr}, {nullptr}, {nullptr}};
            std::vector<float> pk = pack_weights(op, sw, 0);
            if (upload(&op.d_wpk, pk.data(), pk.size()) || upload(&y.biasP, convP_b, C)) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvP%d)", l);
            if (l == 0 && (C == 1 || C == 3) && upload(&op.d_wraw, convP_w, (size_t)C * C * 9)) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvP0 direct)");
            if (wino_op(wino_env, 2, l, C, C, y.H, y.W, l == L - 1)) {
                const int ni = (C % 64) ? 3 : 4, nb = C / (16 * ni);
                const int sc[1] = {C};
                const int wt = 4;
                std::vector<float> pw = pack_weights_wino(C, ni, nb, false, 1, sc, sc, sw);
                if (upload(&op.d_wpk, pw.data(), pw.size())) return fail(EIGEN_ERR_HIP, "weight upload failed (ConvP%d, Winograd form)", l);
                op.wino = true; op.TW = 16; op.NI = ni; op.n_nblk = nb;
                op.macs = (double)((y.H + wt - 1) / wt) * ((y.W + wt - 1) / wt) * (wt + 2) * (wt + 2) * C * C;
            }
        }
    }
    e->have_weights = true;
    return EIGEN_OK;
}

double eigen_prednet_flops_per_step(const eigen_engine* e)
{
    if (!e) return 0;
    double macs = 0;
    for (int l = 0; l < e->L; ++l) {
        const Layer& y = e->layer[l];
        const int C = y.C;
        const double hw = (double)y.H * y.W;
        if (l > 0) macs += (double)e->layer[l - 1].H * e->layer[l - 1].W * C * 2.0 * e->layer[l - 1].C * 9;
        double cin = 3.0 * C + (l < e->L - 1 ? e->layer[l + 1].C : 0);
        macs += hw * 4 * C * cin * 9;
        macs += hw * C * C * 9;
    }
    return 2.0 * macs;
}

int eigen_set_grid(eigen_engine* e, const double* h_planes, int32_t n_planes)
{
    if (!e || !h_planes) return fail(EIGEN_ERR_INVALID, "null argument");
    if (n_planes < 1 || n_planes > 4) return fail(EIGEN_ERR_INVALID, "n_planes must be 1..4");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t n = (size_t)n_planes * e->H * e->W;
    if (e->d_planes) { (void)hipFree(e->d_planes); e->d_planes = nullptr; }
    HIPCHK(hipMalloc((void**)&e->d_planes, n * sizeof(double)));
    HIPCHK(hipMemcpy(e->d_planes, h_planes, n * sizeof(double), hipMemcpyHostToDevice));
    e->n_planes = n_planes;
    e->have_grid = true;
    return EIGEN_OK;
}

static int render_cppn_impl(eigen_engine* e, const eigen_genome_batch* g, int32_t bg, int mode, uint8_t* d_images, double* d_nodes, void* stream)
{
    if (!e->have_grid) return fail(EIGEN_ERR_STATE, "eigen_set_grid has not been called");
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    const int G = g->n_genomes;
    if (G < 1) return fail(EIGEN_ERR_INVALID, "n_genomes < 1");
    const int need_out = (mode == 0) ? e->C0 : (mode == 4 ? 3 : 1);
    if (g->c_out < need_out) return fail(EIGEN_ERR_INVALID, "genome batch provides %d outputs per genome, %d needed", g->c_out, need_out);
    const int total_nodes = g->node_off[G];
    const int total_edges = g->edge_off[total_nodes];
    int max_nodes = 0, max_edges = 0;
    for (int i = 0; i < G; ++i) {
        const int nn = g->node_off[i + 1] - g->node_off[i];
        const int ne = g->edge_off[g->node_off[i + 1]] - g->edge_off[g->node_off[i]];
        if (nn < 1) return fail(EIGEN_ERR_INVALID, "genome %d has no nodes", i);
        max_nodes = std::max(max_nodes, nn); max_edges = std::max(max_edges, ne);
    }
    // every edge must point at a leaf, the constant-1 leaf, or an EARLIER node of the same genome (topological order)
    for (int gi = 0; gi < G; ++gi) {
        const int n0 = g->node_off[gi], n1 = g->node_off[gi + 1];
        for (int n = n0; n < n1; ++n)
            for (int k = g->edge_off[n]; k < g->edge_off[n + 1]; ++k) {
                const int src = g->edge_src[k];
                if (src < -(e->n_planes + 1))
                    return fail(EIGEN_ERR_INVALID, "genome %d: edge %d references leaf %d but only %d planes are set", gi, k, -src - 1, e->n_planes);
                if (src >= n - n0)
                    return fail(EIGEN_ERR_INVALID, "genome %d: edge %d of node %d reads node %d (not topologically earlier)", gi, k, n - n0, src);
            }
        for (int c = 0; c < g->c_out; ++c) {
            const int o = g->out_node[gi * g->c_out + c];
            if (o < 0 || o >= n1 - n0) return fail(EIGEN_ERR_INVALID, "genome %d: output %d names node %d of %d", gi, c, o, n1 - n0);
        }
    }
    const size_t lds = (size_t)max_nodes * CPPN_THREADS * 8 + (size_t)max_nodes * 16 + (size_t)max_edges * 8 + (size_t)(max_nodes + 1) * 4 + (size_t)max_edges * 4 + max_nodes + 64;
    if (lds > 160 * 1024) return fail(EIGEN_ERR_CAPACITY, "genome with %d nodes / %d edges needs %zu B of LDS (> 160 KiB)", max_nodes, max_edges, lds);
    if (e->g_node_off.ensure(G + 1) || e->g_edge_off.ensure(total_nodes + 1) || e->g_node_act.ensure(total_nodes) ||
        e->g_node_bias.ensure(total_nodes) || e->g_node_resp.ensure(total_nodes) || e->g_edge_src.ensure(std::max(total_edges, 1)) ||
        e->g_edge_w.ensure(std::max(total_edges, 1)) || e->g_out_node.ensure((size_t)G * g->c_out))
        return fail(EIGEN_ERR_HIP, "hipMalloc for the genome batch failed");
    HIPCHK(hipMemcpyAsync(e->g_node_off.p, g->node_off, sizeof(int32_t) * (G + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->g_edge_off.p, g->edge_off, sizeof(int32_t) * (total_nodes + 1), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->g_node_act.p, g->node_act, total_nodes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->g_node_bias.p, g->node_bias, sizeof(double) * total_nodes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->g_node_resp.p, g->node_resp, sizeof(double) * total_nodes, hipMemcpyHostToDevice, st));
    if (total_edges) {
        HIPCHK(hipMemcpyAsync(e->g_edge_src.p, g->edge_src, sizeof(int32_t) * total_edges, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->g_edge_w.p, g->edge_w, sizeof(double) * total_edges, hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipMemcpyAsync(e->g_out_node.p, g->out_node, sizeof(int32_t) * G * g->c_out, hipMemcpyHostToDevice, st));
    CppnArgs a;
    a.node_off = e->g_node_off.p; a.edge_off = e->g_edge_off.p; a.node_act = e->g_node_act.p;
    a.node_bias = e->g_node_bias.p; a.node_resp = e->g_node_resp.p; a.edge_src = e->g_edge_src.p; a.edge_w = e->g_edge_w.p;
    a.out_node = e->g_out_node.p; a.planes = e->d_planes; a.n_planes = e->n_planes; a.N = e->H * e->W;
    a.c_out = g->c_out; a.c_dim = e->C0; a.bg = bg; a.mode = mode; a.max_nodes = max_nodes; a.out = d_images; a.out_f64 = d_nodes;
    (void)hipFuncSetAttribute((const void*)cppn_render_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cppn_render_kernel, dim3((a.N + CPPN_THREADS - 1) / CPPN_THREADS, G), dim3(CPPN_THREADS), lds, st, a);
    HIPCHK(hipGetLastError());
    return EIGEN_OK;
}

int eigen_render_cppn(eigen_engine* e, const eigen_genome_batch* g, int32_t bg, int32_t gradient, uint8_t* d_images, void* stream)
{
    if (!e || !g || !d_images) return fail(EIGEN_ERR_INVALID, "null argument");
    if (gradient == 2) {  // get_equilum_image_from_cppn: three output nodes read as h, s, v
        if (e->C0 != 3) return fail(EIGEN_ERR_INVALID, "the h,s,v renderer needs c_dim = 3");
        if (g->c_out < 3) return fail(EIGEN_ERR_INVALID, "the h,s,v renderer needs 3 outputs per genome, the batch has %d", g->c_out);
        return render_cppn_impl(e, g, bg, 4, d_images, nullptr, stream);
    }
    if (gradient != 0 && gradient != 1) return fail(EIGEN_ERR_INVALID, "gradient must be 0, 1 or 2 (h,s,v renderer)");
    return render_cppn_impl(e, g, bg, (gradient == 1) ? 0 : (e->C0 == 1 ? 1 : 2), d_images, nullptr, stream);
}

int eigen_eval_cppn_nodes(eigen_engine* e, const eigen_genome_batch* g, double* d_nodes, void* stream)
{
    if (!e || !g || !d_nodes) return fail(EIGEN_ERR_INVALID, "null argument");
    return render_cppn_impl(e, g, 1, 3, nullptr, d_nodes, stream);
}

int eigen_prednet_rollout(eigen_engine* e, const uint8_t* d_images, int32_t batch, int32_t n_steps, int32_t first_out_step,
                          uint8_t* d_frames, void* stream)
{
    if (!e || !d_images || !d_frames) return fail(EIGEN_ERR_INVALID, "null argument");
    if (!e->have_weights) return fail(EIGEN_ERR_STATE, "eigen_set_prednet_weights has not been called");
    if (batch < 1 || batch > e->B) return fail(EIGEN_ERR_CAPACITY, "batch %d exceeds max_batch %d", batch, e->B);
    if (n_steps < 1 || n_steps > e->cfg.n_repeat + e->cfg.n_ext) return fail(EIGEN_ERR_INVALID, "n_steps %d out of range", n_steps);
    if (first_out_step < 0 || first_out_step >= n_steps) return fail(EIGEN_ERR_INVALID, "first_out_step %d out of range", first_out_step);
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    const int L = e->L;
    const size_t HW = (size_t)e->H * e->W;
    const int n_out = n_steps - first_out_step;
    for (int l = 0; l < L; ++l) {  // reset_state()
        Layer& y = e->layer[l];
        const size_t n = (size_t)batch * y.C * y.H * y.W * sizeof(float);
        HIPCHK(hipMemsetAsync(y.h[0], 0, n, st));
        HIPCHK(hipMemsetAsync(y.c, 0, n, st));
        HIPCHK(hipMemsetAsync(y.P, 0, n, st));
    }
    int cur = 0;  // h[cur] holds the state of the previous step
    // (One stream: the device is busy 99.9 % of a generation and the step's dependency chain is serial.  Round 3 / 4 measured a side stream for the off-chain
    // ConvP_l and two half-populations on two streams -- byte-identical, slower or equal at every shape: profiles/r03_b_ab_w8.txt, r04_b_ab_pipe2.txt; removed in round 5.)
    static const bool skip_zero_sources = !(getenv("EIGEN_NO_T0") && atoi(getenv("EIGEN_NO_T0")));  // A/B measurements only
    hipLaunchKernelGGL(e0_init_kernel, dim3(1024), dim3(256), 0, st, d_images, e->layer[0].E, e->C0, (int)HW, batch);
    HIPCHK(hipGetLastError());
    // ConvP_l (l > 0) on a side stream: pays where the launches are a fraction of a round of the chip -- configs[0] (pop 10 at 64 x 64: 10 to 40 blocks per launch) +7.5 %;
    // neutral at configs[1], -0.7 % at 160 x 120 colour pop 50, -2 % at the headline (profiles/r06_q_side_stream_ab.txt), as in round 3.  Hence: only while the layer-1 maps of
    // the batch are less than one block per CU.  A scheduling choice of the launch, not of the arithmetic.  EIGEN_SIDE_STREAM = 0 / 1 forces it (A/B, tests).
    static const int side_env = getenv("EIGEN_SIDE_STREAM") ? atoi(getenv("EIGEN_SIDE_STREAM")) : -1;
    const bool side_auto = L > 1 && (long long)batch * e->layer[1].H * e->layer[1].W < 512ll * e->n_cu;
    const bool side_on = (side_env >= 0 ? side_env != 0 : side_auto) && !e->profile_convs;
    // One PredNet step of genomes [b0, b0 + nb) on stream s; raw4: that range's partial-chain scratch.
    auto run_step = [&](int t, int b0, int nb, hipStream_t s, float* raw4) -> int {
        auto off = [&](float* p, const Layer& y, int mult = 1) { return p + (size_t)b0 * mult * y.C * y.H * y.W; };
        // bottom-up: E_l from E_{l-1} and the previous prediction P_l
        for (int l = 1; l < L; ++l) {
            Layer& y = e->layer[l];
            ConvArgs a;
            memset(&a, 0, sizeof(a));
            a.src[0].ptr = off(e->layer[l - 1].E, e->layer[l - 1], 2);
            a.bias = y.biasA; a.P = off(y.P, y); a.E = off(y.E, y, 2);
            if (side_on && t > 0) HIPCHK(hipStreamWaitEvent(s, e->ev_join[l], 0));   // P_l of the previous step came from the side stream
            HIPCHK(launch_conv(e, (t == 0 && skip_zero_sources) ? y.convA_t0 : y.convA, a, nb, s));
        }
        // top-down: R_l, then P_l
        for (int l = L - 1; l >= 0; --l) {
            Layer& y = e->layer[l];
            {
                ConvArgs a;
                memset(&a, 0, sizeof(a));
                int k = 0;
                if (l < L - 1 && y.lstm.fused) {  // R_{l+1} of THIS step: its chain runs inside the ConvLSTM launch
                    a.up_src = off(e->layer[l + 1].h[cur ^ 1], e->layer[l + 1]);
                    a.up_C = y.lstm.up_C; a.up_kb = y.lstm.up_kb;
                } else if (l < L - 1) {  // R_{l+1} of THIS step, 2x2 form -> partial chains
                    ConvArgs u;
                    memset(&u, 0, sizeof(u));
                    u.src[0].ptr = off(e->layer[l + 1].h[cur ^ 1], e->layer[l + 1]);
                    u.raw = raw4;
                    HIPCHK(launch_conv(e, y.up4, u, nb, s));
                    a.acc_init = raw4;
                }
                a.src[k++].ptr = off(y.E, y, 2);
                const bool t0 = (t == 0 && skip_zero_sources);
                if (!t0) a.src[k++].ptr = off(y.h[cur], y);
                a.bias = y.bias_lstm; a.c_state = off(y.c, y); a.h_out = off(y.h[cur ^ 1], y); a.peep = y.peep;
                HIPCHK(launch_conv(e, t0 ? y.lstm_t0 : y.lstm, a, nb, s));
            }
            // P_l (l > 0) is only read by ConvA_l of the NEXT step: nothing reads it after the last one
            if (l == 0 || t + 1 < n_steps) {
                ConvArgs a;
                memset(&a, 0, sizeof(a));
                a.src[0].ptr = off(y.h[cur ^ 1], y);
                a.bias = y.biasP; a.Pout = off(y.P, y); a.clip = (l == 0) ? 1 : 0;
                if (l == 0) {
                    if (t + 1 < n_steps) {  // error units of the next step
                        a.E0 = off(y.E, y, 2);
                        a.img = (t + 1 < e->cfg.n_repeat) ? d_images + (size_t)b0 * e->C0 * HW : nullptr;
                        a.requant = e->cfg.requant_feedback;
                    }
                    if (t >= first_out_step) {
                        a.frame_bstride = (long long)n_out * e->C0 * HW;
                        a.frame = d_frames + (size_t)b0 * a.frame_bstride + (size_t)(t - first_out_step) * e->C0 * HW;
                    }
                }
                if (side_on && l > 0) {   // fork: behind the ConvLSTM that produced R_l, beside everything that follows on the main stream
                    HIPCHK(hipEventRecord(e->ev_fork, s));
                    HIPCHK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
                    HIPCHK(launch_conv(e, y.convP, a, nb, e->side));
                    HIPCHK(hipEventRecord(e->ev_join[l], e->side));
                } else
                HIPCHK(launch_conv(e, y.convP, a, nb, s));
            }
        }
        return EIGEN_OK;
    };
    for (int t = 0; t < n_steps; ++t) {
        const int rc = run_step(t, 0, batch, st, e->d_raw4);
        if (rc) return rc;
        cur ^= 1;
    }
    e->hflip = cur;
    return EIGEN_OK;
}

// Dense Farneback flow between the gray images d_gray[0][0] -> d_gray[1][0], sampled into vectors (farneback_kernels.h)
static int farneback_flow(eigen_engine* e, int batch, float* d_vectors, int32_t* d_counts, hipStream_t st)
{
    const int H = e->H, W = e->W;
    const size_t HW = (size_t)H * W, B = (size_t)e->B;
    const eigen_config& c = e->cfg;
    if (!e->fb_I) {
        hipError_t r = hipMalloc((void**)&e->fb_I, B * HW * sizeof(float));
        float** five[] = {&e->fb_R0, &e->fb_R1, &e->fb_M, &e->fb_V};
        for (float** p : five) if (r == hipSuccess) r = hipMalloc((void**)p, B * 5 * HW * sizeof(float));
        for (int i = 0; i < 2 && r == hipSuccess; ++i) r = hipMalloc((void**)&e->fb_flow[i], B * 2 * HW * sizeof(float));
        if (r != hipSuccess) return fail(EIGEN_ERR_HIP, "hipMalloc for the Farneback workspaces: %s", hipGetErrorString(r));
    }
    const int levels = fb_levels_used(H, W, c.fb_levels);
    const FbConst pc = fb_poly_constants(c.fb_poly_n, c.fb_poly_sigma);
    const int m = c.fb_winsize / 2;
    int cur = 0;
    for (int k = levels; k >= 0; --k) {
        const int Hk = H >> k, Wk = W >> k, hw = Hk * Wk;
        const dim3 g1((hw + 255) / 256, batch);
        float* fl = e->fb_flow[cur];
        if (k == levels) HIPCHK(hipMemsetAsync(fl, 0, (size_t)batch * 2 * hw * sizeof(float), st));
        else hipLaunchKernelGGL(fb_upsample_kernel, g1, dim3(256), 0, st, e->fb_flow[cur ^ 1], H >> (k + 1), W >> (k + 1), fl, Hk, Wk);
        const FbBlur bl = fb_blur_kernel(k);
        const dim3 gp((Wk + FB_PX - 1) / FB_PX, (Hk + FB_PY - 1) / FB_PY, batch);
        for (int i = 0; i < 2; ++i) {
            hipLaunchKernelGGL(fb_blur_down_kernel, g1, dim3(256), 0, st, e->d_gray[i][0], H, W, k, bl, e->fb_I);
            hipLaunchKernelGGL(fb_polyexp_kernel, gp, dim3(FB_PX, FB_PY), 0, st, e->fb_I, Hk, Wk, pc, i == 0 ? e->fb_R0 : e->fb_R1);
        }
        hipLaunchKernelGGL(fb_update_matrices_kernel, g1, dim3(256), 0, st, e->fb_R0, e->fb_R1, fl, Hk, Wk, e->fb_M);
        for (int it = 0; it < c.fb_iterations; ++it) {
            hipLaunchKernelGGL(fb_box_v_kernel, g1, dim3(256), 0, st, e->fb_M, Hk, Wk, m, e->fb_V);
            hipLaunchKernelGGL(fb_box_h_solve_kernel, dim3((Wk + FB_HT - 1) / FB_HT, Hk, batch), dim3(FB_HT), 0, st, e->fb_V, Hk, Wk, m, fl);
            if (it + 1 < c.fb_iterations) hipLaunchKernelGGL(fb_update_matrices_kernel, g1, dim3(256), 0, st, e->fb_R0, e->fb_R1, fl, Hk, Wk, e->fb_M);
        }
        cur ^= 1;
    }
    const int step = fb_grid_step(H, W, c.fb_step, e->K);
    hipLaunchKernelGGL(fb_sample_kernel, dim3(batch), dim3(64), 0, st, e->fb_flow[cur ^ 1], H, W, step, e->K, d_vectors, d_counts);
    HIPCHK(hipGetLastError());
    return EIGEN_OK;
}

int eigen_flow(eigen_engine* e, const uint8_t* d_img0, int64_t stride0, const uint8_t* d_img1, int64_t stride1, int32_t batch,
               float* d_vectors, int32_t* d_counts, void* stream)
{
    if (!e || !d_img0 || !d_img1 || !d_vectors || !d_counts) return fail(EIGEN_ERR_INVALID, "null argument");
    if (batch < 1 || batch > e->B) return fail(EIGEN_ERR_CAPACITY, "batch %d exceeds max_batch %d", batch, e->B);
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    const int H = e->H, W = e->W, HW = H * W;
    const eigen_config& c = e->cfg;
    hipLaunchKernelGGL(gray_kernel, dim3((HW + 255) / 256, batch), dim3(256), 0, st, d_img0, (long long)stride0, e->C0, HW, e->d_gray[0][0], batch);
    hipLaunchKernelGGL(gray_kernel, dim3((HW + 255) / 256, batch), dim3(256), 0, st, d_img1, (long long)stride1, e->C0, HW, e->d_gray[1][0], batch);
    if (c.flow_method == EIGEN_FLOW_FARNEBACK) return farneback_flow(e, batch, d_vectors, d_counts, st);
    for (int l = 1; l < e->n_levels; ++l)
        for (int i = 0; i < 2; ++i)
            hipLaunchKernelGGL(pyrdown_kernel, dim3((e->lvH[l] * e->lvW[l] + 255) / 256, batch), dim3(256), 0, st, e->d_gray[i][l - 1],
                               e->lvH[l - 1], e->lvW[l - 1], e->d_gray[i][l], e->lvH[l], e->lvW[l]);
    for (int l = 0; l < e->n_levels; ++l)
        hipLaunchKernelGGL(scharr_kernel, dim3((e->lvH[l] * e->lvW[l] + 255) / 256, batch), dim3(256), 0, st, e->d_gray[0][l], e->lvH[l], e->lvW[l], e->d_deriv[l]);
    const double sc = 1.0 / ((double)(1 << 2) * c.lk_block_size * 255.0);
    const float SC = (float)(sc * sc);
    hipLaunchKernelGGL(mineig_kernel, dim3((W + EIG_T - 1) / EIG_T, (H + EIG_T - 1) / EIG_T, batch), dim3(256), 0, st, e->d_gray[0][0], H, W,
                       c.lk_block_size, SC, e->d_eig);
    hipLaunchKernelGGL(corner_select_kernel, dim3(batch), dim3(1024), 0, st, e->d_eig, H, W, c.lk_quality_level, (float)c.lk_min_distance,
                       c.lk_max_corners, e->d_cand, e->d_corners, e->d_ncorners);
    LKArgs a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l < e->n_levels; ++l) {
        a.I[l] = e->d_gray[0][l]; a.J[l] = e->d_gray[1][l]; a.dI[l] = e->d_deriv[l];
        a.Hs[l] = e->lvH[l]; a.Ws[l] = e->lvW[l];
    }
    a.max_level = e->n_levels - 1; a.win = c.lk_win; a.K = e->K;
    a.max_iter = std::min(std::max(c.lk_max_iter, 0), 100);
    const double eps = std::min(std::max(c.lk_epsilon, 0.0), 10.0);
    a.eps_sq = eps * eps; a.min_eig_thr = c.lk_min_eig_thr;
    a.corners = e->d_corners; a.ncorners = e->d_ncorners; a.next_pts = e->d_next; a.status = e->d_status;
    hipLaunchKernelGGL(lk_track_kernel, dim3(e->K, batch), dim3(64), 0, st, a);
    hipLaunchKernelGGL(compact_vectors_kernel, dim3(batch), dim3(128), 0, st, e->d_corners, e->d_next, e->d_status, e->d_ncorners, e->K, d_vectors, d_counts);
    HIPCHK(hipGetLastError());
    return EIGEN_OK;
}

int eigen_score(eigen_engine* e, int32_t structure, int32_t width, int32_t height, const float* d_vectors, const int32_t* d_counts, int32_t batch,
                double* d_fitness, void* stream)
{
    if (!e || !d_vectors || !d_counts || !d_fitness) return fail(EIGEN_ERR_INVALID, "null argument");
    if (structure < 0 || structure > EIGEN_SCORE_INSIDE_OUTSIDE)
        return fail(EIGEN_ERR_INVALID, "unknown structure %d (the reference raises NameError, generate_illusion.py:606-607)", structure);
    if (batch < 1) return fail(EIGEN_ERR_INVALID, "batch < 1");
    HIPCHK(hipSetDevice(e->cfg.device));
    ScoreArgs a;
    a.vectors = d_vectors; a.counts = d_counts; a.K = e->K; a.structure = structure; a.w = width > 0 ? width : e->W; a.h = height > 0 ? height : e->H; a.fitness = d_fitness;
    if (structure == EIGEN_SCORE_INSIDE_OUTSIDE) {
        const double step = (double)a.w / 5.0;
        const long cells = ((long)((double)a.w / step) + 1) * ((long)((double)a.h / step) + 1);
        if (cells > IO_MAX_CELLS) return fail(EIGEN_ERR_CAPACITY, "inside_outside_score: %ld cells of %dx%d exceed %d", cells, a.w, a.h, IO_MAX_CELLS);
        hipLaunchKernelGGL(inside_outside_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, a);
        HIPCHK(hipGetLastError());
        return EIGEN_OK;
    }
    hipLaunchKernelGGL(score_kernel, dim3(batch), dim3(SCORE_T), 0, (hipStream_t)stream, a);
    HIPCHK(hipGetLastError());
    return EIGEN_OK;
}

static int eval_images_impl(eigen_engine* e, const uint8_t* d_images, int batch, int structure, int pairing, double* h_fitness,
                            float* h_vectors, int32_t* h_counts, hipStream_t st, bool rendered)
{
    const size_t HW = (size_t)e->H * e->W, img_bytes = (size_t)e->C0 * HW;
    const int nr = e->cfg.n_repeat;
    int n_steps, first;
    if (pairing == EIGEN_PAIR_POPULATION) { n_steps = nr + 1; first = nr - 1; }  // prediction@n_repeat and 1st extension
    else { n_steps = nr + 2; first = nr + 1; }                                    // 2nd extension
    if (n_steps > nr + e->cfg.n_ext) return fail(EIGEN_ERR_INVALID, "pairing needs %d extension steps, engine has n_ext=%d", n_steps - nr, e->cfg.n_ext);
    if (!rendered) HIPCHK(hipEventRecord(e->ev[0], st));
    HIPCHK(hipEventRecord(e->ev[1], st));
    int rc = eigen_prednet_rollout(e, d_images, batch, n_steps, first, e->d_frames, st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->ev[2], st));
    const int n_out = n_steps - first;
    if (pairing == EIGEN_PAIR_POPULATION)
        rc = eigen_flow(e, e->d_frames, (int64_t)(n_out * img_bytes), e->d_frames + img_bytes, (int64_t)(n_out * img_bytes), batch, e->d_vectors, e->d_counts, st);
    else
        rc = eigen_flow(e, d_images, (int64_t)img_bytes, e->d_frames, (int64_t)(n_out * img_bytes), batch, e->d_vectors, e->d_counts, st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->ev[3], st));
    rc = eigen_score(e, structure, 0, 0, e->d_vectors, e->d_counts, batch, e->d_fitness, st);
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->ev[4], st));
    HIPCHK(hipMemcpyAsync(h_fitness, e->d_fitness, sizeof(double) * batch, hipMemcpyDeviceToHost, st));
    if (h_vectors) HIPCHK(hipMemcpyAsync(h_vectors, e->d_vectors, sizeof(float) * batch * e->K * 4, hipMemcpyDeviceToHost, st));
    if (h_counts) HIPCHK(hipMemcpyAsync(h_counts, e->d_counts, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[0], e->ev[1])); e->ms[0] = ms;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[1], e->ev[2])); e->ms[1] = ms;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[2], e->ev[3])); e->ms[2] = ms;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[3], e->ev[4])); e->ms[3] = ms;
    return EIGEN_OK;
}

int eigen_eval_population(eigen_engine* e, const eigen_genome_batch* g, int32_t structure, int32_t bg, int32_t gradient, int32_t pairing,
                          double* h_fitness, void* stream)
{
    if (!e || !g || !h_fitness) return fail(EIGEN_ERR_INVALID, "null argument");
    if (g->n_genomes < 1 || g->n_genomes > e->B) return fail(EIGEN_ERR_CAPACITY, "batch %d exceeds max_batch %d", g->n_genomes, e->B);
    if (structure < 0 || structure > 3) return fail(EIGEN_ERR_INVALID, "unknown structure %d", structure);
    HIPCHK(hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipEventRecord(e->ev[0], st));
    int rc = eigen_render_cppn(e, g, bg, gradient, e->d_images, st);
    if (rc) return rc;
    return eval_images_impl(e, e->d_images, g->n_genomes, structure, pairing, h_fitness, nullptr, nullptr, st, true);
}

int eigen_eval_images(eigen_engine* e, const uint8_t* d_images, int32_t batch, int32_t structure, int32_t pairing, double* h_fitness,
                      float* h_vectors, int32_t* h_counts, void* stream)
{
    if (!e || !d_images || !h_fitness) return fail(EIGEN_ERR_INVALID, "null argument");
    if (batch < 1 || batch > e->B) return fail(EIGEN_ERR_CAPACITY, "batch %d exceeds max_batch %d", batch, e->B);
    if (structure < 0 || structure > 3) return fail(EIGEN_ERR_INVALID, "unknown structure %d", structure);
    HIPCHK(hipSetDevice(e->cfg.device));
    return eval_images_impl(e, d_images, batch, structure, pairing, h_fitness, h_vectors, h_counts, (hipStream_t)stream, false);
}

int eigen_get_timings(eigen_engine* e, double* h_ms6)
{
    if (!e || !h_ms6) return fail(EIGEN_ERR_INVALID, "null argument");
    double conv = 0; int launches = 0;
    for (int l = 0; l < e->L; ++l) {
        conv += e->layer[l].convA.ms + e->layer[l].lstm.ms + e->layer[l].convP.ms + e->layer[l].convA_t0.ms + e->layer[l].lstm_t0.ms + e->layer[l].up4.ms;
        launches += e->layer[l].convA.launches + e->layer[l].lstm.launches + e->layer[l].convP.launches + e->layer[l].convA_t0.launches + e->layer[l].lstm_t0.launches + e->layer[l].up4.launches;
    }
    for (int i = 0; i < 4; ++i) h_ms6[i] = e->ms[i];
    h_ms6[4] = conv; h_ms6[5] = launches;
    return EIGEN_OK;
}

// Per-op profile of the roll-out convolutions.  enable=1 brackets every conv launch with HIP events on its stream
// (serialising the stream per launch); rows of h_out (8 doubles each, up to max_ops):
// [layer, epi (+16 for the step-0 operators), NI, TW, launches, total_ms, FLOPs per launch per image (2*MACs of the terms executed), n_nblk]
int eigen_conv_profile(eigen_engine* e, int32_t enable, int32_t reset, double* h_out, int32_t max_ops, int32_t* n_ops)
{
    if (!e) return fail(EIGEN_ERR_INVALID, "null argument");
    e->profile_convs = enable != 0;
    int n = 0;
    for (int l = 0; l < e->L; ++l) {
        ConvOp* ops[6] = {l > 0 ? &e->layer[l].convA : nullptr, &e->layer[l].lstm, &e->layer[l].convP,
                          l > 0 ? &e->layer[l].convA_t0 : nullptr, &e->layer[l].lstm_t0, l < e->L - 1 ? &e->layer[l].up4 : nullptr};
        for (ConvOp* op : ops) {
            if (!op) continue;
            if (h_out && n < max_ops) {
                double* r = h_out + (size_t)n * 8;
                r[0] = op->layer; r[1] = op->epi + ((op == &e->layer[l].convA_t0 || op == &e->layer[l].lstm_t0) ? 16 : 0) + (op->wino ? 32 : 0); r[2] = op->NI; r[3] = op->TW; r[4] = op->launches; r[5] = op->ms; r[6] = 2.0 * op->macs; r[7] = op->n_nblk;
            }
            if (reset) { op->ms = 0; op->launches = 0; }
            ++n;
        }
    }
    if (n_ops) *n_ops = n;
    return EIGEN_OK;
}

static int test_conv_impl(eigen_engine* e, int32_t n_src, const float* const* d_src, const int32_t* cin, const int32_t* up, const float* const* h_w,
                          int32_t cout, int32_t H, int32_t W, int32_t batch, float* d_out, void* stream, int iters, double* h_ms)
{
    if (!e || !d_src || !cin || !up || !h_w || !d_out) return fail(EIGEN_ERR_INVALID, "null argument");
    if (n_src < 1 || n_src > 3) return fail(EIGEN_ERR_INVALID, "n_src must be 1..3");
    HIPCHK(hipSetDevice(e->cfg.device));
    // canonical arithmetic: one chain over the full-resolution sources in list order, plus the chain of the
    // unpooled source in its 2x2 form (EPI_UP4 launch at the source resolution), one fp32 addition
    int n_up = 0, i_up = -1, n_full = 0;
    for (int s = 0; s < n_src; ++s) { if (up[s]) { ++n_up; i_up = s; } else ++n_full; }
    if (n_up > 1 || n_full < 1) return fail(EIGEN_ERR_INVALID, "at most one unpooled source and at least one full-resolution source");
    if (n_up && ((H | W) & 1)) return fail(EIGEN_ERR_INVALID, "an unpooled source needs even H and W");
    ConvOp op;
    op.epi = EPI_RAW; op.nsrc = 0; op.H = H; op.W = W; op.Cout = cout;
    choose_ni(cout, false, &op.NI, &op.n_nblk);
    op.TW = choose_tw(H, W);
    op.krows = 0;
    const float* sw[3][4] = {{nullptr}, {nullptr}, {nullptr}};
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    for (int s = 0; s < n_src; ++s) {
        if (up[s]) continue;
        op.src_C[op.nsrc] = cin[s]; op.krows += pad4(cin[s]) * 9; sw[op.nsrc][0] = h_w[s]; a.src[op.nsrc].ptr = d_src[s];
        op.nsrc++;
    }
    std::vector<float> pk = pack_weights(op, sw, 0);
    HIPCHK(hipMalloc((void**)&op.d_wpk, pk.size() * sizeof(float)));
    HIPCHK(hipMemcpy(op.d_wpk, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    a.raw = d_out;
    ConvOp u;
    ConvArgs ua;
    memset(&ua, 0, sizeof(ua));
    float* d_raw4 = nullptr;
    if (n_up) {
        u.epi = EPI_UP4; u.nsrc = 1; u.src_C[0] = cin[i_up]; u.H = H / 2; u.W = W / 2; u.Cout = cout;
        u.NI = op.NI; u.n_nblk = op.n_nblk; u.TW = choose_tw(u.H, u.W);
        u.krows = pad4(cin[i_up]) * 4;
        const float* uw[4] = {h_w[i_up], nullptr, nullptr, nullptr};
        std::vector<float> pku = pack_weights_up4(u, uw, 0);
        HIPCHK(hipMalloc((void**)&u.d_wpk, pku.size() * sizeof(float)));
        HIPCHK(hipMemcpy(u.d_wpk, pku.data(), pku.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&d_raw4, (size_t)batch * 4 * u.n_nblk * u.NI * 16 * u.H * u.W * sizeof(float)));
        ua.src[0].ptr = d_src[i_up];
        ua.raw = d_raw4;
        a.acc_init = d_raw4;
    }
    auto launch_both = [&]() -> hipError_t {
        if (n_up) { hipError_t ru = launch_conv(e, u, ua, batch, (hipStream_t)stream); if (ru != hipSuccess) return ru; }
        return launch_conv(e, op, a, batch, (hipStream_t)stream);
    };
    const bool prof = e->profile_convs;
    e->profile_convs = false;
    hipError_t r = launch_both();  // also the warm-up of a timed run
#if EIG_TIMING
    {
        // geometry of THIS operator's launches as launch_conv chose it for the warm-up above (tile shape, half blocks, eight-wave
        // blocks): ADVICE r3 -- the round-2 formula here no longer described the round-3 launches
        const int grid_ = op.last_grid, waves_ = op.last_waves;
        unsigned long long* dbg = nullptr;
        (void)hipMalloc((void**)&dbg, (size_t)grid_ * waves_ * 8 * 8);
        (void)hipMemset(dbg, 0, (size_t)grid_ * waves_ * 8 * 8);
        a.dbg = dbg;
        (void)launch_conv(e, op, a, batch, (hipStream_t)stream);
        (void)hipStreamSynchronize((hipStream_t)stream);
        std::vector<unsigned long long> h((size_t)grid_ * waves_ * 8);
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        double s4[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < h.size(); i += 8) { s4[0] += (double)h[i + 5]; s4[1] += (double)h[i + 6]; s4[2] += (double)h[i + 7]; s4[3] += (double)(h[i + 2] - h[i + 1]); }
        const double n = (double)grid_ * waves_;
        fprintf(stderr, "[EIG_TIMING] blocks=%d waves/block=%d per-wave cycles: mfma+dma-issue %.0f  vmcnt-wait %.0f  barrier %.0f  loop-total %.0f\n", grid_, waves_, s4[0] / n, s4[1] / n, s4[2] / n, s4[3] / n);
        a.dbg = nullptr;
        (void)hipFree(dbg);
    }
#endif
    if (iters > 0 && r == hipSuccess) {
        (void)hipEventRecord(e->pev0, (hipStream_t)stream);
        for (int i = 0; i < iters && r == hipSuccess; ++i) r = launch_both();
        (void)hipEventRecord(e->pev1, (hipStream_t)stream);
        (void)hipEventSynchronize(e->pev1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e->pev0, e->pev1);
        if (h_ms) *h_ms = ms / iters;
    }
    e->profile_convs = prof;
    hipError_t r2 = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(op.d_wpk);
    if (u.d_wpk) (void)hipFree(u.d_wpk);
    if (d_raw4) (void)hipFree(d_raw4);
    if (r != hipSuccess) return fail(EIGEN_ERR_HIP, "conv launch: %s", hipGetErrorString(r));
    if (r2 != hipSuccess) return fail(EIGEN_ERR_HIP, "conv sync: %s", hipGetErrorString(r2));
    return EIGEN_OK;
}

int eigen_test_conv(eigen_engine* e, int32_t n_src, const float* const* d_src, const int32_t* cin, const int32_t* up, const float* const* h_w,
                    int32_t cout, int32_t H, int32_t W, int32_t batch, float* d_out, void* stream)
{
    return test_conv_impl(e, n_src, d_src, cin, up, h_w, cout, H, W, batch, d_out, stream, 0, nullptr);
}

// Same launch repeated `iters` times between two HIP events on the launch stream: average kernel time in ms.
int eigen_time_conv(eigen_engine* e, int32_t n_src, const float* const* d_src, const int32_t* cin, const int32_t* up, const float* const* h_w,
                    int32_t cout, int32_t H, int32_t W, int32_t batch, float* d_out, int32_t iters, double* h_ms, void* stream)
{
    if (iters < 1 || !h_ms) return fail(EIGEN_ERR_INVALID, "iters >= 1 and h_ms required");
    return test_conv_impl(e, n_src, d_src, cin, up, h_w, cout, H, W, batch, d_out, stream, iters, h_ms);
}

// Host-only: the graph part of genome flattening (genome.py: _flatten_lists is the specification, statement by statement).
// Keys are mapped to dense local ids once per genome (sorted key table + binary search), after which every set / map of the
// specification is a flat array indexed by id and reused from genome to genome: ~3 us per genome instead of ~20 us with
// node-based hash containers (256 genomes: 5.3 -> 0.8 ms, scripts/host_overhead.py).
namespace {
struct FlattenScratch {
    std::vector<int> keys;                    // sorted unique keys of this genome (fallback when the key range is huge)
    std::vector<int> dmap, dstamp;            // direct table key - kmin -> id, valid where dstamp == epoch
    int epoch = 0, kmin = 0, n_ids = 0;
    bool direct = false;
    std::vector<int> leaf, is_out, required, seen, node_at, state, index, has_c32;
    std::vector<float> c32;
    std::vector<int> ci, co;                  // connection ends as ids
    std::vector<int> in_off, in_src;          // incoming (ALL connections), CSR by out id
    std::vector<int> ex_off, ex_src, ex_fill; // expressed connections, CSR by out id, genome.connections order kept
    std::vector<double> ex_w;
    std::vector<int> last, next, order;
    std::vector<std::pair<int, int>> stack;
    int id_of(int key)
    {
        if (!direct) return (int)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
        const int at = key - kmin;
        if (dstamp[at] != epoch) { dstamp[at] = epoch; dmap[at] = n_ids++; }
        return dmap[at];
    }
    // ids for one genome's keys: NEAT keys are small integers (inputs -1..-n, nodes 0..), so a direct table serves; a
    // sorted table + binary search is the fallback for pathological ranges
    int begin(const int32_t* in_keys, int n_in, const int32_t* out_keys, int n_out, const int32_t* nk, int n_nodes,
              const int32_t* ci_, const int32_t* co_, int nc)
    {
        int lo = 0, hi = 0;
        bool first = true;
        auto span = [&](const int32_t* p, int n) { for (int i = 0; i < n; ++i) { if (first) { lo = hi = p[i]; first = false; } lo = std::min(lo, (int)p[i]); hi = std::max(hi, (int)p[i]); } };
        span(in_keys, n_in); span(out_keys, n_out); span(nk, n_nodes); span(ci_, nc); span(co_, nc);
        const long range = (long)hi - lo + 1;
        direct = range <= (1L << 18);  // 2 MB of tables at most; NEAT keys grow by one per added node
        if (direct) {
            kmin = lo; n_ids = 0;
            if ((long)dstamp.size() < range) { dstamp.assign(range, 0); dmap.resize(range); epoch = 0; }
            if (++epoch == 0x7fffffff) { std::fill(dstamp.begin(), dstamp.end(), 0); epoch = 1; }
            auto reg = [&](const int32_t* p, int n) { for (int i = 0; i < n; ++i) id_of(p[i]); };
            reg(in_keys, n_in); reg(out_keys, n_out); reg(nk, n_nodes); reg(ci_, nc); reg(co_, nc);
            return n_ids;
        }
        keys.clear();
        keys.insert(keys.end(), in_keys, in_keys + n_in); keys.insert(keys.end(), out_keys, out_keys + n_out);
        keys.insert(keys.end(), nk, nk + n_nodes); keys.insert(keys.end(), ci_, ci_ + nc); keys.insert(keys.end(), co_, co_ + nc);
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        return (int)keys.size();
    }
};
}  // namespace

int eigen_flatten_genomes(int32_t G, int32_t n_in, const int32_t* in_keys, int32_t n_out, const int32_t* out_keys, const int32_t* conn_off,
                          const int32_t* conn_in, const int32_t* conn_out, const double* conn_w, const uint8_t* conn_en, const int32_t* node_off,
                          const int32_t* node_key, const uint8_t* node_act, const uint8_t* node_agg_sum, const double* node_bias,
                          const double* node_resp, int32_t cap_nodes, int32_t cap_edges, int32_t* o_node_off, int32_t* o_edge_off, uint8_t* o_act,
                          double* o_bias, double* o_resp, int32_t* o_edge_src, double* o_edge_w, int32_t* o_out_node, uint8_t* o_status)
{
    if (G < 0 || !in_keys || !out_keys || !conn_off || !node_off || !o_node_off || !o_edge_off || !o_status) return fail(EIGEN_ERR_INVALID, "null argument");
    static thread_local FlattenScratch S;
    const int ONE = -(n_in + 1);
    int nn = 0, ne = 0;  // nodes / edges emitted so far
    o_node_off[0] = 0;
    o_edge_off[0] = 0;
    for (int g = 0; g < G; ++g) {
        const int c0 = conn_off[g], c1 = conn_off[g + 1], m0 = node_off[g], m1 = node_off[g + 1];
        const int nc = c1 - c0;
        const int nn0 = nn, ne0 = ne;
        uint8_t status = 0;
        // dense ids
        const int N = S.begin(in_keys, n_in, out_keys, n_out, node_key + m0, m1 - m0, conn_in + c0, conn_out + c0, nc);
        S.leaf.assign(N, -1); S.is_out.assign(N, 0); S.required.assign(N, 0); S.seen.assign(N, 0); S.node_at.assign(N, -1);
        S.state.assign(N, 0); S.index.assign(N, -1); S.has_c32.assign(N, 0); S.c32.assign(N, 0.0f);
        for (int i = 0; i < n_in; ++i) S.leaf[S.id_of(in_keys[i])] = i;
        for (int m = m0; m < m1; ++m) S.node_at[S.id_of(node_key[m])] = m;   // a later duplicate key wins, as in a dict
        S.ci.resize(nc); S.co.resize(nc);
        for (int c = 0; c < nc; ++c) { S.ci[c] = S.id_of(conn_in[c0 + c]); S.co[c] = S.id_of(conn_out[c0 + c]); }
        // incoming lists over ALL connection keys
        S.in_off.assign(N + 1, 0);
        for (int c = 0; c < nc; ++c) ++S.in_off[S.co[c] + 1];
        for (int i = 0; i < N; ++i) S.in_off[i + 1] += S.in_off[i];
        S.in_src.resize(nc);
        S.ex_fill.assign(S.in_off.begin(), S.in_off.end() - 1);
        for (int c = 0; c < nc; ++c) S.in_src[S.ex_fill[S.co[c]]++] = S.ci[c];
        // required_for_output, layer by layer
        S.last.clear();
        for (int k = 0; k < n_out; ++k) {
            const int o = S.id_of(out_keys[k]);
            S.is_out[o] = 1;
            if (!S.seen[o]) { S.seen[o] = 1; S.required[o] = 1; S.last.push_back(o); }
        }
        for (;;) {
            S.next.clear();   // t: sources of connections into the layer added last that were not seen before
            for (int b : S.last)
                for (int j = S.in_off[b]; j < S.in_off[b + 1]; ++j) {
                    const int a = S.in_src[j];
                    if (!S.seen[a] && S.state[a] == 0) { S.state[a] = 1; S.next.push_back(a); }   // state doubles as "in t" here
                }
            if (S.next.empty()) break;
            bool any_layer = false;
            for (int x : S.next) if (S.leaf[x] < 0) any_layer = true;
            for (int x : S.next) S.state[x] = 0;
            if (!any_layer) break;
            for (int x : S.next) { if (S.leaf[x] < 0) S.required[x] = 1; S.seen[x] = 1; }
            S.last.swap(S.next);
        }
        // expressed connections, per destination, in genome.connections order
        S.ex_off.assign(N + 1, 0);
        for (int c = 0; c < nc; ++c) {
            if (!conn_en[c0 + c]) continue;
            const int i = S.ci[c], o = S.co[c];
            if ((!S.required[o] && !S.required[i]) || S.is_out[i]) continue;
            ++S.ex_off[o + 1];
        }
        for (int i = 0; i < N; ++i) S.ex_off[i + 1] += S.ex_off[i];
        const int nex = S.ex_off[N];
        S.ex_src.resize(nex); S.ex_w.resize(nex);
        S.ex_fill.assign(S.ex_off.begin(), S.ex_off.end() - 1);
        for (int c = 0; c < nc; ++c) {
            if (!conn_en[c0 + c]) continue;
            const int i = S.ci[c], o = S.co[c];
            if ((!S.required[o] && !S.required[i]) || S.is_out[i]) continue;
            const int at = S.ex_fill[o]++;
            S.ex_src[at] = i; S.ex_w[at] = conn_w[c0 + c];
        }
        // depth-first post-order from the outputs
        S.order.clear();
        for (int k = 0; k < n_out && !status; ++k) {
            S.stack.clear();
            S.stack.emplace_back(S.id_of(out_keys[k]), 0);
            while (!S.stack.empty() && !status) {
                const int n = S.stack.back().first, ci = S.stack.back().second;
                S.stack.pop_back();
                if (S.leaf[n] >= 0 || S.state[n] == 2) continue;
                const int cn = S.ex_off[n + 1] - S.ex_off[n];
                if (ci == 0) {
                    if (S.state[n] == 1) { status = 2; break; }  // cycle
                    S.state[n] = 1;
                }
                if (ci < cn) {
                    S.stack.emplace_back(n, ci + 1);
                    const int child = S.ex_src[S.ex_off[n] + ci];
                    if (S.leaf[child] < 0 && S.state[child] != 2) {
                        if (S.state[child] == 1) { status = 2; break; }  // cycle
                        S.stack.emplace_back(child, 0);
                    }
                } else {
                    S.state[n] = 2;
                    S.order.push_back(n);
                }
            }
        }
        auto emit_node = [&](uint8_t act, double bias, double resp) -> bool {
            if (nn >= cap_nodes) return false;
            o_act[nn] = act; o_bias[nn] = bias; o_resp[nn] = resp;
            return true;
        };
        auto emit_edge = [&](int src, double w) -> bool {
            if (ne >= cap_edges) return false;
            o_edge_src[ne] = src; o_edge_w[ne] = w; ++ne;
            return true;
        };
        bool full = false, any_c32 = false;
        for (size_t oi = 0; oi < S.order.size() && !status && !full; ++oi) {
            const int n = S.order[oi];
            const int m = S.node_at[n];
            if (m < 0) { status = 2; break; }
            const int e0 = S.ex_off[n], e1 = S.ex_off[n + 1];
            if ((!node_agg_sum[m] && e1 > e0) || node_act[m] == 255) { status = 2; break; }
            if (e1 == e0) { S.has_c32[n] = 1; S.c32[n] = (float)node_bias[m]; any_c32 = true; continue; }
            bool all_const = any_c32;
            if (all_const) for (int j = e0; j < e1; ++j) if (!S.has_c32[S.ex_src[j]]) { all_const = false; break; }
            if (all_const) { status = 1; break; }  // numpy float32 activation of a constant sub-graph: the caller's job
            S.index[n] = nn - nn0;
            if (!emit_node(node_act[m], node_bias[m], node_resp[m])) { full = true; break; }
            int lead = e0;
            if (any_c32) while (lead < e1 && S.has_c32[S.ex_src[lead]]) ++lead;
            if (lead > e0) {  // Python's sum(): the leading run of float32 constants accumulates in float32
                float pre = 0.0f;
                for (int j = e0; j < lead; ++j) {
                    const float t = (float)S.ex_w[j] * S.c32[S.ex_src[j]];
                    pre = j == e0 ? t : (float)(pre + t);
                }
                if (!emit_edge(ONE, (double)pre)) { full = true; break; }
            }
            for (int j = lead; j < e1 && !full; ++j) {
                const int i = S.ex_src[j];
                if (S.leaf[i] >= 0) full = !emit_edge(-(S.leaf[i] + 1), S.ex_w[j]);
                else if (S.has_c32[i]) full = !emit_edge(ONE, (double)((float)S.ex_w[j] * S.c32[i]));
                else full = !emit_edge(S.index[i], S.ex_w[j]);
            }
            if (full) break;
            ++nn;
            o_edge_off[nn] = ne;
        }
        for (int k = 0; k < n_out && !status && !full; ++k) {
            const int o = S.id_of(out_keys[k]);
            if (S.has_c32[o]) {  // constant output plane: identity(1 * (k * 1.0) + 0) == k
                S.index[o] = nn - nn0;
                if (!emit_node(EIGEN_ACT_IDENTITY, 0.0, 1.0) || !emit_edge(ONE, (double)S.c32[o])) { full = true; break; }
                ++nn;
                o_edge_off[nn] = ne;
            }
            if (S.index[o] < 0) { status = 2; break; }
            o_out_node[(size_t)g * n_out + k] = S.index[o];
        }
        if (full) return fail(EIGEN_ERR_CAPACITY, "eigen_flatten_genomes: output capacity (%d nodes, %d edges) exceeded at genome %d", cap_nodes, cap_edges, g);
        if (status) { nn = nn0; ne = ne0; }  // empty segment: the caller handles this genome
        o_status[g] = status;
        o_node_off[g + 1] = nn;
    }
    return EIGEN_OK;
}

int eigen_test_det_math(eigen_engine* e, const float* d_x, int32_t n, float* d_exp, float* d_sig, float* d_tanh, void* stream)
{
    if (!e || !d_x) return fail(EIGEN_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    hipLaunchKernelGGL(det_math_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_x, n, d_exp, d_sig, d_tanh);
    HIPCHK(hipGetLastError());
    return EIGEN_OK;
}

// Stage-level access for the parity tests: the dense field of the last Farneback eigen_flow call, float [batch][2][H][W] (dx, dy planes).
int eigen_debug_dense_flow(eigen_engine* e, int32_t batch, float* h_flow, void* stream)
{
    if (!e || !h_flow) return fail(EIGEN_ERR_INVALID, "null argument");
    if (e->cfg.flow_method != EIGEN_FLOW_FARNEBACK || !e->fb_I) return fail(EIGEN_ERR_STATE, "no Farneback flow has been computed by this engine");
    if (batch < 1 || batch > e->B) return fail(EIGEN_ERR_CAPACITY, "batch %d exceeds max_batch %d", batch, e->B);
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    const int levels = fb_levels_used(e->H, e->W, e->cfg.fb_levels);
    // level k writes fb_flow[(levels - k) & 1]; the full-resolution field is level 0's
    HIPCHK(hipMemcpy(h_flow, e->fb_flow[levels & 1], sizeof(float) * (size_t)batch * 2 * e->H * e->W, hipMemcpyDeviceToHost));
    return EIGEN_OK;
}

// Stage-level access for the parity tests: corner list of the last eigen_flow call.
int eigen_debug_corners(eigen_engine* e, int32_t batch, float* h_corners, int32_t* h_ncorners, float* h_next, uint8_t* h_status, void* stream)
{
    if (!e) return fail(EIGEN_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (h_corners) HIPCHK(hipMemcpy(h_corners, e->d_corners, sizeof(float) * batch * e->K * 2, hipMemcpyDeviceToHost));
    if (h_ncorners) HIPCHK(hipMemcpy(h_ncorners, e->d_ncorners, sizeof(int32_t) * batch, hipMemcpyDeviceToHost));
    if (h_next) HIPCHK(hipMemcpy(h_next, e->d_next, sizeof(float) * batch * e->K * 2, hipMemcpyDeviceToHost));
    if (h_status) HIPCHK(hipMemcpy(h_status, e->d_status, (size_t)batch * e->K, hipMemcpyDeviceToHost));
    return EIGEN_OK;
}

}  // extern "C"
