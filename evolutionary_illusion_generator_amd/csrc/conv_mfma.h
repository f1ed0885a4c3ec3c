// conv_mfma.h -- fused 3x3 convolution on the CDNA4 fp32 matrix cores (gfx950), implicit GEMM.
//
// Replaces the chainer Convolution2D calls inside chainer_prednet's PredNet/ConvLSTM (`PredNet/net.py`), which the
// reference reaches through test_prednet (/root/reference/generate_illusion.py:533-537).
//
// GEMM view of one launch:  M = pixels of a batch of images (block tile: 256 pixels = 16x16 of one image, or 8x8
// of four), N = output channels (block tile: NI x 16), K = sum over sources of 9 x channels.  Each of the 4
// waves of a block owns 64 pixels x NI*16 channels = 4 x NI accumulators of v_mfma_f32_16x16x4_f32.
//
// Canonical arithmetic: every output is ONE fp32 fma chain over k = (source, channel, ky, kx);
// the f32 MFMA is exactly such a chain in k order, so the result is bit-identical to oracle/eig_oracle.c.
// Channel counts are padded to multiples of 4 with zero weights AFTER the real channels of each source, which
// appends exact no-op terms (fma(a, 0, acc) == acc) and leaves the chain of real terms untouched.
//
// Per K-block (16 channels of one source) the block stages
//   - the input tile with its 1-pixel halo, [16][NIMG][TH+2][TW+2] fp32, through registers (zero fill at the image
//     border, 2x nearest unpooling folded into the gather for the R_{l+1} source), and
//   - the weight slab [144][NI*16] fp32 straight into LDS with global_load_lds (16 B per lane),
// then runs 36 MFMA steps whose A operand is gathered from the halo tile at (pixel + tap) -- im2col never exists
// in memory.  k advances by 4 per step, (channel, tap) = divmod(k, 9), so the per-lane LDS offset pattern has period
// 9 steps (= 4 channels): nine precomputed address registers + immediates, no address VALU in the loop.
//
// Row <-> pixel map of a 16-row MFMA sub-tile: 2 image rows x 8 columns; row r = 4q + reg covers
// (dy, dx) = (reg >> 1, 2q + (reg & 1)), so the four accumulator registers of a lane are one 2x2 pooling window
// (max-pool and the 2x2 patch stores of the epilogues stay inside the lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eig {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_RAW = 0, EPI_LSTM = 1, EPI_CONVA = 2, EPI_CONVP = 3, EPI_LSTM_PACKED = 4 };

struct ConvSrc {
    const float* ptr;  // [B][C][H>>up][W>>up]
    int C;             // real channels
    int Cpad;          // padded to a multiple of 4
    int up;            // 1: half resolution, nearest-unpooled x2 on the fly
    int _pad;
};

struct ConvArgs {
    ConvSrc src[3];
    int nsrc;
    int H, W, B;
    int tilesX, tilesY;
    int n_nblk;
    int krows;          // packed weight rows per N-block (sum of Cpad*9)
    const float* wpk;   // [n_nblk][krows][NB]
    int Cout;           // real output channels (per gate for the LSTM)
    int clip;           // ConvP: clipped_relu(., 1) instead of relu
    const float* bias;  // LSTM: [4][C]; ConvA / ConvP: [C]
    // EPI_LSTM
    float* c_state;     // [B][C][H][W] in place
    float* h_out;       // [B][C][H][W]
    const float* peep;  // [3][C][H][W]  (c_i, c_f, c_o)
    // EPI_CONVA
    const float* P;     // [B][C][H/2][W/2]
    float* E;           // [B][2C][H/2][W/2]
    // EPI_CONVP
    float* Pout;        // [B][C][H][W]
    const uint8_t* img; // layer 0 only: next input frame (uint8 [B][C][H][W]) or nullptr = feed the prediction back
    float* E0;          // layer 0 only: error units for the NEXT step, or nullptr
    uint8_t* frame;     // layer 0 only: quantised prediction out, or nullptr
    long long frame_bstride;
    int requant;
    int _pad2;
    // EPI_RAW
    float* raw;         // [B][Cout][H][W]
};

// ---- deterministic fp32 transcendental kernels: same operations, same order as oracle/eig_oracle.c ----
__device__ __forceinline__ float det_expf(float x)
{
    x = fminf(x, 80.0f);
    x = fmaxf(x, -80.0f);
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = fmaf(p, r2, r) + 1.0f;
    const float s = __int_as_float(((int)n + 127) << 23);
    return y * s;
}
__device__ __forceinline__ float det_sigmoidf(float x) { return 1.0f / (1.0f + det_expf(-x)); }
__device__ __forceinline__ float det_tanhf(float x)
{
    float ax = fabsf(x);
    if (ax < 0.625f) {
        const float z = x * x;
        float p = -5.70498872745e-3f;
        p = fmaf(p, z, 2.06390887954e-2f);
        p = fmaf(p, z, -5.37397155531e-2f);
        p = fmaf(p, z, 1.33314422036e-1f);
        p = fmaf(p, z, -3.33332819422e-1f);
        const float pz = p * z;
        return fmaf(pz, x, x);
    }
    ax = fminf(ax, 10.0f);
    const float t = det_expf(2.0f * ax);
    const float r = 1.0f - 2.0f / (t + 1.0f);
    return x < 0.0f ? -r : r;
}
__device__ __forceinline__ float relu_f(float v) { return v > 0.0f ? v : 0.0f; }

template <int TW> struct TileGeom {
    static constexpr int TH = (TW == 16) ? 16 : 8;
    static constexpr int NIMG = 256 / (TH * TW);
    static constexpr int S = TW + 2;
    static constexpr int PH = TH + 2;
    static constexpr int PLANE = NIMG * PH * S;  // floats per channel in LDS
};

constexpr int KC = 16;  // channels per K-block

template <int NI, int TW> constexpr int conv_lds_bytes() { return (KC * TileGeom<TW>::PLANE + KC * 9 * NI * 16) * 4; }

template <int NI, int TW, int EPI>
__global__ void __launch_bounds__(256, 2) conv3x3_mfma(const ConvArgs a)
{
    using G = TileGeom<TW>;
    constexpr int TH = G::TH, NIMG = G::NIMG, S = G::S, PH = G::PH, PLANE = G::PLANE;
    constexpr int NB = NI * 16;
    constexpr int NPOS_R = (PLANE + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const in_lds = lds;
    float* const w_lds = lds + KC * PLANE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;
    const int col = lane & 15;

    // ---- block -> (N-block, image group, tile); tile index fastest so that co-resident blocks share a weight slab
    const int tiles = a.tilesX * a.tilesY;
    const int ngroups = (a.B + NIMG - 1) / NIMG;
    int bid = blockIdx.x;
    const int nblk = bid / (ngroups * tiles);
    bid -= nblk * ngroups * tiles;
    const int bgrp = bid / tiles;
    const int t = bid - bgrp * tiles;
    const int tyi = t / a.tilesX, txi = t - tyi * a.tilesX;

    // ---- staging positions of this thread inside the haloed tile
    int pos_img[NPOS_R], pos_gy[NPOS_R], pos_gx[NPOS_R];
    bool pos_ok[NPOS_R];
#pragma unroll
    for (int r = 0; r < NPOS_R; ++r) {
        const int pos = tid + r * 256;
        const int img = pos / (PH * S);
        const int rem = pos - img * (PH * S);
        const int yy = rem / S, xx = rem - yy * S;
        pos_img[r] = bgrp * NIMG + img;
        pos_gy[r] = tyi * TH + yy - 1;
        pos_gx[r] = txi * TW + xx - 1;
        pos_ok[r] = (pos < PLANE) && (pos_img[r] < a.B) && pos_gy[r] >= 0 && pos_gy[r] < a.H && pos_gx[r] >= 0 && pos_gx[r] < a.W;
    }

    // ---- A-operand gather addresses: lane row r = lane&15 = 4*rq + rreg -> pixel (dy, dx); k-slot j = lane>>4
    int addrA[9];
    {
        const int r = col;
        const int dy = (r & 3) >> 1, dx = 2 * (r >> 2) + (r & 1);
        int base;
        if (TW == 16) base = (wv * 4 + dy) * S + dx;  // sub-tile mi adds (mi>>1)*2*S + (mi&1)*8
        else base = wv * PH * S + dy * S + dx;        // one image per wave; sub-tile mi adds mi*2*S
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int k = 4 * s + q;
            const int c = k / 9, tap = k - 9 * c;
            addrA[s] = base + c * PLANE + (tap / 3) * S + (tap % 3);
        }
    }
    const int boff = q * NB + col;

    f32x4 acc[4][NI];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* wslab = a.wpk + (size_t)nblk * a.krows * NB;

    for (int s = 0; s < a.nsrc; ++s) {
        const ConvSrc src = a.src[s];
        const int Hs = a.H >> src.up, Ws = a.W >> src.up;
        const int chs = Hs * Ws;
        const float* sp[NPOS_R];
#pragma unroll
        for (int r = 0; r < NPOS_R; ++r)
            sp[r] = src.ptr + (size_t)pos_img[r] * src.C * chs + (pos_ok[r] ? ((pos_gy[r] >> src.up) * Ws + (pos_gx[r] >> src.up)) : 0);

        for (int c0 = 0; c0 < src.Cpad; c0 += KC) {
            const int kc = min(KC, src.Cpad - c0);
            __syncthreads();  // everyone is done reading the previous K-block
            // weight slab -> LDS, direct (lane-linear image)
            {
                const int n16 = kc * 9 * (NB / 4);
                for (int base = wv * 64; base < n16; base += 256) {
                    const int ch = base + lane;
                    if (ch < n16)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wslab + (size_t)ch * 4),
                                                         (__attribute__((address_space(3))) void*)(w_lds + (size_t)base * 4), 16, 0, 0);
                }
            }
            // input tile with halo -> registers -> LDS
            {
                float v[KC][NPOS_R];
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int r = 0; r < NPOS_R; ++r) {
                        const bool ok = pos_ok[r] && (c < kc) && (c0 + c < src.C);
                        v[c][r] = ok ? sp[r][(size_t)(c0 + c) * chs] : 0.0f;
                    }
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int r = 0; r < NPOS_R; ++r)
                        if (c < kc && (r == 0 || tid + r * 256 < PLANE)) in_lds[c * PLANE + tid + r * 256] = v[c][r];
            }
            wslab += (size_t)kc * 9 * NB;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            const int nper = kc >> 2;
            for (int per = 0; per < nper; ++per) {
                const float* ap = in_lds + per * 4 * PLANE;
                const float* bp = w_lds + per * 36 * NB + boff;
#pragma unroll
                for (int st = 0; st < 9; ++st) {
                    float av[4], bv[NI];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int moff = (TW == 16) ? ((mi >> 1) * 2 * S + (mi & 1) * 8) : (mi * 2 * S);
                        av[mi] = ap[addrA[st] + moff];
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) bv[ni] = bp[st * 4 * NB + ni * 16];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        }
    }

    // ---------------------------------------------------------------- epilogue
    const int HW = a.H * a.W;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        int img, py0, px0;
        if (TW == 16) {
            const int sidx = wv * 4 + mi;
            img = 0; py0 = 2 * (sidx >> 1); px0 = 8 * (sidx & 1) + 2 * q;
        } else {
            img = wv; py0 = 2 * mi; px0 = 2 * q;
        }
        const int b = bgrp * NIMG + img;
        const int gy0 = tyi * TH + py0, gx0 = txi * TW + px0;
        if (b >= a.B || gy0 >= a.H || gx0 >= a.W) continue;

        if (EPI == EPI_RAW) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int o = nblk * NB + ni * 16 + col;
                if (o >= a.Cout) continue;
                float* dst = a.raw + ((size_t)b * a.Cout + o) * HW;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int gy = gy0 + (reg >> 1), gx = gx0 + (reg & 1);
                    if (gy < a.H && gx < a.W) dst[gy * a.W + gx] = acc[mi][ni][reg];
                }
            }
        } else if (EPI == EPI_LSTM) {
            const int ch = nblk * 16 + col;
            if (ch >= a.Cout) continue;
            const float bi = a.bias[ch], bf = a.bias[a.Cout + ch], bc = a.bias[2 * a.Cout + ch], bo = a.bias[3 * a.Cout + ch];
            const size_t cbase = ((size_t)b * a.Cout + ch) * HW;
            const size_t pbase = (size_t)ch * HW;
            const size_t pstride = (size_t)a.Cout * HW;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gy = gy0 + (reg >> 1), gx = gx0 + (reg & 1);
                if (gy >= a.H || gx >= a.W) continue;
                const int pix = gy * a.W + gx;
                const float cold = a.c_state[cbase + pix];
                float zi = acc[mi][0][reg] + bi; zi = fmaf(a.peep[pbase + pix], cold, zi);
                float zf = acc[mi][1][reg] + bf; zf = fmaf(a.peep[pstride + pbase + pix], cold, zf);
                const float zc = acc[mi][2][reg] + bc;
                float zo = acc[mi][3][reg] + bo; zo = fmaf(a.peep[2 * pstride + pbase + pix], cold, zo);
                const float ii = det_sigmoidf(zi), ff = det_sigmoidf(zf), gg = det_tanhf(zc), oo = det_sigmoidf(zo);
                const float gi = gg * ii;
                const float cnew = fmaf(ff, cold, gi);
                a.c_state[cbase + pix] = cnew;
                a.h_out[cbase + pix] = oo * det_tanhf(cnew);
            }
        } else if (EPI == EPI_CONVA) {
            const int Ho = a.H >> 1, Wo = a.W >> 1;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int ch = nblk * NB + ni * 16 + col;
                if (ch >= a.Cout) continue;
                const float bb = a.bias[ch];
                const float v0 = relu_f(acc[mi][ni][0] + bb), v1 = relu_f(acc[mi][ni][1] + bb);
                const float v2 = relu_f(acc[mi][ni][2] + bb), v3 = relu_f(acc[mi][ni][3] + bb);
                const float A = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                const size_t o = (((size_t)b * a.Cout + ch) * Ho + (gy0 >> 1)) * Wo + (gx0 >> 1);
                const float p = a.P[o];
                const size_t e = (((size_t)b * 2 * a.Cout + ch) * Ho + (gy0 >> 1)) * Wo + (gx0 >> 1);
                a.E[e] = relu_f(A - p);
                a.E[e + (size_t)a.Cout * Ho * Wo] = relu_f(p - A);
            }
        } else if (EPI == EPI_CONVP) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int ch = nblk * NB + ni * 16 + col;
                if (ch >= a.Cout) continue;
                const float bb = a.bias[ch];
                const size_t base = ((size_t)b * a.Cout + ch) * HW;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int gy = gy0 + (reg >> 1), gx = gx0 + (reg & 1);
                    if (gy >= a.H || gx >= a.W) continue;
                    const int pix = gy * a.W + gx;
                    float v = relu_f(acc[mi][ni][reg] + bb);
                    if (a.clip) v = fminf(v, 1.0f);
                    a.Pout[base + pix] = v;
                    if (a.frame) a.frame[(size_t)b * a.frame_bstride + (size_t)ch * HW + pix] = (uint8_t)(int)(v * 255.0f);
                    if (a.E0) {
                        float x;
                        if (a.img) x = (float)a.img[base + pix] / 255.0f;
                        else if (a.requant) x = (float)(uint8_t)(int)(v * 255.0f) / 255.0f;
                        else x = v;
                        const size_t e = ((size_t)b * 2 * a.Cout + ch) * HW + pix;
                        a.E0[e] = relu_f(x - v);
                        a.E0[e + (size_t)a.Cout * HW] = relu_f(v - x);
                    }
                }
            }
        }
    }
}

// E_0 for the first step: P_0 = 0  ->  E = [relu(x), relu(-x)] = [x, 0]
__global__ void e0_init_kernel(const uint8_t* img, float* E0, int C, int HW, int B)
{
    const size_t n = (size_t)B * C * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / ((size_t)C * HW);
        const size_t r = i - b * (size_t)C * HW;
        const float x = (float)img[i] / 255.0f;
        E0[b * 2 * C * HW + r] = relu_f(x - 0.0f);
        E0[b * 2 * C * HW + (size_t)C * HW + r] = relu_f(0.0f - x);
    }
}

__global__ void det_math_kernel(const float* x, int n, float* e, float* s, float* t)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (e) e[i] = det_expf(x[i]);
        if (s) s[i] = det_sigmoidf(x[i]);
        if (t) t[i] = det_tanhf(x[i]);
    }
}

}  // namespace eig
