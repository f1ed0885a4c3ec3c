// MEASUREMENT VARIANT, NOT COMPILED INTO THE LIBRARY (round 6; profiles/r06_g_lstm0_fused.txt): the image layer's ConvLSTM with the chain of its unpooled source inside one
// launch, three builds (single role with plain staging 2.04 ms per launch; staging software-pipelined 1.67 ms; two roles -- eight MFMA waves beside eight VALU waves -- 1.72 ms)
// against 1.58 ms for the two launches it would replace (one-block 2x2 pass 0.79 + lstm0_direct_kernel 0.79).  Bit-exact in all three.  What it would take: the MFMA role
// staged and pipelined like conv3x3_mfma (LDS-DMA, immediate-offset gathers) -- its naive K-block here runs the matrix pipe at about a third of the pass's rate.
// Kept as the starting point; it was wired in through ConvArgs::up_src / up_C / up_wpk ([Cup padded to 8][4 taps][4 classes][16 columns]) from launch_conv's image-layer branch.
// ConvLSTM of the image layer WITH the chain of its unpooled source R_1 inside (round 6): what the one-block 2x2 pass (EPI_UP4C) + lstm0_direct_kernel do in two
// launches through a [B][4 classes][16][H/2][W/2] tensor of partial chains (0.8 GB written and read back per step at 256 x 256 colour, pop 256).  One block = 64 x 8
// output pixels = 32 x 4 pixels of R_1, SIXTEEN waves in two roles that run side by side -- the matrix pipe under one half of the block, the VALU under the other:
//   waves 8-15  (MFMA) wave m owns source-pixel row Y = m >> 1, columns 16 (m & 1) .. + 15 as the 16 rows of its MFMA tiles, one accumulator tile per parity class (py, px)
//               of the output pixels (2 Y + py, 2 X + px); an MFMA step = one channel of R_1 with its four taps (a, b) as k: A = R_1[c][Y + a - 1 + py][X + b - 1 + px]
//               gathered from the staged tile (zeros outside the map), B = the pre-summed 2x2-form weights of (class, channel, tap) for the 16 columns gate * 4 + output
//               channel -- the chain of conv3x3_mfma<4, 16, EPI_UP4C>: from 0 over (channel, a, b) ascending (oracle/eig_oracle.c: conv_up2x2_chain); K-blocks of 8
//               channels, K-block kb + 1 fetched into registers before the MFMAs of kb; at the end the accumulators go to LDS in pixel order;
//   waves 0-7   (VALU) one thread per pixel: the 4 x C chains over E_0 then h_0 as lstm0_direct_kernel runs them, one INPUT channel (nine staged values, all C x 4
//               chains) per unit of work, ceil(units / K-blocks) units between the two barriers of a K-block; then + the unpooled source's chain (one fp32 addition)
//               and the gate epilogue.
// 48 KB of LDS and <= 64 VGPRs: two blocks per CU.  Same chains in the same order: bit-identical (EIGEN_NO_L0FUSE=1 restores the two launches).
// wup: [Cup padded to 8][4 taps][4 classes][16 columns] (eigen_engine.hip: pack_weights_l0up).
constexpr int LF_TX = 64, LF_TY = 8, LF_PIX = LF_TX * LF_TY, LF_THREADS = 2 * LF_PIX;
constexpr int LF_RS = 36;                      // floats per staged row of R_1: columns X0 - 1 .. X0 + 32 (+ 2 of padding)
constexpr int LF_R1 = 8 * 6 * LF_RS;           // one K-block of R_1: [8 channels][rows Y0 - 1 .. Y0 + 4][36]
constexpr int LF_W = 8 * 4 * 64;               // one K-block of weights: [8 channels][4 taps][4 classes][16 columns]
constexpr int LF_ZP = LF_PIX + 1;              // floats per (channel, gate) plane of the exchanged chains (+ 1: the sixteen columns of a lane group on sixteen banks)
template <int C, bool T0>
__global__ void __launch_bounds__(LF_THREADS, 8) lstm0_fused_kernel(const float* __restrict__ srcE, const float* __restrict__ srcH, const float* __restrict__ wgt,
                                                                    const ConvArgs a)
{
    constexpr int CE = T0 ? C : 2 * C, CH = T0 ? 0 : C;
    constexpr int NU = CE + CH;                // units of VALU work: input channels
    constexpr int K = NU * 9;
    __shared__ float tile[NU][LF_TY + 2][LF_TX + 2];
    __shared__ __attribute__((aligned(16))) float stage[(LF_R1 + LF_W) > (4 * C * LF_ZP) ? (LF_R1 + LF_W) : (4 * C * LF_ZP)];   // K-block staging, then zup[C][4 gates][8 x 64 (+ 1)]
    const int tid = threadIdx.x;
    const bool mfma_role = __builtin_amdgcn_readfirstlane(tid >> 9) != 0;
    const int x0 = blockIdx.x * LF_TX, y0 = blockIdx.y * LF_TY, b = blockIdx.z;
    const int HW = a.H * a.W;
    constexpr int PLANE = (LF_TY + 2) * (LF_TX + 2);
    for (int i = tid; i < NU * PLANE; i += LF_THREADS) {
        const int c = i / PLANE;
        const int r = i - c * PLANE;
        const int yy = r / (LF_TX + 2), xx = r - yy * (LF_TX + 2);
        const int gy = y0 + yy - 1, gx = x0 + xx - 1;
        float v = 0.0f;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            v = (c < CE) ? srcE[((size_t)b * 2 * C + c) * HW + gy * a.W + gx] : srcH[((size_t)b * C + (c - CE)) * HW + gy * a.W + gx];
        tile[c][yy][xx] = v;
    }
    const int Hs = a.H >> 1, Ws = a.W >> 1;
    const int Y0 = y0 >> 1, X0 = x0 >> 1;
    const int nkb = (a.up_C + 7) >> 3;
    float* const r1 = stage;
    float* const wl = stage + LF_R1;
    float* const zup = stage;
    const int t = tid & (LF_PIX - 1);
    __syncthreads();   // the E_0 / h_0 tile is staged
    // The two roles are two separate code paths with the SAME barrier sequence (two per K-block, one behind the exchange): the registers of one role are not live in
    // the other (merged into one loop the kernel needed both sets at once: 29 VGPRs in scratch under the 64-register cap of two sixteen-wave blocks per CU).
    if (mfma_role) {
        const int lane = t & 63, m = t >> 6, q = lane >> 4, r = lane & 15;
        const int Yl = m >> 1, Xl = 16 * (m & 1) + r;          // this lane's source pixel (A operand row) inside the block
        const int ta = q >> 1, tb = q & 1;                      // this lane's tap (k index of the step)
        const float* const R1 = a.up_src + (size_t)b * a.up_C * Hs * Ws;
        const size_t kstride = (size_t)8 * Hs * Ws;
        // staging of a K-block: element i = t + 512 j of the [8][6][36] tile and one 16-byte piece of the weights per thread; the position is the same for every K-block
        // (the channel advances by 8); K-block kb + 1 is fetched into registers BEFORE the MFMAs of kb and written to LDS behind them
        int eoff[4], ech[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = t + LF_PIX * j;
            const int c = i / (6 * LF_RS);
            const int rr = i - c * (6 * LF_RS);
            const int yy = rr / LF_RS, xx = rr - yy * LF_RS;
            const int gy = Y0 - 1 + yy, gx = X0 - 1 + xx;
            const bool ok = i < LF_R1 && xx < 34 && gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
            ech[j] = ok ? c : (1 << 20);                   // (a channel index that is never < up_C: outside the map / the tile)
            eoff[j] = ok ? (c * Hs + gy) * Ws + gx : 0;
        }
        float sv[4];
        f32x4 sw;
        auto fetch = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sv[j] = (kb * 8 + ech[j] < a.up_C) ? R1[kb * kstride + eoff[j]] : 0.0f;
            sw = reinterpret_cast<const f32x4*>(a.up_wpk + (size_t)kb * LF_W)[t];
        };
        f32x4 acc[4];
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) acc[cls] = (f32x4){0.f, 0.f, 0.f, 0.f};
        fetch(0);
        for (int kb = 0; kb < nkb; ++kb) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (t + LF_PIX * j < LF_R1) r1[t + LF_PIX * j] = sv[j];
            reinterpret_cast<f32x4*>(wl)[t] = sw;
            __syncthreads();
            if (kb + 1 < nkb) fetch(kb + 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* const pc = r1 + (c * 6 + Yl + ta) * LF_RS + Xl + tb;
                const float* const wc = wl + (c * 4 + q) * 64 + r;
#pragma unroll
                for (int cls = 0; cls < 4; ++cls)
                    acc[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(pc[(cls >> 1) * LF_RS + (cls & 1)], wc[cls * 16], acc[cls], 0, 0, 0);
            }
            __syncthreads();   // (everyone is done with this K-block's operands)
        }
        // exchange: D[row = 4 q + e][column r = gate * 4 + channel] of class (py, px) -> zup[channel][gate][2 Yl + py][2 (16 (m & 1) + 4 q + e) + px]
        if ((r & 3) < C) {
            const int och = r & 3, g = r >> 2;
#pragma unroll
            for (int cls = 0; cls < 4; ++cls)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    zup[(och * 4 + g) * LF_ZP + (2 * Yl + (cls >> 1)) * LF_TX + 2 * (16 * (m & 1) + 4 * q + e) + (cls & 1)] = acc[cls][e];
        }
        __syncthreads();
        return;
    }
    // ---- VALU role: pixel (ty, tx) of the block, its 4 x C chains
    const int tx = t & (LF_TX - 1), ty = t / LF_TX;
    const f32x4* const w4 = reinterpret_cast<const f32x4*>(wgt) + (T0 ? C * (3 * C * 9) : 0);
    float z[C][4];
#pragma unroll
    for (int o = 0; o < C; ++o)
#pragma unroll
        for (int g = 0; g < 4; ++g) z[o][g] = 0.0f;
    auto unit = [&](int c) __attribute__((always_inline)) {   // input channel c: nine staged values into all 4 x C chains (every chain sees (c, ky, kx) ascending)
        const float* const tp = &tile[0][0][0] + (c * (LF_TY + 2) + ty) * (LF_TX + 2) + tx;
        float v[9];
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9) v[k9] = tp[(k9 / 3) * (LF_TX + 2) + (k9 % 3)];
#pragma unroll
        for (int o = 0; o < C; ++o) {
            const f32x4* const wc = w4 + o * K + c * 9;
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) {
                const f32x4 w = wc[k9];
#pragma unroll
                for (int g = 0; g < 4; ++g) z[o][g] = fmaf(v[k9], w[g], z[o][g]);
            }
        }
    };
    const int upk = (NU + nkb - 1) / nkb;   // units per K-block
    int u = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
#pragma unroll 1
        for (int n = 0; n < upk && u < NU; ++n, ++u) unit(u);
        __syncthreads();
    }
#pragma unroll 1
    for (; u < NU; ++u) unit(u);   // (fewer K-blocks than units: the rest)
    __syncthreads();
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy >= a.H || gx >= a.W) return;
    const int pix = gy * a.W + gx;
#pragma unroll
    for (int o = 0; o < C; ++o) {
#pragma unroll
        for (int g = 0; g < 4; ++g) z[o][g] = z[o][g] + zup[(o * 4 + g) * LF_ZP + ty * LF_TX + tx];
        const float bi = a.bias[o], bf = a.bias[C + o], bc = a.bias[2 * C + o], bo = a.bias[3 * C + o];
        const size_t cbase = ((size_t)b * C + o) * HW;
        const size_t pbase = (size_t)o * HW;
        const size_t pstride = (size_t)C * HW;
        const float cold = a.c_state[cbase + pix];
        float cnew, hnew;
        lstm_cell(z[o][0], z[o][1], z[o][2], z[o][3], bi, bf, bc, bo, cold, a.peep[pbase + pix], a.peep[pstride + pbase + pix], a.peep[2 * pstride + pbase + pix], cnew, hnew);
        a.c_state[cbase + pix] = cnew;
        a.h_out[cbase + pix] = hnew;
    }
}

