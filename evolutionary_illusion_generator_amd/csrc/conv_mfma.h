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
// The ConvLSTM's unpooled source R_{l+1} heads the chain in its 2x2 form: the 3x3 window of an output pixel covers only
// 2x2 distinct pixels of the half-resolution source (which ones depends on the pixel's parity class), so the nine weights
// are pre-summed per class on the host and an EPI_UP4 launch AT THE SOURCE RESOLUTION runs 4 terms per channel instead of
// 9 for each class; the ConvLSTM launch that follows runs its own chain over E_l and h_l and adds the two (ConvArgs::acc_init:
// the four registers of a lane are the four classes of one source pixel; one fp32 addition, as chainer adds its convolutions).
// Channel counts are padded to multiples of 4 with zero weights AFTER the real channels of each source, which
// appends exact no-op terms (fma(a, 0, acc) == acc) and leaves the chain of real terms untouched.
//
// Per K-block (KC = 8 channels of one source) both operands travel global -> LDS by LDS-DMA (buffer_load ... lds,
// 16 B per lane; no staging registers, no ds_write pass) into the buffer NOT being computed on:
//   - the input tile with its halo as ALIGNED 16-byte chunks of the source rows, [KC][NIMG][TH+2][TW+8] fp32; buffer
//     descriptors make the DMA free of per-lane address arithmetic and turn out-of-image / padded-channel slots into
//     hardware zero fill;
//   - the weight slab [KC*9][16 channels][NI tiles] fp32 (a lane's NI values of a row are one ds_read_b128).
// The DMA instructions of K-block k+1 are interleaved with the KC*9/4 = 18 MFMA steps of K-block k; one barrier per
// K-block.  The A operand is gathered from the halo tile at (pixel + tap) -- im2col never exists in memory.  k advances
// by 4 per step, (channel, tap) = divmod(k, 9), so the per-lane LDS offset pattern has period 9 steps (= 4 channels):
// nine precomputed address registers + immediates, no address VALU in the loop.
// (Layers whose width is not a multiple of 4 use 4-byte flat-address DMA into [KC][NIMG][TH+2][TW+2]: VEC = false.)
// The wide instantiations (16x16 tiles, NI >= 3) stage BRANCH-FREE: a DMA is v_add_u32 ... clamp + s_add m0 + buffer_load,
// everything that must not be read (padded channels, rows past the slab, `no next K-block`) is pushed out of the
// descriptor's range by a saturating add -- every non-MFMA instruction in the loop costs matrix-pipe time.
//
// Row <-> pixel map ("class-major"): the four 16-row MFMA sub-tiles of a wave are the four PARITY CLASSES (y & 1, x & 1) of the
// wave's 64 pixels, sub-tile mi = class (mi >> 1, mi & 1).  16-wide tiles: the wave owns rows 4 wv .. 4 wv + 3 of the 16 x 16
// tile = 2 x 8 pooling windows; MFMA row r = 4 q + reg <-> window (wy, wx) = (reg >> 1, 2 q + (reg & 1)), pixel
// (4 wv + 2 wy + py, 2 wx + px).  8-wide tiles: the wave owns one 8 x 8 image tile, row r <-> window (q, reg).  Consequences:
//   - the four classes of a pooling window are the SAME register of the four sub-tile accumulators of one lane (max-pool in-lane);
//   - a lane owns 4 (2) image rows x 4 (8) CONTIGUOUS columns: every epilogue access is a 16-byte access of 4 pixels;
//   - the 16 pixels of one A-operand read sit on 16 distinct even (odd) LDS banks and the neighbouring tap kx + 1 of the other
//     half of the lane group on the odd (even) ones: the gather of two adjacent taps is conflict-free (it was 2-way with the
//     2-rows-x-8-columns sub-tiles of round 1: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.34);
//   - all 16 rows of an MFMA share their class, so the class-dependent 2x2-form weights of an unpooled source are an ordinary
//     B operand per sub-tile (what lets that chain run inside the ConvLSTM kernel).
//
// Block granularity (round 3, template parameter SPLIT): the tile above with four waves x four classes (SPLIT 0) is
// what launches that fill the chip many times over run.  A launch of a few hundred blocks takes as long as the CU that received
// ceil(blocks / 256) of them, so small launches get the same arithmetic in finer pieces: SPLIT 1 = EIGHT waves on the same tile (two
// classes per wave), SPLIT 2 = four-wave HALF blocks of two images (two classes per wave; 8 x 8 tiles, or -- TW = 4 -- strips of
// 4 columns x 16 rows for maps that square tiles cover badly, e.g. 20 x 15).  Which lane computes a pixel changes, its chain does not.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// s_setprio level of a wave in its PROLOGUE / EPILOGUE (0: none).  Same-box A/Bs at 256^2 colour, 256 genomes
// (profiles/r02_b_ab_fuseup.txt): prologue at 1 -> +0.4..0.5 % (a young wave's ~400 prologue instructions took 19K cycles beside
// the old partner wave's MFMA stream, scripts/timeline.py; outranking it gets the wave into its own K loop sooner), at 2 or 3 the
// same or less; the epilogue at 1 -> nothing; the K loop at 1 or 3 -> -0.5 % (EIG_KLOOP_PRIO).
#ifndef EIG_PRO_PRIO
#define EIG_PRO_PRIO 1
#endif
#ifndef EIG_EPI_PRIO
#define EIG_EPI_PRIO 0
#endif
#ifndef EIG_KLOOP_PRIO
#define EIG_KLOOP_PRIO 0  // s_setprio level of a wave while it is inside the K loop (0: none)
#endif
#ifndef EIG_S8
#define EIG_S8 16  // LDS row stride of 8-wide tiles (16-byte staging): 16 or 20, see TileGeom
#endif
#ifndef EIG_ABLATE
#define EIG_ABLATE 0  // measurement-only builds (scripts/ablate_conv.py): 1 = no staging after the first K-block, 2 = ConvLSTM gate math removed
#endif

namespace eig {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_RAW = 0, EPI_LSTM = 1, EPI_CONVA = 2, EPI_CONVP = 3, EPI_LSTM_PACKED = 4, EPI_UP4 = 5, EPI_UP4C = 6 };
constexpr int epi_taps(int epi) { return (epi == EPI_UP4 || epi == EPI_UP4C) ? 4 : 9; }

struct ConvSrc {
    const float* ptr;  // [B][Ct][H][W]
    int C;             // real channels
    int Cpad;          // padded to a multiple of 4
    int _reserved;     // (was: in-kernel x2 unpooling; unpooled sources are evaluated by an EPI_UP4 launch at their own resolution)
    int Ct;            // channels of the TENSOR (image stride); C < Ct reads only its first C channels (step-0 operators)
};

struct ConvArgs {
    ConvSrc src[3];
    int nsrc;
    int H, W, B;
    int tilesX, tilesY;
    int n_nblk;
    // conv_wino4.h: a block computes nwalk consecutive N-blocks of its tile, nparts = n_nblk / nwalk blocks per tile; mg = reciprocals ceil(2^32 / d) of the divisors
    // of the block -> (tile, part, image) map (0 = not exact for this launch's range: divide), d = nparts, tilesX * tilesY, tilesX
    int nwalk, nparts;
    unsigned mg[3];
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
    int tile_map;       // block -> tile order, see the kernel
    // EPI_RAW
    float* raw;         // [B][Cout][H][W];  EPI_UP4: [B][4 parity classes][n_nblk*NB][H][W];  EPI_UP4C: [B][4][n_nblk*16][H][W]
    // Chain of an unpooled source (written by an EPI_UP4 launch at HALF this resolution), [B][4][n_nblk*NB][H/2][W/2], added to
    // this launch's chain with one fp32 addition after the K loop; nullptr: none.
    const float* acc_init;
    // The unpooled source R_{l+1} ITSELF, for the operators that chain it inside their own launch: the Winograd ConvLSTMs (conv_wino4.h: in the same
    // chains, weights in wpk) -- no EPI_UP4 pass and no partial-chain tensor exist for them.
    const float* up_src;   // [B][up_C][H/2][W/2]
    int up_C;              // real channels of the unpooled source
    int up_kb;             // its K-blocks: ceil(up_C / KC)
    const float* zeros; // >= 64 zero bytes in device memory: DMA source for out-of-image / padded-channel positions
    unsigned long long* dbg;  // EIG_TIMING builds only: per-block cycle counters
};

// ---- deterministic fp32 transcendental kernels: same operations, same order as oracle/eig_oracle.c ----
__device__ __forceinline__ float det_expf_core(float x)   // x in [-80, 80] (det_expf clamps; det_tanhf calls it with 2 |x| in [1.25, 20], where the clamps are identities)
{
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
__device__ __forceinline__ float det_expf(float x)
{
    x = fminf(x, 80.0f);
    x = fmaxf(x, -80.0f);
    return det_expf_core(x);
}
// EIG_GATE_ORDER: element-wise order of the ConvLSTM gate epilogue (oracle/eig_oracle.c: lstm_cell states the
// same thing).  1 (default, round 4): the order chainer_prednet's ConvLSTM.__call__ evaluates, wherever that order is knowable --
// peephole `c_g(c) = W * c` as a ROUNDED PRODUCT added last, F.sigmoid as chainer's CPU forward computes it,
// tanh(x * 0.5) * 0.5 + 0.5, the cell update as two rounded products and one addition (cc = tanh(cc) * ii; cc += ff * c).
// 0: rounds 1-3 (fmaf peepholes and cell update, 1 / (1 + exp(-x))); kept for same-box A/B builds (scripts/ab_gate_order.sh).
#ifndef EIG_GATE_ORDER
#define EIG_GATE_ORDER 1
#endif
__device__ __forceinline__ float det_tanhf(float x);
#if EIG_GATE_ORDER
__device__ __forceinline__ float det_sigmoidf(float x) { return det_tanhf(x * 0.5f) * 0.5f + 0.5f; }
#else
__device__ __forceinline__ float det_sigmoidf(float x) { return 1.0f / (1.0f + det_expf(-x)); }
#endif
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
    const float t = det_expf_core(2.0f * ax);
    // 2 / (t + 1), t + 1 in [4.49, 4.9e8]: v_rcp_f32 and one fused refinement step instead of the ten-instruction IEEE division sequence -- the SAME fp32 number for every
    // fp32 divisor in [1, 1e9], and this function the same for ALL 2^32 inputs (exhaustive check on the device: scripts/fastdiv_check.hip, profiles/r06_o_fastdiv_check.txt);
    // the oracle keeps the plain division.  Five of these per ConvLSTM cell.
    const float d = t + 1.0f, rc = __builtin_amdgcn_rcpf(d);
    float q = 2.0f * rc;
    q = fmaf(fmaf(-d, q, 2.0f), rc, q);
    const float r = 1.0f - q;
    return x < 0.0f ? -r : r;
}
__device__ __forceinline__ float relu_f(float v) { return v > 0.0f ? v : 0.0f; }

// One ConvLSTM cell: z* = the four gates' convolution sums (chain over E_l, h_l + chain of the unpooled R_{l+1}), b* the h_*
// convolutions' biases, p* the peephole weights, cold the cell state.  -> new cell state, new hidden state.  (This translation
// unit is compiled with -ffp-contract=off: a * b + c below is a rounded product and an addition.)
__device__ __forceinline__ void lstm_cell(float zi_, float zf_, float zc_, float zo_, float bi, float bf, float bc, float bo,
                                          float cold, float pi_, float pf_, float po_, float& cn, float& hn)
{
#if EIG_GATE_ORDER
    const float zi = (zi_ + bi) + pi_ * cold;
    const float zf = (zf_ + bf) + pf_ * cold;
    const float zc = zc_ + bc;
    const float zo = (zo_ + bo) + po_ * cold;
    const float ii = det_sigmoidf(zi), ff = det_sigmoidf(zf), oo = det_sigmoidf(zo);
    float cc = det_tanhf(zc);
    cc = cc * ii;
    const float fc = ff * cold;
    cn = cc + fc;
    hn = oo * det_tanhf(cn);
#else
    float zi = zi_ + bi; zi = fmaf(pi_, cold, zi);
    float zf = zf_ + bf; zf = fmaf(pf_, cold, zf);
    const float zc = zc_ + bc;
    float zo = zo_ + bo; zo = fmaf(po_, cold, zo);
    const float ii = det_sigmoidf(zi), ff = det_sigmoidf(zf), gg = det_tanhf(zc), oo = det_sigmoidf(zo);
    const float gi = gg * ii;
    cn = fmaf(ff, cold, gi);
    hn = oo * det_tanhf(cn);
#endif
}

// LDS geometry of the haloed input tile.
//  VEC (every layer whose width is a multiple of 4 -- all real PredNet shapes): a row holds the ALIGNED 16-byte chunks
//  x0-4 .. x0+TW+3 of the source row, so the whole tile is moved by 16 B/lane DMA; output column px, tap kx sits at
//  LDS column px + kx + 3.  An unpooled (half-resolution) source is staged at ITS resolution (rows y0/2-1 ..,
//  chunks x0/2-4 ..) and the x2 nearest unpooling happens in the gather address ((py+ky-1)>>1, (px+kx-1)>>1).
//  !VEC (odd widths): rows of TW+2 floats at output resolution, 4 B/lane DMA, unpooling folded into the DMA source.
template <int TW, bool VEC, bool HALF = false> struct TileGeom {
    static constexpr int TH = (TW == 8) ? 8 : 16;            // 16 x 16 of one image, 8 x 8 or (TW = 4) 4 wide x 16 tall of several
    static constexpr int NIMG = HALF ? 2 : 256 / (TH * TW);  // HALF (SPLIT == 2, 8- and 4-wide tiles): a 128-pixel block of two images
    // row stride in floats.  VEC: the aligned chunks x0-4 .. x0+TW+3 (TW + 8 floats).  16-wide: S = 24, 2 S = 16 (mod 32): the two
    // window rows x 8 even columns of a class sub-tile cover 16 distinct banks.  8-wide: S = 16 puts the four window rows of a
    // sub-tile on the same banks (4-way conflict on the A gather); one more chunk per row (EIG_S8 = 20: 2 S = 8 mod 32) would
    // make it conflict-free but takes the block from 78 to 88 KB of LDS = ONE block per CU: measured 362 vs 396+ evals/s at
    // 160x120 colour, so the conflicts are the cheaper evil (scripts/ab_bench.sh).
    static constexpr int S = VEC ? (TW == 16 ? TW + 8 : (TW == 4 ? 12 : EIG_S8)) : TW + 2;  // 4-wide: chunks x0-4 .. x0+7
    static constexpr int XO = VEC ? 3 : 0;
    static constexpr int PH = TH + 2;
    static constexpr int PLANE = NIMG * PH * S;  // floats per channel in LDS
};

#ifndef EIG_DMA_EARLY
#define EIG_DMA_EARLY 0  // measurement builds: > 0 = DMA op j of the next K-block is issued at MFMA step j * EIG_DMA_EARLY instead of spread over all steps
#endif
#ifndef EIG_KC
#define EIG_KC 8
#endif
constexpr int KC = EIG_KC;  // channels per K-block

// Branch-free staging (the wide instantiations, VEC && TW == 16 && NI == 4): every lane of every DMA instruction is live --
// lanes without a slot read out of range (zeros) into padding -- so the input area is rounded up to whole 256-chunk rounds.
template <int NI, int TW, bool VEC> constexpr bool conv_fast_dma() { return VEC && TW == 16 && KC == 8 && NI >= 3; }  // narrow tiles: the padding would cost a block per CU
// weight area: NI == 4 keeps its exact 4.5 rounds (the half round is issued by all four waves, two of them repeating the
// other two); narrower slabs are rounded up to whole rounds, the extra lanes fetch the following rows into the padding
// (NT = threads per block: 256, or 512 for the eight-wave instantiations -- a round is NT 16-byte chunks)
template <int NI, int TW, bool VEC, int TAPS = 9, int NT = 256> constexpr int conv_w_floats()
{
    return (conv_fast_dma<NI, TW, VEC>() && !(NI == 4 && TAPS == 9)) ? ((KC * TAPS * NI * 4 + NT - 1) / NT) * NT * 4 : KC * TAPS * NI * 16;
}
template <int NI, int TW, bool VEC, int NT = 256, bool HALF = false> constexpr int conv_in_floats()
{
    return conv_fast_dma<NI, TW, VEC>() ? ((KC * TileGeom<TW, VEC>::PLANE / 4 + NT - 1) / NT) * NT * 4 : KC * TileGeom<TW, VEC, HALF>::PLANE;
}
// ONEKB: operators whose whole K fits ONE K-block (ConvA of the image layer: 6 channels) never stage a second buffer
template <int NI, int TW, bool VEC, int TAPS = 9, bool ONEKB = false, int NT = 256, bool HALF = false> constexpr int conv_lds_bytes()
{
    return (ONEKB ? 1 : 2) * (conv_in_floats<NI, TW, VEC, NT, HALF>() + conv_w_floats<NI, TW, VEC, TAPS, NT>()) * 4;
}

#ifndef EIG_TIMING
#define EIG_TIMING 0  // measurement-only builds: per-wave s_memtime breakdown of the K loop into a.dbg
#endif
#ifndef EIG_CONV_OCC
#define EIG_CONV_OCC 2  // blocks per CU the register allocation is capped for (__launch_bounds__ 2nd argument)
#endif
constexpr int CONV_THREADS = 256;  // 4 waves per block (W8 instantiations: 512 = 8 waves)
// SPLIT = 1 (W8): EIGHT waves share the block's 256-pixel x NB-column tile -- waves 0..3 own parity classes (0, px) of the four 64-pixel
// regions, waves 4..7 classes (1, px): two 16-row sub-tiles x NI accumulators per wave instead of four.  Same LDS tile, same
// DMA bytes, half the accumulators and half the epilogue per wave, so FOUR waves per SIMD are resident (two blocks per CU):
// while one block's waves run prologue / epilogue the SIMD still has two waves feeding the matrix pipe (a lone wave sustains
// 0.70 of peak, two 0.93: scripts/mfma_occupancy.hip), and small maps whose launches do not fill the chip get twice as many
// waves out of the same tiles.  Which lane computes a pixel changes, its fma chain does not: results are bit-identical.
// SPLIT = 2 (H4, 8-wide tiles): the same two-classes-per-wave waves in FOUR-wave blocks of TWO images (128 pixels): wave w computes
// classes (w >> 1, px) of image w & 1.  Twice as many blocks of half the work, each staging its own weight slab: for launches of
// FEWER THAN TWO BLOCKS PER CU, whose time is the CU that received ceil(blocks / 256) of them -- 300 blocks on 256 CUs
// cost two block times, 600 half blocks three halves.
#ifndef EIG_W8_OCC
#define EIG_W8_OCC 4  // waves per SIMD the W8 register allocation is capped for
#endif

#ifndef EIG_UP4_OCC
#define EIG_UP4_OCC 2  // the 2x2-form pass would fit a third block per CU (88 VGPRs, 48 KB of LDS); measured: no faster (96.9 vs 95.8 ms)
#endif
#ifndef EIG_ONEKB_OCC
#define EIG_ONEKB_OCC 4  // single-K-block operators: one LDS buffer (32 KB), four blocks per CU -- their time is prologue + DMA round trip +
#endif                   // epilogue around 216 MFMAs per wave, which only other blocks' MFMAs can cover
template <int NI, int TW, int EPI, bool VEC, bool ONEKB = false, bool FUSE_ = false, int SPLIT = 0>   // (FUSE_: the in-kernel 2x2-form chain of rounds 3-5, removed in round 6; the parameter keeps the kernels' names)
__global__ void __launch_bounds__(SPLIT == 1 ? 512 : CONV_THREADS, SPLIT ? EIG_W8_OCC : (ONEKB ? EIG_ONEKB_OCC : ((EPI == EPI_UP4 && NI == 4 && TW == 16 && VEC) ? EIG_UP4_OCC : EIG_CONV_OCC)))
conv3x3_mfma(const ConvArgs a)
{
    constexpr bool W8 = SPLIT != 0;   // two parity classes per wave (SPLIT 1: eight waves on the 256-pixel tile; SPLIT 2: four waves on two images)
    constexpr bool H4 = SPLIT == 2;
    static_assert(!H4 || TW == 8 || TW == 4, "SPLIT 2: 8- or 4-wide tiles (one image per wave) only");
    // TW = 4: regions of 4 columns x 16 rows (2 x 8 pooling windows) for maps that 8 x 8 tiles cover badly -- 20 x 15 (the top layer of the
    // reference's own 160 x 120): five strips = 94 % instead of six 8 x 8 tiles = 78 %.  A lane owns rows 4 q .. 4 q + 3 x the 4 columns, so
    // every epilogue indexes its accumulators exactly as the 16-wide map does; only as half blocks (the 12-float rows cost LDS).
    static_assert(TW != 4 || (H4 && VEC && EPI != EPI_LSTM_PACKED && EPI != EPI_UP4C), "4-wide tiles: half blocks with 16-byte staging only");
    static_assert(!FUSE_, "the in-kernel chain of an unpooled source was removed in round 6 (a ConvLSTM chains it inside only in its Winograd forms)");
    static_assert(!W8 || (VEC && !ONEKB && EPI != EPI_UP4C && EPI != EPI_LSTM_PACKED), "W8: 16-byte staging, per-pixel or pooled epilogues");
    constexpr int NT = SPLIT == 1 ? 512 : CONV_THREADS;  // threads per block; a DMA round is NT 16-byte chunks
    constexpr int NWS = H4 ? 2 : 4;              // 64-pixel regions per block
    constexpr int MI_N = W8 ? 2 : 4;             // 16-row sub-tiles (parity classes) per wave
    using G = TileGeom<TW, VEC, H4>;
    constexpr int TH = G::TH, NIMG = G::NIMG, S = G::S, XO = G::XO, PH = G::PH, PLANE = G::PLANE;
    constexpr int NB = NI * 16;
    // EPI_UP4: the 2x2 form of `unpool x2 -> conv3x3`.  The launch runs at the SOURCE resolution; a block
    // computes, for its parity class (py, px), the partial chains of the output pixels (2Y+py, 2X+px) of its tile:
    // k = (channel, a, b), 4 taps per channel, tap (a, b) reads source pixel (Y+a-1+py, X+b-1+px) -- inside the same
    // haloed tile a 3x3 convolution stages.  The ConvLSTM launch that follows starts its accumulators from the result.
    // EPI_UP4C: the same pass for operators with <= 16 output columns (the packed image-layer ConvLSTM): ONE block computes all four
    // classes of its tile -- the four N-tiles of the NI = 4 kernel ARE the four classes, each with its own gather (the class
    // only shifts the tap window) and its own 16 weight columns, so the tile is staged once instead of four times.
    constexpr int TAPS = epi_taps(EPI);
    static_assert(EPI != EPI_UP4C || NI == 4, "EPI_UP4C: the four N-tiles are the four parity classes");
    const unsigned long long t_entry = EIG_TIMING ? __builtin_readcyclecounter() : 0;
#if EIG_PRO_PRIO
    // Prologue and epilogue outrank the SIMD partner's K loop: the two waves of a SIMD (one per resident block) are arbitrated by
    // priority, then age, and a young wave's ~400 prologue instructions were taking 19K cycles beside an old wave's MFMA stream
    // (scripts/timeline.py) -- time during which the matrix pipe has ONE wave feeding it
    __builtin_amdgcn_s_setprio(EIG_PRO_PRIO);
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool FAST = conv_fast_dma<NI, TW, VEC>();
    constexpr int INF = conv_in_floats<NI, TW, VEC, NT, H4>();  // floats of the input area (KC * PLANE, padded for FAST)
    constexpr int BUF = INF + conv_w_floats<NI, TW, VEC, epi_taps(EPI), NT>();  // floats per LDS buffer: [KC][PLANE] inputs | [KC*9][NB] weights

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave of the block: its DMA share
    const int ws = W8 ? (wv % NWS) : wv;                       // which 64-pixel region of the tile (16-wide: rows 4 ws ..; 8-wide: image ws)
    const int ch2 = W8 ? (wv / NWS) : 0;                       // W8 / H4: row parity py of the two classes this wave computes
    const int q = lane >> 4;
    const int col = lane & 15;

    // ---- block -> (N-block, image group, tile), XCD-aware: workgroup b runs on XCD b % 8 (observed dispatch order;
    // speed only, never correctness).  Within one XCD consecutive workgroups walk the N-blocks of ONE tile, so the
    // n_nblk blocks that read the same input tile share it through that XCD's L2 instead of re-reading it from HBM
    // n_nblk times; the weight slabs they stream are K-block-sized and stay L2-resident as well.
    const int tiles = a.tilesX * a.tilesY;
    const int ngroups = (a.B + NIMG - 1) / NIMG;
    const int ntile = ngroups * tiles;
    const int xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int nnb = (EPI == EPI_UP4) ? a.n_nblk * 4 : a.n_nblk;  // EPI_UP4: the four parity classes of a tile are neighbours too
    const int nbe = xi % nnb;
    const int nblk = (EPI == EPI_UP4) ? nbe >> 2 : nbe;
    const int cls = (EPI == EPI_UP4) ? nbe & 3 : 0;
    // tile_map 1: every XCD owns a CONTIGUOUS range of tiles, so the tiles in flight on one XCD are spatial neighbours and
    // their overlapping halos (1.7x the tile in the 16-byte-chunk layout) are L2 hits; 0: tiles interleaved over the XCDs.
    const int tlin = a.tile_map ? xcd * ((ntile + 7) >> 3) + xi / nnb : (xi / nnb) * 8 + xcd;
    if (tlin >= ntile) return;  // grid is padded to a multiple of 8 tiles
    const int bgrp = tlin / tiles;
    const int t = tlin - bgrp * tiles;
    const int tyi = t / a.tilesX, txi = t - tyi * a.tilesX;
    const int y0 = tyi * TH, x0 = txi * TW;

    // ---- staging slots of this thread (K-block invariant): LDS slot tid + 256 r -> pixel offset in the source plane
    // (-1: zero fill) and image.  VEC: slots are the 16-B chunks of the whole K-block tile (the channel inside the
    // K-block is (tid + 256 r) / slots-per-channel); !VEC: slots are the floats of ONE channel plane.
    constexpr int RC = S / 4;
    constexpr int PER_C = VEC ? NIMG * PH * RC : PLANE;
    constexpr int NR = VEC ? (KC * PER_C + NT - 1) / NT : (PLANE + NT - 1) / NT;
    int sl_off[NR], sl_img[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int p = tid + r * NT;
        int rem = VEC ? p % PER_C : p;
        constexpr int RW = VEC ? RC : S;
        const int img = rem / (PH * RW);
        rem -= img * (PH * RW);
        const int yy = rem / RW, xx = rem - yy * RW;
        const int gy = y0 + yy - 1, gx = VEC ? (x0 - 4 + 4 * xx) : (x0 + xx - 1);
        const int b = bgrp * NIMG + img;
        const bool ok = (VEC ? p < KC * PER_C : p < PLANE) && b < a.B && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        // VEC: byte offset of the chunk inside ONE image of the source, channel part included; -1 = out of range,
        // which the buffer bounds check turns into zeros.  !VEC: pixel offset.
        sl_off[r] = ok ? (VEC ? ((p / PER_C) * a.H * a.W + gy * a.W + gx) * 4 : gy * a.W + gx) : -1;
        sl_img[r] = VEC ? img : b;
    }

    // ---- K loop: K-blocks of KC channels, enumerated across the sources, two LDS buffers.
    // Both operands go global -> LDS by DMA (global_load_lds: no staging registers, no ds_write pass; every lane
    // supplies the global address of its LDS slot, or of a zero buffer outside the image / for padded channels).
    // The DMA instructions of K-block kb+1 are INTERLEAVED with the 18 MFMA steps of K-block kb (one every other
    // step): a DMA issued while the wave would anyway be waiting for the matrix pipe costs nothing, whereas a burst
    // of them ahead of the MFMAs measured ~0.7 % of the K-block time per instruction.  ONE barrier per K-block.
    constexpr int NWR = (KC * TAPS * (NB / 4) + NT - 1) / NT;     // weight DMA rounds per K-block
    constexpr int NIN = VEC ? NR : KC * NR;                       // input DMA ops per K-block
    constexpr int NOPS = NWR + NIN;
    constexpr int NSTEP = KC * TAPS / 4;                          // 18 (8 for the 2x2 form)
    struct KB { int s, c0, kc; };
    // per-source scalars picked with selects (indexing a.src[] with a run-time index would put the argument struct in scratch
    // and turn everything derived from it -- descriptors, LDS-DMA bases -- into per-lane values)
    const int cpad0 = a.src[0].Cpad, cpad1 = a.src[1].Cpad, cpad2 = a.src[2].Cpad;
    auto cpad_of = [&](int si) __attribute__((always_inline)) { return si == 0 ? cpad0 : (si == 1 ? cpad1 : cpad2); };
    auto kb_first = [&]() { KB k; k.s = 0; k.c0 = 0; k.kc = min(KC, cpad0); return k; };
    auto kb_next = [&](KB k) __attribute__((always_inline)) {
        k.c0 += KC;
        if (k.c0 >= cpad_of(k.s)) { k.c0 = 0; ++k.s; }
        k.kc = (k.s < a.nsrc) ? min(KC, cpad_of(k.s) - k.c0) : 0;
        return k;
    };
    // Buffer descriptors (VEC path): one per source covering the NIMG images of this block, one for this N-block's
    // weight slab.  A DMA then needs no per-lane address arithmetic at all: voffset is a K-block-invariant register,
    // the channel / K-block part is a scalar offset, and out-of-image or padded-channel slots are simply out of
    // range of the descriptor, which the hardware turns into zeros.
    const int b0 = bgrp * NIMG;
    const int nimg_here = min(NIMG, a.B - b0);
    auto make_rsrc = [&](const float* ptr, int C, int Ct) {
        const size_t plane = (size_t)(a.H * a.W);
        return __builtin_amdgcn_make_buffer_rsrc((void*)(ptr + (size_t)b0 * Ct * plane), 0, (int)(((size_t)(nimg_here - 1) * Ct + C) * plane * 4), 0x00020000);
    };
    // (unused sources alias source 0: constant indices only, see cpad_of above)
    const bool has1 = a.nsrc > 1, has2 = a.nsrc > 2;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.src[0].ptr, a.src[0].C, a.src[0].Ct);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(has1 ? a.src[1].ptr : a.src[0].ptr, has1 ? a.src[1].C : a.src[0].C, has1 ? a.src[1].Ct : a.src[0].Ct);
    const __amdgpu_buffer_rsrc_t rs2 = make_rsrc(has2 ? a.src[2].ptr : a.src[0].ptr, has2 ? a.src[2].C : a.src[0].C, has2 ? a.src[2].Ct : a.src[0].Ct);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wpk + ((size_t)cls * a.n_nblk + nblk) * a.krows * NB), 0, a.krows * NB * 4, 0x00020000);
    auto buf_dma16 = [&](int si, float* lds_dst, int voff, int soff) {
        auto l = (__attribute__((address_space(3))) void*)lds_dst;
        if (si == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, l, 16, voff, soff, 0, 0);
        else if (si == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, l, 16, voff, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, l, 16, voff, soff, 0, 0);
    };
    // one DMA instruction (op j of K-block k) into buffer `buf`; wrow = first weight row of that K-block
    auto dma_op = [&](int j, const KB& k, int wrow, float* buf) {
        ConvSrc src;
        src.ptr = k.s == 0 ? a.src[0].ptr : (k.s == 1 ? a.src[1].ptr : a.src[2].ptr);
        src.C = k.s == 0 ? a.src[0].C : (k.s == 1 ? a.src[1].C : a.src[2].C);
        src.Cpad = cpad_of(k.s);
        src.Ct = k.s == 0 ? a.src[0].Ct : (k.s == 1 ? a.src[1].Ct : a.src[2].Ct);
        if (j < NWR) {
            const int n16 = k.kc * TAPS * (NB / 4);
            const int base = j * NT + wv * 64, ch = base + lane;
            if (ch < n16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(buf + INF + (size_t)base * 4), 16,
                                                         tid * 16, (wrow * NB + j * NT * 4) * 4, 0, 0);
            return;
        }
        const int chs = a.H * a.W;
        if (VEC) {
            const int r = j - NWR;
            const int p = tid + r * NT;
            const int soff = k.c0 * chs * 4;
            const bool tail = k.c0 + KC > src.C;  // padded channels present: mask them explicitly (rare, tiny layers)
            if (r < NR) {
                int vo = sl_off[r < NR ? r : 0];
                if (NIMG > 1) vo = vo < 0 ? vo : vo + sl_img[r < NR ? r : 0] * src.Ct * chs * 4;
                if (tail && k.c0 + p / PER_C >= src.C) vo = -1;
                if (p < k.kc * PER_C) buf_dma16(k.s, buf + (r * NT + wv * 64) * 4, vo, soff);
            }
        } else {
            const int c = (j - NWR) / NR, r = (j - NWR) % NR;
            if (c < k.kc) {
                const bool pok = sl_off[r] >= 0;
                const int gy = pok ? sl_off[r] / a.W : 0, gx = pok ? sl_off[r] - gy * a.W : 0;
                const float* g = (pok && (k.c0 + c < src.C))
                                     ? src.ptr + ((size_t)sl_img[r] * src.Ct + k.c0 + c) * chs + gy * a.W + gx : a.zeros;
                if (tid + r * NT < PLANE)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(buf + c * PLANE + r * NT + wv * 64), 4, 0, 0);
            }
        }
    };

    // Branch-free variant (FAST): no exec masking, no per-lane predicates, no control flow -- the nine DMA instructions of a
    // K-block cost ~4 instructions each instead of ~12 (every non-MFMA instruction in the wave's stream costs matrix-pipe
    // time).  The K-block's byte offset is folded into the lane offset with a SATURATING add, so that the descriptor's
    // range check sees the whole offset: channels >= C (padding) and rows >= krows fall out of range and read zeros,
    // `no next K-block` is an offset of 2^31, and an invalid slot (-1) stays 0xffffffff.  NIMG == 1 here (TW == 16).
    constexpr int LASTW = (NWR - 1) * NT;                           // first chunk of the last weight round (NI = 4, 9 taps: 1024)
    const int vw_full = tid * 16;                                   // whole weight rounds: chunk j*NT + tid
    const int vw_last = (LASTW + (wv & 1) * 64 + lane) * 16;        // the last round holds 128 chunks: waves 2,3 repeat waves 0,1 (W8: waves 2..7 skip it)
    static_assert(!FAST || NI != 4 || TAPS != 9 || (KC * 9 * (NB / 4)) % NT == 128, "FAST staging: the last weight round of NI = 4 must hold 128 chunks");
    static_assert(!FAST || TAPS != 4 || NI != 4 || (KC * 4 * (NB / 4)) % NT == 0, "FAST staging, 2x2 form: whole weight rounds");
    const int aH = a.H, aW = a.W;
    // descriptor of the source a K-block reads, from SCALAR selects of base pointer and size (selecting between whole
    // descriptors ends in a scratch table + waterfall loop)
    auto src_base = [&](const float* ptr, int Ct) __attribute__((always_inline)) { return ptr + (size_t)b0 * Ct * (a.H * a.W); };
    auto src_bytes = [&](int C, int Ct) __attribute__((always_inline)) { return ((nimg_here - 1) * Ct + C) * (a.H * a.W) * 4; };
    const float* const sp0 = src_base(a.src[0].ptr, a.src[0].Ct);
    const float* const sp1 = has1 ? src_base(a.src[1].ptr, a.src[1].Ct) : sp0;
    const float* const sp2 = has2 ? src_base(a.src[2].ptr, a.src[2].Ct) : sp0;
    const int sn0 = src_bytes(a.src[0].C, a.src[0].Ct);
    const int sn1 = has1 ? src_bytes(a.src[1].C, a.src[1].Ct) : sn0;
    const int sn2 = has2 ? src_bytes(a.src[2].C, a.src[2].Ct) : sn0;
    const unsigned long long su0 = (unsigned long long)sp0, sd1 = (unsigned long long)sp1 - su0, sd2 = (unsigned long long)sp2 - (unsigned long long)sp1;
    auto rsrc_of = [=](int si) __attribute__((always_inline)) {
        // additive form + readfirstlane: plain 3-way selects were turned into a table in scratch memory indexed by si
        const unsigned long long u = su0 + (si > 0 ? sd1 : 0ull) + (si > 1 ? sd2 : 0ull);
        const int n = sn0 + (si > 0 ? sn1 - sn0 : 0) + (si > 1 ? sn2 - sn1 : 0);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
    };
    // (captures by VALUE: with by-reference captures the closure -- pointers to these locals -- stayed in scratch memory)
    auto dma_fast = [=](int j, const KB k, const __amdgpu_buffer_rsrc_t rs_k, unsigned soff_in, unsigned soff_w, float* buf) __attribute__((always_inline)) {
        if (j < NWR) {
            const unsigned soff = soff_w;
            const bool lastw = TAPS == 9 && NI == 4 && (j == NWR - 1);
            if (SPLIT == 1 && lastw && wv >= 2) return;  // wave-uniform: 128 chunks = two waves
            const unsigned vo = __builtin_elementwise_add_sat((unsigned)(lastw ? vw_last : vw_full + j * NT * 16), soff);
            float* dst = buf + INF + (lastw ? (LASTW + (wv & 1) * 64) * 4 : (j * NT + wv * 64) * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)dst, 16, (int)vo, 0, 0, 0);
            return;
        }
        const int r = j - NWR;
        const unsigned soff = soff_in;
        const int slot = r < NR ? sl_off[r < NR ? r : 0] : -1;
        const unsigned vo = __builtin_elementwise_add_sat((unsigned)slot, soff);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (__attribute__((address_space(3))) void*)(buf + (r * NT + wv * 64) * 4), 16, (int)vo, 0, 0, 0);
    };

    const unsigned long long t_setup = EIG_TIMING ? __builtin_readcyclecounter() : 0;  // slots and descriptors ready
    int nkb = 0;
    for (int s = 0; s < a.nsrc; ++s) nkb += (cpad_of(s) + KC - 1) / KC;
    KB cur_kb = kb_first();
    int wrow = 0;  // first packed weight row of the current K-block

#pragma unroll
    for (int j = 0; j < NOPS; ++j) {
        if constexpr (FAST) dma_fast(j, cur_kb, rsrc_of(0), 0u, 0u, lds);
        else dma_op(j, cur_kb, wrow, lds);
    }
    // (the first K-block is in flight: everything below up to the wait overlaps its latency)

    // ---- A-operand gather addresses: lane row r = lane&15 = 4*rq + rreg -> pixel (dy, dx); k-slot j = lane>>4.
    // k = 4*step + j, (channel, tap) = divmod(k, 9): period 9 steps = 4 channels -> nine address registers per layout.
    int addrA[9];
    // class-major map: MFMA row r of sub-tile (py, px) is pixel (2 wy + py, 2 wx + px) of the wave's region
    const int g_wy = (TW == 16) ? (col & 3) >> 1 : (TW == 4 ? 2 * (col >> 2) + ((col & 3) >> 1) : col >> 2);
    const int g_wx = (TW == 16) ? 2 * (col >> 2) + (col & 1) : (TW == 4 ? col & 1 : col & 3);
    const int g_base = ((TW == 16) ? (ws * 4 + 2 * g_wy) * S + 2 * g_wx + XO      // sub-tile of class (py, px) adds py * S + px
                                   : ws * PH * S + 2 * g_wy * S + 2 * g_wx + XO)  // one image per wave
                       + ch2 * S;  // W8: this wave's classes are (ch2, 0) and (ch2, 1) -- the row parity goes into the base register
    {
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int k = 4 * s + q;
            const int c = k / 9, tap = k - 9 * c;
            const int ky = tap / 3, kx = tap - 3 * ky;
            addrA[s] = g_base + c * PLANE + ky * S + kx;
        }
    }
    // 2x2 form: k = 4*step + q -> channel = step, tap (a, b) = (q >> 1, q & 1): one address, the channel is an immediate
    const int addr4 = g_base + ((q >> 1) + (cls >> 1)) * S + (q & 1) + (cls & 1);
    const int boff = q * NB + col * NI;  // weight slab row k = [16 channels][NI tiles]: a lane's NI values are contiguous

    // acc[mi]: !W8 sub-tile mi = class (mi >> 1, mi & 1); W8 sub-tile mi = class (ch2, mi)
    f32x4 acc[MI_N][NI];
#pragma unroll
    for (int mi = 0; mi < MI_N; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned long long t_prewait = EIG_TIMING ? __builtin_readcyclecounter() : 0;  // first K-block issued, gather addresses ready
#if EIG_KLOOP_PRIO
    // (tried: a wave inside the K loop outranking its SIMD partner -- 0.5 % slower: it starves the partner's prologue / epilogue)
    __builtin_amdgcn_s_setprio(EIG_KLOOP_PRIO);
#endif
#if EIG_PRO_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // The chain of the unpooled source (an EPI_UP4 launch at half this resolution wrote it) is added to this launch's chain with
    // one fp32 addition after the K loop.  Sub-tile mi of a lane is parity class mi and its registers 0..3 are the source pixels
    // (wy, wx) of the lane's windows: per (class, N-tile) two 8-byte loads (16-wide tiles: rows wy = 0, 1, columns 2q, 2q + 1) or
    // one 16-byte load (8-wide tiles: row q, columns 0..3) from plane `class` of the chain tensor.  Issued in one burst they
    // saturate the wave's outstanding vector-memory operations and the CU's address path in front of the first MFMA (round 1:
    // 21K cycles, scripts/timeline.py), so they are SPREAD over the MFMA steps of K-block 0 and land long before the K loop ends.
    constexpr bool HAS_UP = (EPI == EPI_LSTM || EPI == EPI_LSTM_PACKED || EPI == EPI_RAW);  // operators that can be handed an unpooled source's chain
    constexpr int UPV = (TW != 8) ? 2 : 1;              // loads per (class, N-tile)
    constexpr int NUPL = HAS_UP ? MI_N * NI * UPV : 0;  // loads per lane
    f32x4 upc[MI_N][HAS_UP ? NI : 1];
    const bool has_up = HAS_UP && a.acc_init != nullptr;
    int up_off[UPV];
    const float* up_base = a.acc_init;
    int up_hw = 0, up_cstride = 0;
    if (has_up) {
        const int Hs = a.H >> 1, Ws = a.W >> 1;
        up_hw = Hs * Ws;
        up_cstride = a.n_nblk * NB * up_hw;
        const int b = bgrp * NIMG + (TW == 16 ? 0 : ws);
#pragma unroll
        for (int v = 0; v < UPV; ++v) {
            // window row / first window column of this load, in pixels of THIS launch's resolution
            const int gy0 = (TW == 16) ? y0 + 4 * ws + 2 * v : (TW == 4 ? tyi * TH + 4 * q + 2 * v : tyi * TH + 2 * q);
            const int gx0 = (TW == 16) ? x0 + 4 * q : txi * TW;
            // element offset inside this wave's image block [4][n_nblk*NB][Hs][Ws] (the image is wave-uniform); windows outside
            // the image read element 0 instead -- their accumulators are never stored
            up_off[v] = (b < a.B && gy0 < a.H && gx0 < a.W) ? ((nblk * NB + col) * up_hw + (gy0 >> 1) * Ws + (gx0 >> 1)) * 4 : 0;
        }
        const int bw = __builtin_amdgcn_readfirstlane(b);
        up_base = a.acc_init + (size_t)(bw < a.B ? bw : 0) * 4 * up_cstride;
    }
    // buffer loads: the lane part of the address is one of UPV byte offsets (up_off), the (N-tile, class) part a scalar -- with flat
    // addresses the compiler hoists the loop-invariant 64-bit pointers out of the K loop (128 VGPRs, spilled)
    const __amdgpu_buffer_rsrc_t rs_up = __builtin_amdgcn_make_buffer_rsrc((void*)up_base, 0, has_up ? 4 * up_cstride * 4 : 0, 0x00020000);
    auto up_load = [&](int mi, int ni, int v) __attribute__((always_inline)) {
        const int soff = (ni * 16 * up_hw + (W8 ? 2 * ch2 + mi : mi) * up_cstride) * 4;  // plane = parity class of sub-tile mi
        if constexpr (TW != 8) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_up, up_off[v], soff, 0));
            upc[mi][ni][2 * v] = t[0]; upc[mi][ni][2 * v + 1] = t[1];
        } else {
            upc[mi][ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_up, up_off[0], soff, 0));
        }
    };

    unsigned long long t_mfma = 0, t_wait = 0, t_bar = 0, t_all0 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
    auto kiter = [&](const int kb, auto kfirst_tag) __attribute__((always_inline)) {
        const unsigned long long tk0 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
        float* const cur = lds + (kb & 1) * BUF;
        float* const nxt = lds + ((kb & 1) ^ 1) * BUF;
        const KB nxt_kb = kb_next(cur_kb);
        const int wrow_nxt = wrow + cur_kb.kc * TAPS;
        const bool more_kb = (kb + 1 < nkb) && EIG_ABLATE != 1;
        const __amdgpu_buffer_rsrc_t rs_nxt = rsrc_of(nxt_kb.s);
        // byte offsets of the next K-block inside its source / the weight slab; 2^31 = `nothing to stage` (out of range)
        const unsigned soff_in_nxt = more_kb ? (unsigned)(nxt_kb.c0 * (aH * aW) * 4) : 0x80000000u;
        const unsigned soff_w_nxt = more_kb ? (unsigned)(wrow_nxt * NB * 4) : 0x80000000u;

        // Every LDS offset of the operand gather is an immediate (nine address registers per period of 9 steps, sub-tile and
        // channel-period offsets folded into the instruction): the four A reads of a step pair up into two ds_read2_b32, no
        // address VALU.
        const float* const in_lds = cur;
        const float* const w_lds = cur + INF + boff;
        auto body = [&](auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;  // K-block 0 of an operator with an unpooled-source chain: issue its loads
            constexpr int PL = PLANE;
            const int* const ad = addrA;
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                // second period only for a full K-block (wave-uniform).  FAST: always -- a 4-channel K-block's upper
                // channels are staged as zeros (out of range), so the extra steps add exact zeros, and the straight-line
                // body keeps the DMA instructions free of control flow
                if (FAST || (TAPS == 9 ? (st < 9 || cur_kb.kc > 4) : st < cur_kb.kc)) {
                    const int per = st / 9, s9 = st % 9;
                    float av[MI_N], bv[NI];
                    float avc[MI_N][3];  // EPI_UP4C: the gathers of classes 1..3
#pragma unroll
                    for (int mi = 0; mi < MI_N; ++mi) {
                        const int moff = W8 ? mi : (mi >> 1) * S + (mi & 1);  // parity class (py, px) of sub-tile mi (W8: py is in the base register)
                        av[mi] = (TAPS == 9) ? in_lds[ad[s9] + per * 4 * PL + moff] : in_lds[addr4 + st * PL + moff];
                        if constexpr (EPI == EPI_UP4C) {  // class c = (py, px) shifts the tap window by (py, px)
#pragma unroll
                            for (int c = 1; c < 4; ++c) avc[mi][c - 1] = in_lds[addr4 + ((c >> 1) * S + (c & 1)) + st * PL + moff];
                        }
                    }
                    if (NI == 4) {  // one ds_read_b128
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(w_lds + st * 4 * NB);
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) bv[ni] = b4[ni];
                    } else {
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) bv[ni] = w_lds[st * 4 * NB + ni];
                    }
#pragma unroll
                    for (int mi = 0; mi < MI_N; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32((EPI == EPI_UP4C && ni > 0) ? avc[mi][ni > 0 ? ni - 1 : 0] : av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
                }
                if constexpr (ONEKB) {
                    // nothing to stage: the only K-block is being computed on (and there is no second LDS buffer)
                } else if constexpr (FAST) {
#pragma unroll
                    for (int j = 0; j < NOPS; ++j)
                        if ((EIG_DMA_EARLY ? (j * EIG_DMA_EARLY < NSTEP ? j * EIG_DMA_EARLY : NSTEP - 1) : j * NSTEP / NOPS) == st) dma_fast(j, nxt_kb, rs_nxt, soff_in_nxt, soff_w_nxt, nxt);
                } else if (more_kb) {
#pragma unroll
                    for (int j = 0; j < NOPS; ++j)
                        if (j * NSTEP / NOPS == st) dma_op(j, nxt_kb, wrow_nxt, nxt);
                }
                if constexpr (FIRST) {
#pragma unroll
                    for (int mi = 0; mi < MI_N; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                            for (int v = 0; v < UPV; ++v)
                                if (((mi * NI + ni) * UPV + v) * NSTEP / NUPL == st) up_load(mi, ni, v);
                    __builtin_amdgcn_sched_barrier(0);  // keep them in their step: left alone the scheduler sinks all of them to the end of the K-block
                }
            }
        };
        body(kfirst_tag);
        wrow = wrow_nxt;
        cur_kb = nxt_kb;
        const unsigned long long tk1 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA for K-block kb+1 has landed
        const unsigned long long tk2 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
        __syncthreads();
        if (EIG_TIMING) { const unsigned long long tk3 = __builtin_readcyclecounter(); t_mfma += tk1 - tk0; t_wait += tk2 - tk1; t_bar += tk3 - tk2; }
    };
    // K-block 0 of an operator with an unpooled-source chain is peeled: its steps carry that chain's loads
    int kb_begin = 0;
    if constexpr (HAS_UP) {
        if (has_up) { kiter(0, std::true_type{}); kb_begin = 1; }
    }
    for (int kb = kb_begin; kb < nkb; ++kb) kiter(kb, std::false_type{});
#if EIG_KLOOP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#if EIG_EPI_PRIO
    __builtin_amdgcn_s_setprio(EIG_EPI_PRIO);
#endif
    const unsigned long long t_loop1 = EIG_TIMING ? __builtin_readcyclecounter() : 0;

    // ---------------------------------------------------------------- epilogue
    if (has_up) {
#pragma unroll
        for (int mi = 0; mi < MI_N; ++mi)
#pragma unroll
            for (int ni = 0; ni < (HAS_UP ? NI : 1); ++ni) acc[mi][ni] = acc[mi][ni] + upc[mi][ni];
    }
    const int HW = a.H * a.W;
    auto timeline_record = [&](unsigned long long t_mid) {  // measurement builds only (EIG_TIMING): per-wave timeline
        if (EIG_TIMING && a.dbg && lane == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* d = a.dbg + ((size_t)blockIdx.x * (W8 ? 8 : 4) + wv) * 8;  // (the host sizes the buffer for 8 waves per block)
            d[0] = t_entry; d[1] = t_all0; d[2] = t_loop1; d[3] = __builtin_readcyclecounter();
            d[4] = (unsigned long long)hwid | ((unsigned long long)xcc << 32); d[5] = t_mfma; d[6] = t_setup; d[7] = t_prewait; (void)t_mid; (void)t_bar;
        }
    };
    // ---- what a lane owns (class-major map): 4 SEGMENTS of 4 horizontally contiguous pixels; element j of segment s is the
    // accumulator register seg_reg(s, j) of sub-tile (class) seg_mi(s, j).  16-wide tiles: segment s = row 4 wv + s, columns
    // 4 q .. 4 q + 3; 8-wide tiles: segment s = row 2 q + (s >> 1), columns 4 (s & 1) .. + 3 of the wave's image.
    // Pooled / half-resolution view: register reg of every sub-tile is window (wy, wx): 16-wide (reg >> 1, 2 q + (reg & 1)) of the
    // wave's 2 x 8 windows, 8-wide (q, reg).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int eb = bgrp * NIMG + (TW == 16 ? 0 : ws);          // image of this wave
    const int ey0 = (TW == 16) ? y0 + 4 * ws : tyi * TH;       // first row / column of the wave's region
    const int ex0 = (TW == 16) ? x0 : txi * TW;
    auto seg_row = [&](int sgi) __attribute__((always_inline)) { return ey0 + ((TW == 16) ? sgi : (TW == 4 ? 4 * q + sgi : 2 * q + (sgi >> 1))); };
    auto seg_col = [&](int sgi) __attribute__((always_inline)) { return ex0 + ((TW == 16) ? 4 * q : (TW == 4 ? 0 : 4 * (sgi & 1))); };
    // A wave walks its segments sl = 0 .. NSEG-1.  !W8: all four (sgi = sl).  W8: the two whose row parity is this wave's ch2
    // (16-wide: rows ch2, ch2 + 2 of the region; 8-wide: row 2 q + ch2, both column halves); element j of such a segment is
    // register 2 sl + (j >> 1) of the wave's sub-tile j & 1 -- compile-time indices either way.
    constexpr int NSEG = W8 ? 2 : 4;
#define EIG_SGI(sl) (W8 ? ((TW != 8) ? ch2 + 2 * (sl) : 2 * ch2 + (sl)) : (sl))
#define EIG_SEG_MI(sl, j) (W8 ? ((j) & 1) : ((TW != 8) ? 2 * ((sl) & 1) + ((j) & 1) : 2 * ((sl) >> 1) + ((j) & 1)))
#define EIG_SEG_REG(sl, j) (W8 ? 2 * (sl) + ((j) >> 1) : ((TW != 8) ? 2 * ((sl) >> 1) + ((j) >> 1) : 2 * ((sl) & 1) + ((j) >> 1)))
    const bool alive = eb < a.B;
    if (!alive && !(W8 && EPI == EPI_CONVA)) { timeline_record(0); return; }  // (W8 ConvA: every wave takes part in the pooling exchange)

    if constexpr (EPI == EPI_RAW || EPI == EPI_UP4 || EPI == EPI_UP4C) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            float* dst;
            if (EPI == EPI_UP4C) {   // N-tile ni is class ni of the 16 output columns; column = gate * 4 + channel of the packed ConvLSTM: channels >= Cout are never read
                if ((col & 3) >= a.Cout) continue;
                dst = a.raw + (((size_t)eb * 4 + ni) * a.n_nblk * 16 + nblk * 16 + col) * HW;
            }
            else if (EPI == EPI_UP4) dst = a.raw + (((size_t)eb * 4 + cls) * a.n_nblk * NB + nblk * NB + ni * 16 + col) * HW;
            else {
                const int o = nblk * NB + ni * 16 + col;
                if (o >= a.Cout) continue;
                dst = a.raw + ((size_t)eb * a.Cout + o) * HW;
            }
#pragma unroll
            for (int sgi = 0; sgi < NSEG; ++sgi) {
                const int gy = seg_row(EIG_SGI(sgi)), gx = seg_col(EIG_SGI(sgi));
                if (gy >= a.H || gx >= a.W) continue;
                if (VEC) {  // W % 4 == 0: the whole segment is inside, one aligned 16-byte store
                    *reinterpret_cast<f32x4*>(dst + gy * a.W + gx) = (f32x4){acc[EIG_SEG_MI(sgi, 0)][ni][EIG_SEG_REG(sgi, 0)], acc[EIG_SEG_MI(sgi, 1)][ni][EIG_SEG_REG(sgi, 1)],
                                                                              acc[EIG_SEG_MI(sgi, 2)][ni][EIG_SEG_REG(sgi, 2)], acc[EIG_SEG_MI(sgi, 3)][ni][EIG_SEG_REG(sgi, 3)]};
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gx + j < a.W) dst[gy * a.W + gx + j] = acc[EIG_SEG_MI(sgi, j)][ni][EIG_SEG_REG(sgi, j)];
            }
        }
    } else if constexpr (EPI == EPI_LSTM) {
        const int ch = nblk * 16 + col;
        if (ch < a.Cout) {
            const float bi = a.bias[ch], bf = a.bias[a.Cout + ch], bc = a.bias[2 * a.Cout + ch], bo = a.bias[3 * a.Cout + ch];
            const size_t cbase = ((size_t)eb * a.Cout + ch) * HW;
            const size_t pbase = (size_t)ch * HW;
            const size_t pstride = (size_t)a.Cout * HW;
            auto cell = [&](float zi_, float zf_, float zc_, float zo_, float cold, float pi_, float pf_, float po_, float& cn, float& hn) __attribute__((always_inline)) {
                if (EIG_ABLATE == 2) { cn = (zi_ + bi) + (zf_ + bf) + pi_ * cold; hn = (zc_ + bc) + (zo_ + bo) + pf_ * po_; return; }  // measurement only: gate math removed
                lstm_cell(zi_, zf_, zc_, zo_, bi, bf, bc, bo, cold, pi_, pf_, po_, cn, hn);
            };
#pragma unroll
            for (int sgi = 0; sgi < NSEG; ++sgi) {
                const int gy = seg_row(EIG_SGI(sgi)), gx = seg_col(EIG_SGI(sgi));
                if (gy >= a.H || gx >= a.W) continue;
                const int pix = gy * a.W + gx;
                if (VEC) {
                    // W % 4 == 0: the four pixels of the segment are ONE aligned 16-byte access per tensor -- a quarter of the memory
                    // instructions of per-pixel accesses and whole 32-byte sectors per (channel, row)
                    const f32x4 cold4 = *reinterpret_cast<const f32x4*>(a.c_state + cbase + pix);
                    const f32x4 pi4 = *reinterpret_cast<const f32x4*>(a.peep + pbase + pix);
                    const f32x4 pf4 = *reinterpret_cast<const f32x4*>(a.peep + pstride + pbase + pix);
                    const f32x4 po4 = *reinterpret_cast<const f32x4*>(a.peep + 2 * pstride + pbase + pix);
                    f32x4 cn4, hn4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float cn, hn;
                        cell(acc[EIG_SEG_MI(sgi, j)][0][EIG_SEG_REG(sgi, j)], acc[EIG_SEG_MI(sgi, j)][1][EIG_SEG_REG(sgi, j)], acc[EIG_SEG_MI(sgi, j)][2][EIG_SEG_REG(sgi, j)],
                             acc[EIG_SEG_MI(sgi, j)][3][EIG_SEG_REG(sgi, j)], cold4[j], pi4[j], pf4[j], po4[j], cn, hn);
                        cn4[j] = cn; hn4[j] = hn;
                    }
                    *reinterpret_cast<f32x4*>(a.c_state + cbase + pix) = cn4;
                    *reinterpret_cast<f32x4*>(a.h_out + cbase + pix) = hn4;
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (gx + j >= a.W) continue;
                    float cn, hn;
                    cell(acc[EIG_SEG_MI(sgi, j)][0][EIG_SEG_REG(sgi, j)], acc[EIG_SEG_MI(sgi, j)][1][EIG_SEG_REG(sgi, j)], acc[EIG_SEG_MI(sgi, j)][2][EIG_SEG_REG(sgi, j)],
                         acc[EIG_SEG_MI(sgi, j)][3][EIG_SEG_REG(sgi, j)], a.c_state[cbase + pix + j], a.peep[pbase + pix + j], a.peep[pstride + pbase + pix + j],
                         a.peep[2 * pstride + pbase + pix + j], cn, hn);
                    a.c_state[cbase + pix + j] = cn;
                    a.h_out[cbase + pix + j] = hn;
                }
            }
        }
    } else if constexpr (EPI == EPI_LSTM_PACKED) {
        // C <= 4 (layer 0): ONE 16-column tile holds the 4 gates x 4 channel slots, column = 4*gate + channel, so the
        // four gates of a cell sit in four lanes (same q, columns ch, ch+4, ch+8, ch+12) and each of those lanes holds them for
        // its four windows (reg 0..3) of every class sub-tile.  A 4x4 transpose across the four lanes gives lane (group
        // g = col>>2, channel ch) ALL FOUR gates of register g: one cell per lane and sub-tile instead of four cells in a
        // quarter of the lanes (the gate math is ~190 instructions per cell).
        // Round d: every lane offers its register (g+d)&3, lane g pulls from group (g-d)&3 -> that group's gate, register g.
        const int ch = col & 3, g = col >> 2;
        const bool active = ch < a.Cout;
        const int cc = active ? ch : 0;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            float r[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int sel = (g + d) & 3;
                const float v = sel == 0 ? acc[mi][0][0] : (sel == 1 ? acc[mi][0][1] : (sel == 2 ? acc[mi][0][2] : acc[mi][0][3]));
                r[d] = (d == 0) ? v : __shfl(v, (lane & ~12) | (((g - d) & 3) << 2), 64);
            }
            // r[d] is gate (g-d)&3; gate s arrived in round (g-s)&3
            auto gate = [&](int sg) {
                const int d = (g - sg) & 3;
                return d == 0 ? r[0] : (d == 1 ? r[1] : (d == 2 ? r[2] : r[3]));
            };
            const float ai = gate(0), af = gate(1), ac = gate(2), ao = gate(3);
            // pixel of (class mi, register g)
            const int gy = ey0 + ((TW == 16) ? 2 * (g >> 1) + (mi >> 1) : 2 * q + (mi >> 1));
            const int gx = ex0 + ((TW == 16) ? 4 * q + 2 * (g & 1) + (mi & 1) : 2 * g + (mi & 1));
            if (!active || gy >= a.H || gx >= a.W) continue;
            const float bi = a.bias[cc], bf = a.bias[a.Cout + cc], bc = a.bias[2 * a.Cout + cc], bo = a.bias[3 * a.Cout + cc];
            const size_t cbase = ((size_t)eb * a.Cout + cc) * HW;
            const size_t pbase = (size_t)cc * HW;
            const size_t pstride = (size_t)a.Cout * HW;
            const int pix = gy * a.W + gx;
            const float cold = a.c_state[cbase + pix];
            float cnew, hnew;
            lstm_cell(ai, af, ac, ao, bi, bf, bc, bo, cold, a.peep[pbase + pix], a.peep[pstride + pbase + pix], a.peep[2 * pstride + pbase + pix], cnew, hnew);
            a.c_state[cbase + pix] = cnew;
            a.h_out[cbase + pix] = hnew;
        }
    } else if constexpr (EPI == EPI_CONVA) {
        // max-pool: window `reg` of the lane = the same register of the four class sub-tiles.  Pooled pixels of a lane: 16-wide tiles
        // 2 rows x 2 contiguous columns (two 8-byte accesses per tensor), 8-wide tiles 1 row x 4 columns (one 16-byte access).
        const int Ho = a.H >> 1, Wo = a.W >> 1;
        const size_t plane = (size_t)Ho * Wo;
        constexpr int NPR = (TW != 8) ? 2 : 1, NPC = (TW != 8) ? 2 : 4;  // pooled rows x contiguous pooled columns per lane
        const int pyo = (ey0 >> 1) + ((TW == 16) ? 0 : (TW == 4 ? 2 * q : q)), pxo = (ex0 >> 1) + ((TW == 16) ? 2 * q : 0);
        const bool vec_ok = VEC && (TW != 8 || (a.W % 8) == 0);  // 16-wide: W % 4 == 0 makes the pooled pairs 8-byte aligned; 8-wide: 16-byte rows need Wo % 4 == 0
        // W8: a pooling window's classes (0, px) and (1, px) sit in the SAME lane of the two waves ws and ws + 4 -- each takes the
        // max over its own two, the pair meets through LDS (free after the K loop), then wave ch2 stores pooled row ch2 (16-wide)
        // or the N-tiles of parity ch2 (8-wide).  max(max(v0, v1), max(v2, v3)) as before.
        float* const xch = lds;  // [waves][NI][4 registers][64 lanes]
        if constexpr (W8) {
            __syncthreads();  // every wave is done reading the last K-block's operands
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int chx = nblk * NB + ni * 16 + col;
                const float bbx = chx < a.Cout ? a.bias[chx] : 0.0f;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    xch[((wv * NI + ni) * 4 + reg) * 64 + lane] = fmaxf(relu_f(acc[0][ni][reg] + bbx), relu_f(acc[1][ni][reg] + bbx));
            }
            __syncthreads();
            if (!alive) { timeline_record(0); return; }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int ch = nblk * NB + ni * 16 + col;
            if (ch >= a.Cout) continue;
            if (W8 && TW == 8 && NI > 1 && (ni & 1) != ch2) continue;   // 8-wide: the pair splits the N-tiles
            if (W8 && TW == 8 && NI == 1 && ch2 != 0) continue;
            const float bb = a.bias[ch];
            float A[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if constexpr (W8) {
                    const float m0 = xch[(((ws) * NI + ni) * 4 + reg) * 64 + lane], m1 = xch[(((ws + NWS) * NI + ni) * 4 + reg) * 64 + lane];
                    A[reg] = fmaxf(m0, m1);
                } else {
                    const float v0 = relu_f(acc[0][ni][reg] + bb), v1 = relu_f(acc[1][ni][reg] + bb);
                    const float v2 = relu_f(acc[MI_N - 2][ni][reg] + bb), v3 = relu_f(acc[MI_N - 1][ni][reg] + bb);
                    A[reg] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                }
            }
            const float* Pc = a.P + ((size_t)eb * a.Cout + ch) * plane;
            float* Ec = a.E + ((size_t)eb * 2 * a.Cout + ch) * plane;
            float* Ec2 = Ec + (size_t)a.Cout * plane;
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                if (W8 && TW != 8 && pr != ch2) continue;  // 16- / 4-wide: wave ch2 of the pair stores pooled row ch2
                const int yo = pyo + pr;
                if (yo >= Ho || pxo >= Wo) continue;
                const size_t o = (size_t)yo * Wo + pxo;
                if (vec_ok) {
                    if constexpr (TW != 8) {
                        const f32x2 p2 = *reinterpret_cast<const f32x2*>(Pc + o);
                        const float A0 = A[2 * pr], A1 = A[2 * pr + 1];
                        *reinterpret_cast<f32x2*>(Ec + o) = (f32x2){relu_f(A0 - p2[0]), relu_f(A1 - p2[1])};
                        *reinterpret_cast<f32x2*>(Ec2 + o) = (f32x2){relu_f(p2[0] - A0), relu_f(p2[1] - A1)};
                    } else {
                        const f32x4 p4 = *reinterpret_cast<const f32x4*>(Pc + o);
                        *reinterpret_cast<f32x4*>(Ec + o) = (f32x4){relu_f(A[0] - p4[0]), relu_f(A[1] - p4[1]), relu_f(A[2] - p4[2]), relu_f(A[3] - p4[3])};
                        *reinterpret_cast<f32x4*>(Ec2 + o) = (f32x4){relu_f(p4[0] - A[0]), relu_f(p4[1] - A[1]), relu_f(p4[2] - A[2]), relu_f(p4[3] - A[3])};
                    }
                    continue;
                }
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) {
                    if (pxo + pc >= Wo) continue;
                    const float Av = A[(TW != 8) ? 2 * pr + pc : pc];
                    const float p = Pc[o + pc];
                    Ec[o + pc] = relu_f(Av - p);
                    Ec2[o + pc] = relu_f(p - Av);
                }
            }
        }
    } else if constexpr (EPI == EPI_CONVP) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int ch = nblk * NB + ni * 16 + col;
            if (ch >= a.Cout) continue;
            const float bb = a.bias[ch];
            const size_t base = ((size_t)eb * a.Cout + ch) * HW;
#pragma unroll
            for (int sgi = 0; sgi < NSEG; ++sgi) {
                const int gy = seg_row(EIG_SGI(sgi)), gx = seg_col(EIG_SGI(sgi));
                if (gy >= a.H || gx >= a.W) continue;
                const int pix = gy * a.W + gx;
                if (VEC && !a.frame && !a.E0) {  // layers > 0: P only, one aligned 16-byte store per segment
                    f32x4 v4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = relu_f(acc[EIG_SEG_MI(sgi, j)][ni][EIG_SEG_REG(sgi, j)] + bb);
                        if (a.clip) v = fminf(v, 1.0f);
                        v4[j] = v;
                    }
                    *reinterpret_cast<f32x4*>(a.Pout + base + pix) = v4;
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (gx + j >= a.W) continue;
                    float v = relu_f(acc[EIG_SEG_MI(sgi, j)][ni][EIG_SEG_REG(sgi, j)] + bb);
                    if (a.clip) v = fminf(v, 1.0f);
                    a.Pout[base + pix + j] = v;
                    if (a.frame) a.frame[(size_t)eb * a.frame_bstride + (size_t)ch * HW + pix + j] = (uint8_t)(int)(v * 255.0f);
                    if (a.E0) {
                        float x;
                        if (a.img) x = (float)a.img[base + pix + j] / 255.0f;
                        else if (a.requant) x = (float)(uint8_t)(int)(v * 255.0f) / 255.0f;
                        else x = v;
                        const size_t e = ((size_t)eb * 2 * a.Cout + ch) * HW + pix + j;
                        a.E0[e] = relu_f(x - v);
                        a.E0[e + (size_t)a.Cout * HW] = relu_f(v - x);
                    }
                }
            }
        }
    }
#undef EIG_SGI
#undef EIG_SEG_MI
#undef EIG_SEG_REG
    timeline_record(0);
}

// ConvP of the image layer (C = 1 or 3 channels in and out, K = 9 C <= 27): with 27 multiply-adds per output the matrix
// pipe is irrelevant and the MFMA kernel's fixed cost per block (prologue, one DMA round trip, epilogue for 216 MFMAs)
// is everything -- the operator is HBM-bound (read R_0 and the next frame, write P_0, the uint8 frame and E_0: 18 B per
// pixel and channel).  One thread per pixel, all C outputs; the haloed tile goes through LDS.  Arithmetic = the same fp32
// fma chain in (channel, ky, kx) order as the MFMA path (out-of-image taps multiply a staged 0, as there), then the
// epilogue of EPI_CONVP verbatim, so the results are bit-identical.
constexpr int P0_TX = 64, P0_TY = 8;
template <int C>
__global__ void __launch_bounds__(P0_TX * P0_TY) convp0_direct_kernel(const float* __restrict__ src, const float* __restrict__ wgt /*[C][C][3][3]*/,
                                                                     const ConvArgs a)
{
    __shared__ float tile[C][P0_TY + 2][P0_TX + 2];
    const int tx = threadIdx.x & (P0_TX - 1), ty = threadIdx.x / P0_TX;
    const int x0 = blockIdx.x * P0_TX, y0 = blockIdx.y * P0_TY, b = blockIdx.z;
    const int HW = a.H * a.W;
    const float* sb = src + (size_t)b * C * HW;
    for (int i = threadIdx.x; i < C * (P0_TY + 2) * (P0_TX + 2); i += P0_TX * P0_TY) {
        const int c = i / ((P0_TY + 2) * (P0_TX + 2));
        const int r = i - c * ((P0_TY + 2) * (P0_TX + 2));
        const int yy = r / (P0_TX + 2), xx = r - yy * (P0_TX + 2);
        const int gy = y0 + yy - 1, gx = x0 + xx - 1;
        tile[c][yy][xx] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? sb[(size_t)c * HW + gy * a.W + gx] : 0.0f;
    }
    __syncthreads();
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy >= a.H || gx >= a.W) return;
    const int pix = gy * a.W + gx;
#pragma unroll
    for (int o = 0; o < C; ++o) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc = fmaf(tile[c][ty + ky][tx + kx], wgt[((o * C + c) * 3 + ky) * 3 + kx], acc);
        const size_t base = ((size_t)b * C + o) * HW;
        float v = relu_f(acc + a.bias[o]);
        if (a.clip) v = fminf(v, 1.0f);
        a.Pout[base + pix] = v;
        if (a.frame) a.frame[(size_t)b * a.frame_bstride + (size_t)o * HW + pix] = (uint8_t)(int)(v * 255.0f);
        if (a.E0) {
            float x;
            if (a.img) x = (float)a.img[base + pix] / 255.0f;
            else if (a.requant) x = (float)(uint8_t)(int)(v * 255.0f) / 255.0f;
            else x = v;
            const size_t e = ((size_t)b * 2 * C + o) * HW + pix;
            a.E0[e] = relu_f(x - v);
            a.E0[e + (size_t)C * HW] = relu_f(v - x);
        }
    }
}

// ConvLSTM of the image layer (C = 1 or 3 channels, 12 gate outputs at most, K = 9 x 3C): like ConvP_0 it is bound by the fixed
// cost of an MFMA block (prologue, two DMA round trips and the epilogue around 108 MFMAs per wave), not by arithmetic.  One thread
// per pixel computes all 4 x C chains with the same fp32 fma order as the MFMA path -- source E_0 (channel, ky, kx), then h_0;
// out-of-image taps multiply a staged 0 -- adds the chain of the unpooled source (one fp32 addition, ConvArgs::acc_init: class
// (y & 1, x & 1) of source pixel (y / 2, x / 2), column = gate * 4 + channel as the packed MFMA layout has it) and runs the gate
// epilogue of EPI_LSTM_PACKED verbatim: bit-identical results.  T0: the step-0 operator (first half of E_0 only, no h_0).
// The FOUR GATE chains of an output channel run side by side (round 6): a staged value is read from LDS once per output channel and multiplied into four
// accumulators by the four gate weights of its tap, which lie together in the packed weight table wgt: [C outputs][K taps = (channel, ky, kx) over E_0 then h_0][4 gates]
// (T0: the table of the step-0 operator behind it) -- 243 LDS reads per pixel instead of 972 (the kernel was LDS-bound: 972 reads x 2 clocks per wave at 1024 waves per
// CU = 0.83 ms of its 1.02), one s_load_dwordx4 per tap.  (Measured before: all 12 chains from each staged value 1.21 ms, one chain at a time 1.07 ms, the 81 staged
// values held in registers 1.44 ms: profiles/r05_f_w4_timeline.txt.)
constexpr int L0_TX = 64, L0_TY = 8;   // (64 x 8 pixels per block: the haloed tile is 1.29 x the pixels it serves; 64 x 4: 1.55 x)
template <int C, bool T0>
__global__ void __launch_bounds__(L0_TX * L0_TY) lstm0_direct_kernel(const float* __restrict__ srcE, const float* __restrict__ srcH,
                                                                    const float* __restrict__ wgt, const ConvArgs a)
{
    constexpr int CE = T0 ? C : 2 * C, CH = T0 ? 0 : C;
    constexpr int K = (CE + CH) * 9;
    __shared__ float tile[CE + (CH ? CH : 1)][L0_TY + 2][L0_TX + 2];
    const int tx = threadIdx.x & (L0_TX - 1), ty = threadIdx.x / L0_TX;
    const int x0 = blockIdx.x * L0_TX, y0 = blockIdx.y * L0_TY, b = blockIdx.z;
    const int HW = a.H * a.W;
    constexpr int PLANE = (L0_TY + 2) * (L0_TX + 2);
    for (int i = threadIdx.x; i < (CE + CH) * PLANE; i += L0_TX * L0_TY) {
        const int c = i / PLANE;
        const int r = i - c * PLANE;
        const int yy = r / (L0_TX + 2), xx = r - yy * (L0_TX + 2);
        const int gy = y0 + yy - 1, gx = x0 + xx - 1;
        float v = 0.0f;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            v = (c < CE) ? srcE[((size_t)b * 2 * C + c) * HW + gy * a.W + gx] : srcH[((size_t)b * C + (c - CE)) * HW + gy * a.W + gx];
        tile[c][yy][xx] = v;
    }
    __syncthreads();
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy >= a.H || gx >= a.W) return;
    const int pix = gy * a.W + gx;
    const f32x4* const w4 = reinterpret_cast<const f32x4*>(wgt) + (T0 ? C * (3 * C * 9) : 0);   // (the step-0 table lies behind the full one)
    const int Hs = a.H >> 1, Ws = a.W >> 1;
    const size_t up_hw = (size_t)Hs * Ws;
    const float* up = a.acc_init ? a.acc_init + (((size_t)b * 4 + ((gy & 1) * 2 + (gx & 1))) * 16) * up_hw + (size_t)(gy >> 1) * Ws + (gx >> 1) : nullptr;
#pragma unroll
    for (int o = 0; o < C; ++o) {
        float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < CE + CH; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = tile[c][ty + ky][tx + kx];
                    const f32x4 w = w4[o * K + (c * 3 + ky) * 3 + kx];   // (wave-uniform address: a scalar load)
#pragma unroll
                    for (int g = 0; g < 4; ++g) z[g] = fmaf(v, w[g], z[g]);
                }
        if (up) {
#pragma unroll
            for (int g = 0; g < 4; ++g) z[g] = z[g] + up[(size_t)(g * 4 + o) * up_hw];
        }
        const float bi = a.bias[o], bf = a.bias[C + o], bc = a.bias[2 * C + o], bo = a.bias[3 * C + o];
        const size_t cbase = ((size_t)b * C + o) * HW;
        const size_t pbase = (size_t)o * HW;
        const size_t pstride = (size_t)C * HW;
        const float cold = a.c_state[cbase + pix];
        float cnew, hnew;
        lstm_cell(z[0], z[1], z[2], z[3], bi, bf, bc, bo, cold, a.peep[pbase + pix], a.peep[pstride + pbase + pix], a.peep[2 * pstride + pbase + pix], cnew, hnew);
        a.c_state[cbase + pix] = cnew;
        a.h_out[cbase + pix] = hnew;
    }
}

#ifdef EIG_ENGINE_UNIT   // (non-template kernels: defined in the engine's translation unit only -- this header is included by three)
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
#endif

}  // namespace eig
