// conv_wino4.h -- the 3x3 convolutions of PredNet layers >= 1 (ConvLSTM over E_l / unpooled R_{l+1} / h_l, ConvA, ConvP) as Winograd F(4x4, 3x3) on the fp32
// matrix pipe: 36 multiply-adds per channel and 4x4 output pixels where F(2x2, 3x3) (rounds 4-5: a sixteen-wave kernel, removed in round 6) needs 64 and the direct form 144.
// DESIGN.md section 3.1.
//
// Why: at fp32 only FEWER multiply-adds make the roll-out faster, and the parity half of the question was answered BEFORE this kernel was written
// (tests/studies/winograd_study.py --large-tiles, profiles/r05_c_winograd_large_tiles_study.json): on the operators that take it (layers >= 1) F(4x4, 3x3) in fp32 is
// indistinguishable from F(2x2, 3x3) -- the byte flips against the reference-order implementations are decided in the image layer, which stays direct.
// The canonical arithmetic of an operator that takes this form is stated operation by operation in oracle/eig_oracle.c (wino4_in1d / wino4_w1d / wino4_out1d,
// wino_input / wino_chains / wino_finish with m = 4); this kernel executes exactly those operations -- results are identical bit for bit:
//   input transform   1-D on six values, rows of the 6x6 patch first, then columns:  T0 = fmaf(4, d0, fmaf(-5, d2, d4));  p = fmaf(-4, d2, d4), q = fmaf(-4, d1, d3):
//                     T1 = p + q, T2 = p - q;  r = d4 - d2, s = d3 - d1:  T3 = fmaf(2, s, r), T4 = fmaf(-2, s, r);  T5 = fmaf(4, d1, fmaf(-5, d3, d5))
//   36 chains         M_pos[o][T] = fmaf(V_pos[c][T], U_pos[o][c], .) over (source, channel) ascending -- v_mfma_f32_16x16x4_f32
//   output transform  s = m1 + m2, d = m1 - m2, u = m3 + m4, w = m3 - m4:  Y0 = (m0 + s) + u, Y1 = fmaf(2, w, d), Y2 = fmaf(4, u, s), Y3 = fmaf(8, w, d) + m5;
//                     along nu in every row xi first (in-lane), then along xi
// then bias and the operator's epilogue (lstm_cell of conv_mfma.h / relu + 2x2 max-pool + error units / relu).  A ConvLSTM's unpooled source rides in the same
// chains: its 6x6 patch has rows / columns (a, b, b, c, c, d), p and q above are then the same operation on the same operands, T2 is an exact zero, and the positions
// with xi = 2 or nu = 2 are chains of exact zeros -- not built, not read, not multiplied: 25 of 36.
//
// Block = 16 x 32 output pixels of one image = 32 tiles of 4x4 = two REGIONS of 16 tiles (the upper / lower 8 x 32 pixels = 2 x 8 tiles; MFMA row r = 8 ty + tx:
// with the 40-float plane rows the sixteen 16-byte patch reads of a lane group fall on sixteen different bank slots) x NI 16-column N-tiles; TWELVE waves (three per
// SIMD, <= 168 VGPRs), one block per CU.  No transformed-input buffer --
//   wave (rg, xi), xi = 0..5, multiplies region rg for the six positions (xi, nu = 0..5): 6 NI accumulator tiles; it BUILDS its A operands itself: lane (q, col) holds
//     channel q of the K-block for tile col, reads the three or four patch rows that row xi of B^T d needs (per row three aligned 16-byte reads out of the
//     channel's plane: conflict-free, where 4-byte reads of the two end columns were 8-way conflicted and cost the first version 25 %), 12-18 operations for the
//     row, 12 for the column pass;
//   K-block = FOUR channels (one MFMA k-step): 6 NI MFMAs per wave in six chunks with the operand read of the next chunk, staging in slices between the chunks;
//   LDS (156 KB): plane ring 3 x 4 x [18 rows][40 floats] (one DMA instruction per wave and K-block) and U ring 3 x 36 KB ([36 pos][4 ch][16 cols][NI]; the packed
//     weights are [N-blocks][K-blocks of 4][36][4][16][NI], a position of a K-block = one contiguous KB = one LDS-DMA instruction, three per wave and K-block);
//     U(j) is fetched during K-block j - 2 and waited for at its end, plane(j) during j - 3 and waited for at the end of j - 2 (vmcnt(1)): both are visible to
//     everyone during K-block j - 1, whose last instructions read the first operands of j in front of the barrier -- the barrier never drains the matrix pipe;
//   output transform: along nu in-lane (c_xi,b: 4 x NI vectors), then the six waves of a region exchange their c rows through LDS in two rounds of 96 KB (N-tiles 0, 1
//     then 2, 3), and the outputs are finished in image order (see the epilogue).
// WALK (round 6): a block computes `nwalk` consecutive N-blocks of ITS tile (ConvArgs::nwalk, nparts = n_nblk / nwalk blocks per tile).  Everything per-lane is the same
// for every N-block of a tile; what changes is scalar (the weight descriptor, the output channel base).  LDS map: plane slots 0, 1 | U slot 0 | X = U slots 1, 2, plane slot
// 2, 12 KB.  Right behind the last K-block of N-block n the waves issue PART A of the prologue of n + 1 (U slab of K-block 0, planes of K-blocks 0, 1: 60 KB outside X) and
// only then run the finishing phase of n, whose exchange rounds live in X; part B (U slab of K-block 1, plane of K-block 2 -> X) follows behind the barrier at the top of
// n + 1 and is waited for at the end of its K-block 0: the DMA round trip, the block dispatch and the address set-up of a fresh block are hidden behind the exchange
// (profiles/r06_c_walk_v2_timeline.txt; DESIGN.md section 3.1 has the A/Bs and register tables of the three builds).
// TALL: a second block shape, 32 rows x 16 columns, for narrow maps (see w4_row below).
#pragma once
#include "wino_launch.h"

namespace eig {

constexpr int W4_UPOS = 256;                        // floats per position of a U slot: 4 ch x 16 cols x NI (NI = 3: 192 used)
constexpr int W4_U_FLOATS = W4_NPOS * W4_UPOS;      // 36 KB
constexpr int W4_NUS = 3, W4_NPS = 3;
// Two block shapes (template parameter TALL), chosen per operator by the MAP SIZE alone (eigen_engine.hip: the one that covers the map with fewer blocks, ties -> wide):
//   wide  16 rows x 32 columns: regions 8 x 32 (2 x 8 tiles, MFMA row r = 8 ty + tx), plane = 18 rows x 10 chunks (row stride 40 floats), 3 DMA parts per channel;
//   tall  32 rows x 16 columns: regions 16 x 16 (4 x 4 tiles, r = 4 ty + tx), plane = 34 rows x 6 chunks at a row stride of 7 chunks (28 floats: the sixteen 16-byte patch
//         reads of a lane group -- tiles (ty, tx) at chunk 28 ty + tx -- fall on sixteen different bank slots), 4 DMA parts per channel (waves 0-3 issue a second
//         instruction), no walk (its plane slots are 16 KB).  Same chains, same order: the shape of the block does not show in a single bit.  What it is for: the reference's
//         own 160 x 120 -- maps of 80 x 60 / 40 x 30 take 10 / 3 tall blocks where they take 12 / 4 wide ones.
constexpr int w4_row(bool tall) { return tall ? 28 : 40; }              // floats per plane row
constexpr int w4_ps(bool tall) { return tall ? 1024 : 768; }            // floats per channel plane (whole DMA parts of 256)
constexpr int w4_pslot(bool tall) { return W4_KC * w4_ps(tall); }       // 12 KB / 16 KB
// LDS map (floats).  wide: plane slots 0, 1 | U slot 0 | X = U slot 1, U slot 2, plane slot 2, 12 KB -- X (96 KB) is the exchange area of the finishing phase: what a
// walking block prefetches for its next N-block behind the last K-block (U slab of K-block 0, planes of K-blocks 0, 1: 60 KB) lies outside it.  tall (no walk): plane
// slots 0, 1, 2 | U slots 0, 1, 2, the exchange area over the U ring.
constexpr int w4_p(bool tall, int k) { return tall ? k * w4_pslot(true) : (k < 2 ? k * w4_pslot(false) : 2 * w4_pslot(false) + 3 * W4_U_FLOATS); }
constexpr int w4_u(bool tall, int k) { return tall ? 3 * w4_pslot(true) + k * W4_U_FLOATS : 2 * w4_pslot(false) + k * W4_U_FLOATS; }
constexpr int w4_x(bool tall) { return tall ? w4_u(true, 0) : w4_u(false, 1); }
constexpr int W4_X_FLOATS = W4_WAVES * 2 * 4 * 64 * 4;          // one exchange round = two N-tiles: [12 waves][2][4 e][64 lanes] 16-byte vectors = 96 KB
constexpr int wino4_lds_bytes() { return (w4_x(false) + W4_X_FLOATS) * 4; }   // 159744, both shapes (tall: 3 x 16 KB + 3 x 36 KB)
static_assert((w4_u(true, 2) + W4_U_FLOATS) * 4 == wino4_lds_bytes() && (w4_p(false, 2) + w4_pslot(false)) * 4 <= wino4_lds_bytes(), "LDS maps");
#ifndef EIG_W4_DEEPU
#define EIG_W4_DEEPU 1   // 0: measurement builds only (the half blocks' U slabs waited for one K-block earlier, as the full blocks do: profiles/r06_y_*)
#endif
#ifndef EIG_W4_DIAG
#define EIG_W4_DIAG 0   // measurement builds only (WRONG RESULTS): 1 no wait for the K loop's DMAs, 2 no barrier in the K loop, 4 no A-operand build, 8 no plane DMA, 16 no U DMA
#endif

// the 1-D transforms (oracle/eig_oracle.c: wino4_in1d / wino4_out1d); fmaf = one rounding (this translation unit is compiled with -ffp-contract=off)
__device__ __forceinline__ void w4_in1d(float d0, float d1, float d2, float d3, float d4, float d5, float* T)
{
    T[0] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
    const float p = fmaf(-4.0f, d2, d4), q = fmaf(-4.0f, d1, d3);
    T[1] = p + q; T[2] = p - q;
    const float r = d4 - d2, s = d3 - d1;
    T[3] = fmaf(2.0f, s, r); T[4] = fmaf(-2.0f, s, r);
    T[5] = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
}
// row XI of B^T d only: the patch rows in d[] are the ones w4_rows(XI) lists, in that order
template <int XI> __device__ __forceinline__ float w4_row(const float* d)
{
    if constexpr (XI == 0) return fmaf(4.0f, d[0], fmaf(-5.0f, d[1], d[2]));                 // rows 0, 2, 4
    else if constexpr (XI == 5) return fmaf(4.0f, d[0], fmaf(-5.0f, d[1], d[2]));            // rows 1, 3, 5
    else {                                                                                   // rows 1, 2, 3, 4
        if constexpr (XI == 1) return fmaf(-4.0f, d[1], d[3]) + fmaf(-4.0f, d[0], d[2]);
        else if constexpr (XI == 2) return fmaf(-4.0f, d[1], d[3]) - fmaf(-4.0f, d[0], d[2]);
        else if constexpr (XI == 3) return fmaf(2.0f, d[2] - d[0], d[3] - d[1]);
        else return fmaf(-2.0f, d[2] - d[0], d[3] - d[1]);
    }
}
constexpr int w4_nrows(int xi) { return (xi == 0 || xi == 5) ? 3 : 4; }
constexpr int w4_rowidx(int xi, int k) { return xi == 0 ? 2 * k : xi == 5 ? 2 * k + 1 : k + 1; }   // k-th patch row that row xi of B^T d reads
// the same row of an UNPOOLED patch: patch rows (0..5) are source rows s + (0, 1, 1, 2, 2, 3)
constexpr int w4_uprow(int i) { return (i + 1) >> 1; }

template <typename V4> __device__ __forceinline__ void w4_out1d(const V4& m0, const V4& m1, const V4& m2, const V4& m3, const V4& m4, const V4& m5, V4* Y)
{
    const V4 s = m1 + m2, d = m1 - m2, u = m3 + m4, w = m3 - m4;
    Y[0] = (m0 + s) + u;
    for (int e = 0; e < 4; ++e) { Y[1][e] = fmaf(2.0f, w[e], d[e]); Y[2][e] = fmaf(4.0f, u[e], s[e]); Y[3][e] = fmaf(8.0f, w[e], d[e]) + m5[e]; }
}

// HALF (round 6): a block of ONE region -- 8 rows x 32 columns, six waves (wave = row xi of the position grid; twelve with NSPLIT below), everything else as the wide shape
// with rg = 0.  For launches of at most one half block per CU (configs[0]: 10 to 40 blocks): their time is ONE block's time, and a half block has the CU's matrix pipe to
// itself for half the multiply-adds.  A choice by launch size (eigen_engine.hip), like the walk; the chains do not depend on it.
//
// PACK (round 6; a half block): for maps of 16 or 20 columns and 13 to 16 rows (4 x 4 or 5 x 4 tiles: the 20 x 15 top layer of the reference's own 160 x 120), which fill
// half / 62 % of a wide block.  The block's sixteen MFMA rows are either the 4 x 4 tiles of tile columns 0-3 of ONE image (a "main" block: MFMA row r = 4 ty + tx, the
// tall shape's region) or, for 5-column maps, tile column 4 of FOUR consecutive images (an "edge" block: r = 4 image + ty) -- five blocks per four images, every MFMA row
// a real tile (the launch's B main blocks first, then the edge blocks).  The plane container of a channel is [2][18 rows][7 chunks] in the tall shape's LDS map (row stride 28 floats): a main block fills [0] with its image's
// columns -4 .. 23; an edge block puts columns 12 .. 23 of image 2 p + b into chunks 3 b .. 3 b + 2 of [p] -- in both, the sixteen 16-byte patch reads of a lane group
// fall on sixteen different bank slots (a linear tile list with 5-tile rows cannot: profiles/r06_z_pmc_ref160_lds_linear_pack.txt, half of the LDS cycles were conflicts).  Sixteen
// (channel, part) DMA instructions per K-block over the block's waves; the A-operand base, the exchange slot and the output address of a lane come from its tile's
// (image, ty, tx).  Operators without an unpooled source, ConvLSTM / ConvP.
//
// NSPLIT (round 6; a half block of TWELVE waves): wave (ng, xi) multiplies the block's sixteen tiles for the six positions of row xi and the N-tiles 2 ng, 2 ng + 1 only --
// twelve MFMAs per K-block instead of 24, three waves on every SIMD instead of two on two of them: the six-wave half block leaves the matrix pipe of two SIMDs
// half empty.  Both waves of a row build the same A operands; the exchange is ONE round (wave (ng, xi) publishes its two N-tiles), and the sixteen finishing units of
// the region go over twelve waves (two for (0, xi < 4), one for the others).  64-column N-blocks, ConvLSTM / ConvP.
template <int NI, int EPI, bool TALL = false, bool HALF = false, bool PACK = false, bool NSPLIT = false>
__global__ void __launch_bounds__((HALF && !NSPLIT) ? W4_THREADS / 2 : W4_THREADS, 3) wino4_kernel(const ConvArgs a)
{
    static_assert(!NSPLIT || (HALF && NI == 4 && EPI != EPI_CONVA), "N-split: a half block, 64-column N-blocks, ConvLSTM / ConvP");
    constexpr int NIW = NSPLIT ? 2 : NI;             // N-tiles a wave multiplies
    // Half blocks: a K-block is 0.5 us of matrix work, less than the round trip of an LDS-DMA that misses the L2.  DEEPU: the U slab of K-block j, issued in j - 2, is waited
    // for at the end of j - 1 instead of j - 2 (the wait at the end of a K-block lets that K-block's own U and plane fetches stay in flight); the first B operand of a K-block
    // is then read behind the barrier in front of it.  Worth 1 % (profiles/r06_y_half_blocks_nsplit_pack.txt): what bounds a half block is the RATE of its U stream.
    constexpr bool DEEPU = HALF && EIG_W4_DEEPU;
    static_assert(!(TALL && HALF), "half blocks exist in the wide shape only");
    static_assert(!PACK || (HALF && EPI != EPI_CONVA), "packed tiles: a half block, ConvLSTM / ConvP");
    constexpr bool TGEO = TALL || PACK;              // plane rows of 28 floats, 16 KB plane slots, the tall LDS map
    static_assert(EPI == EPI_LSTM || EPI == EPI_CONVA || EPI == EPI_CONVP, "conv_wino4.h: ConvLSTM, ConvA, ConvP");
    static_assert(EPI != EPI_LSTM || NI == 4, "ConvLSTM: the four N-tiles are the four gates");
    static_assert(NI == 3 || NI == 4, "N-blocks of 48 or 64 columns");
    constexpr int U4 = wino4_u_floats(NI);
    constexpr int KC = W4_KC, PS = w4_ps(TGEO), W4_ROW = w4_row(TGEO);
    constexpr int W4_P0 = w4_p(TGEO, 0), W4_P1 = w4_p(TGEO, 1), W4_P2 = w4_p(TGEO, 2), W4_U0 = w4_u(TGEO, 0), W4_U1 = w4_u(TGEO, 1), W4_U2 = w4_u(TGEO, 2);
    constexpr int PK_ROWS = 18, PK_CX = 7, PK_IMG = PK_ROWS * PK_CX;   // PACK: chunks of one image's plane
    constexpr bool WALK = !TALL && !HALF;                     // (tall blocks: one N-block per block -- with 16 KB plane slots the prefetched part of a next N-block does not fit beside the exchange area)
    constexpr int RGH = TALL ? 16 : 8;               // rows of a region
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned long long tq_entry = EIG_TIMING ? __builtin_readcyclecounter() : 0;   // measurement builds (-DEIG_TIMING=1, scripts/timeline_wino4.py)
    float* const Pb = lds;                          // (slot offsets W4_P0 / W4_P1 / W4_P2, W4_U0 / W4_U1 / W4_U2 are absolute)
    float* const Ub = lds;
    float* const xb = lds + w4_x(TGEO);
    const int wv_o = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int rg_o = HALF ? 0 : wv_o & 1, xi_o = (HALF && !NSPLIT) ? wv_o : wv_o >> 1;
    const int ng_o = NSPLIT ? wv_o & 1 : 0;          // N-split: the wave's N-tile pair
    constexpr int NWAVES = (HALF && !NSPLIT) ? W4_WAVES / 2 : W4_WAVES;
    constexpr int UPW = 36 / NWAVES;                  // positions of a U slab a wave fetches: 3 (HALF: 6 -- its own row of the position grid)
    // The lane index, opaque to the compiler: every per-lane quantity of the K loop and of the finishing phase is derived from a FRESH copy at the point of use, so
    // that nothing per-lane is hoisted out of the walk and carried in registers (or scratch) across the phase that does not need it -- the walking kernel of round 5
    // spilled 42 registers that way, the first version of this one 9.
    auto lane_id = [&]() __attribute__((always_inline)) { int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(l)); return l; };

    const int tiles = a.tilesX * a.tilesY;            // (PACK: the tiles of ONE image, tilesX x tilesY of 4 x 4 pixels)
    const bool pk5 = PACK && a.tilesX == 5;           // (PACK: 5-column maps have edge blocks)
    const int ntile = PACK ? a.B + (pk5 ? (a.B + 3) >> 2 : 0) : a.B * tiles;   // (PACK: the main blocks of the B images, then the edge blocks of the image groups of four)
    const int xcd = blockIdx.x & 7, xi_ = blockIdx.x >> 3;
    // (divisions by launch constants through their host-side reciprocals, ConvArgs::mg: a run-time integer division is ~25 dependent scalar instructions:
    // profiles/r05_f_w4_timeline.txt)
    auto dv = [&](int x, int d, unsigned m) __attribute__((always_inline)) { return m ? (int)__umulhi((unsigned)x, m) : x / d; };
    // Block -> (tile, part): workgroup b runs on XCD b % 8; an XCD owns a contiguous range of tiles (tile_map = 0: tiles interleaved over the XCDs, A/B only) and the
    // nparts blocks of a tile are consecutive on it, so they stream ONE set of planes through that XCD's L2 while the 32 / nparts tiles in flight per N-block share its U slab.
    const int q0 = dv(xi_, a.nparts, a.mg[0]);
    const int part = xi_ - q0 * a.nparts;
    const int tlin = a.tile_map ? xcd * ((ntile + 7) >> 3) + q0 : q0 * 8 + xcd;
    if (tlin >= ntile) return;
    const int nwalk = WALK ? a.nwalk : 1, nb0 = part * nwalk;   // this block computes N-blocks nb0 .. nb0 + nwalk - 1 of its tile
    const bool pk_edge_o = pk5 && tlin >= a.B;
    const int eb = PACK ? (pk_edge_o ? 4 * (tlin - a.B) : tlin) : dv(tlin, tiles, a.mg[1]);   // (PACK: the block's (first) image)
    const int t_ = PACK ? 0 : tlin - eb * tiles;
    const int tyi = dv(t_, a.tilesX, a.mg[2]), txi = t_ - tyi * a.tilesX;
    const int y0 = PACK ? 0 : tyi * (TALL ? 32 : (HALF ? 8 : 16)), x0 = PACK ? 0 : txi * (TALL ? 16 : 32);
    const int HW = a.H * a.W;

    const bool up_fused = !PACK && EPI == EPI_LSTM && a.up_src != nullptr;
    const int nkb0 = a.src[0].C / KC;
    const int nkbu = up_fused ? (a.up_C / KC) : 0;
    const bool has1 = a.nsrc > 1;
    const int nkb_o = nkb0 + nkbu + (has1 ? (a.src[1].C / KC) : 0);
    const int up_lo_o = nkb0, up_hi_o = nkb0 + nkbu;
#define EIG4_WAITCNT(imm) do { __builtin_amdgcn_s_waitcnt(imm); asm volatile("" ::: "memory"); } while (0)
#define EIG4_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define EIG4_LDS_BARRIER() do { EIG4_WAITCNT(0xC07F); EIG4_BARRIER(); } while (0)   /* lgkmcnt(0) only: global loads / LDS-DMAs in flight are NOT waited for (__syncthreads would) */
#define EIG4_IS_UP(kb) ((kb) >= up_lo && (kb) < up_hi)

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // ---- plane fetch (every wave: channel wv / 3 of the K-block, part wv % 3 of its plane): lane = 16-byte chunk of the 18 x 10-chunk haloed plane (unpooled source:
    // 10 rows x 6 chunks at half resolution, row stride 10 chunks); rows / chunks outside the image are out of the descriptor's range through a saturating add = zeros
    constexpr unsigned W4_POSB = 4 * 16 * NI * 4;   // bytes per position of a packed K-block
    // One N-block of the walk, as a lambda called from the loop below.  (Written as the loop's body the same code kept 7 VGPRs in scratch -- K-loop lane invariants reloaded
    // inside the K loop -- and 30 SGPRs in VGPR lanes; as a lambda: 158 VGPRs, nothing in scratch.  Called three times with compile-time indices, i.e. the walk as
    // straight-line code: 157 VGPRs and 0.6 % SLOWER, profiles/r06_m_walk_clean_ab.txt.)
    auto walk_body = [&](const int it) __attribute__((always_inline)) {
    const int nblk = nb0 + it;
    // An opaque zero: the scalars of a phase are derived from the wave index, the K-block counts and the tile coordinates offset by it INSIDE the walk, so that the
    // compiler does not hoist the LDS addresses of every DMA instruction, the source selection of the prologue and of the K loops' six role variants and the address
    // arithmetic of the finishing phase out of the walk and keep them live in SGPRs throughout (first walking build: 106 SGPRs + 88 spilled to VGPR lanes, 10 VGPRs in
    // scratch, reloads inside the K loop; the non-walking kernel: 88 / 0 / 0).
    int zs = 0;
    asm volatile("" : "+s"(zs));
    const int wv = wv_o + zs, rg = rg_o + zs, xi = xi_o + zs, ng = ng_o + zs, nkb = nkb_o + zs, up_lo = up_lo_o + zs, up_hi = up_hi_o + zs;
    const int pch = HALF ? wv >> 1 : wv / 3, ppart = HALF ? wv & 1 : wv - pch * 3;   // this wave's plane DMA: channel, part of 64 chunks
    const bool has_plane = !(NSPLIT && !PACK) || wv_o < 8;   // (N-split: the eight (channel, part) pairs of a half block's plane go to waves 0-7; wave-uniform)
    const int eb_i = eb + zs, y0_i = y0 + zs, x0_i = x0 + zs;
    const bool pk_edge = (pk_edge_o ? 1 : 0) + zs != 0;
    const unsigned long long sb0 = (unsigned long long)(a.src[0].ptr + (size_t)eb_i * a.src[0].Ct * HW);
    const unsigned long long sb1 = has1 ? (unsigned long long)(a.src[1].ptr + (size_t)eb_i * a.src[1].Ct * HW) : sb0;
    // (PACK: the descriptor's range takes in the four images of an edge block -- lanes whose image does not exist carry the out-of-range offset)
    const int ist0 = a.src[0].Ct * HW * 4, ist1 = has1 ? a.src[1].Ct * HW * 4 : ist0;   // bytes from an image to the next
    const int sz0 = a.src[0].C * HW * 4 + (PACK ? 3 * ist0 : 0), sz1 = has1 ? a.src[1].C * HW * 4 + (PACK ? 3 * ist1 : 0) : sz0;
    const int Hh = a.H >> 1, Wh = a.W >> 1, HWh = Hh * Wh;
    const unsigned long long sbu = up_fused ? (unsigned long long)(a.up_src + (size_t)eb_i * a.up_C * HWh) : sb0;
    const int szu = up_fused ? a.up_C * HWh * 4 : sz0;
    const int lane = lane_id();
    const int q = lane >> 4, col = lane & 15;
    // wide: 18 rows x 10 chunks (unpooled source: 10 rows x 6 chunks at half resolution); tall: 34 rows x 6 chunks at a row stride of 7 (18 rows x 4 chunks), four parts
    // per channel: waves 0-3 fetch part 3 of channel wv with a second instruction (roff2 / uoff2)
    constexpr int CPR = TALL ? 7 : 10, CREAL = TALL ? 6 : 10, NCHK = TALL ? 34 * 7 : (HALF ? 100 : 180), UROWS = TALL ? 18 : (HALF ? 6 : 10), UCHK = TALL ? 4 : 6;
    auto plane_offsets = [&](int part, int& ro, int& uo) __attribute__((always_inline)) {
        const int c = lane + 64 * part;
        const int row = c / CPR, cx = c - row * CPR;
        const int gy = y0_i - 1 + row, gx = x0_i - 4 + 4 * cx;
        ro = (c < NCHK && cx < CREAL && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (gy * a.W + gx) * 4 : -1;
        const int hy = (y0_i >> 1) - 1 + row, hx = (x0_i >> 1) - 4 + 4 * cx;
        uo = (row < UROWS && cx < UCHK && hy >= 0 && hy < Hh && hx >= 0 && hx < Wh) ? (hy * Wh + hx) * 4 : -1;
    };
    // PACK: wave wv issues the (channel, part) pairs idx = wv + NWAVES k < 16 (k = 0, 1, 2): channel idx >> 2, part idx & 3; chunk c = lane + 64 part of [2][18][7]
    int pk_ro[3] = {-1, -1, -1};
    unsigned pk_mul[3] = {0u, 0u, 0u};   // the chunk's image, relative to the block's first
    const int pk_n = (wv_o < 4 ? 2 : 1) + (NSPLIT ? 0 : 1);   // (wave-uniform: 16 (channel, part) pairs over the block's six / twelve waves)
    if constexpr (PACK) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = lane + 64 * ((wv_o + NWAVES * k) & 3);
            const int p = c >= PK_IMG ? 1 : 0, c1 = c - PK_IMG * p;
            const int row = c1 / PK_CX, slot = c1 - row * PK_CX;
            const int b = slot >= 3 ? (slot >= 6 ? 2 : 1) : 0;                       // (edge: three chunks per image, slot 6 unused)
            const int irel = pk_edge ? 2 * p + b : 0, cx = pk_edge ? 4 + slot - 3 * b : slot;
            const int gy = row - 1, gx = 4 * cx - 4;
            const bool ok = (pk_edge ? (c < 2 * PK_IMG && b < 2) : c < PK_IMG) && eb_i + irel < a.B && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            pk_ro[k] = ok ? (gy * a.W + gx) * 4 : -1;
            pk_mul[k] = ok ? (unsigned)irel : 0u;
        }
    }
    int roff, uoff, roff2 = -1, uoff2 = -1;
    plane_offsets(ppart, roff, uoff);
    // second plane instruction of some waves: tall -- part 3 of channel wv (waves 0-3); half -- channel 3, part wv & 1 (waves 0, 1: eight parts over six waves)
    const bool two_parts = (TALL && wv_o < 4) || (HALF && !PACK && !NSPLIT && wv_o < 2);   // (wave-uniform)
    if (TALL) plane_offsets(3, roff2, uoff2);
    if (HALF) { roff2 = roff; uoff2 = uoff; }
    const int ch2 = HALF ? 3 : wv, part2 = HALF ? ppart : 3;
    auto dma_plane_at = [&](int jj, int slot_off) __attribute__((always_inline)) {   // (prologue) this wave's plane DMA of K-block min(jj, nkb - 1) -> the plane slot at float offset slot_off
        const int j = jj < nkb ? jj : nkb - 1;
        const bool up = EIG4_IS_UP(j);
        const bool s1 = j >= up_hi;
        // (arithmetic selection: a select between captured variables becomes a select between their ADDRESSES -- a table in scratch memory)
        const unsigned long long mu = 0ull - (unsigned long long)up, m1 = 0ull - (unsigned long long)s1;
        const unsigned long long u = sb0 + ((sb1 - sb0) & m1) + ((sbu - sb0) & mu);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        const int sz = sz0 + ((sz1 - sz0) & (int)m1) + ((szu - sz0) & (int)mu);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(sz), 0x00020000);
        const int base = (up_lo & (int)mu) + (up_hi & (int)m1), hw = HW + ((HWh - HW) & (int)mu);
        if constexpr (PACK) {
            const unsigned ist = (unsigned)(ist0 + ((ist1 - ist0) & (int)m1));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k >= pk_n) break;
                const int idx = wv_o + NWAVES * k;
                const unsigned coffk = (unsigned)((j - base) * KC + (idx >> 2)) * (unsigned)(hw * 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Pb + slot_off + (idx >> 2) * PS + (idx & 3) * 256), 16,
                                                         (int)__builtin_elementwise_add_sat((unsigned)pk_ro[k] + pk_mul[k] * ist, coffk), 0, 0, 0);
            }
            return;
        }
        if (!has_plane) return;
        const unsigned coff = (unsigned)((j - base) * KC + pch) * (unsigned)(hw * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Pb + slot_off + pch * PS + ppart * 256), 16,
                                                 (int)__builtin_elementwise_add_sat((unsigned)roff + (((unsigned)uoff - (unsigned)roff) & (unsigned)mu), coff), 0, 0, 0);
        if (two_parts) {
            const unsigned coff2 = (unsigned)((j - base) * KC + ch2) * (unsigned)(hw * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Pb + slot_off + ch2 * PS + part2 * 256), 16,
                                                     (int)__builtin_elementwise_add_sat((unsigned)roff2 + (((unsigned)uoff2 - (unsigned)roff2) & (unsigned)mu), coff2), 0, 0, 0);
        }
    };
    // "the U fetch of this K-block has landed, its plane fetch (one instruction, two for waves 0-3 of a tall block) may stay in flight"
    auto wait_u = [&]() __attribute__((always_inline)) {
        const int npl = PACK ? pk_n : (two_parts ? 2 : (has_plane ? 1 : 0));   // plane instructions of this wave per K-block (wave-uniform)
        switch (npl + (DEEPU ? UPW : 0)) {   // (DEEPU: ... and the U fetch of this K-block)
            case 0: EIG4_WAITCNT(0x0F70); break;
            case 1: EIG4_WAITCNT(0x0F71); break;
            case 2: EIG4_WAITCNT(0x0F72); break;
            case 3: EIG4_WAITCNT(0x0F73); break;
            case 4: EIG4_WAITCNT(0x0F74); break;
            case 5: EIG4_WAITCNT(0x0F75); break;
            case 6: EIG4_WAITCNT(0x0F76); break;
            case 7: EIG4_WAITCNT(0x0F77); break;
            case 8: EIG4_WAITCNT(0x0F78); break;
            default: EIG4_WAITCNT(0x0F79); break;
        }
    };
    // ---- U fetch (every wave: positions 3 wv .. 3 wv + 2 of the next K-block in line): one position of a packed 4-channel K-block = one contiguous KB (NI = 3: 768 B).
    // The scalar offset runs along the packed K-blocks (no index arithmetic in the K loop; the last two fetches read the next N-block's first K-blocks or the
    // buffer's padding into slots nobody reads).
    const int uvo = (NI == 4 || lane < 48) ? lane * 16 : -1;
    // The prologue of N-block nb in two parts.  A: the U slab of K-block 0 and the planes of K-blocks 0, 1 -- what the first operand build and K-block 0 read; a walking
    // block issues it for N-block nb + 1 right behind the last K-block of nb, ahead of nb's finishing phase (whose exchange area X it lies outside of).  B: the U slab of
    // K-block 1 and the plane of K-block 2, into X -- behind the barrier at the top of the next N-block (they are waited for at the end of K-block 0).
    auto prologue_a = [&](int nb) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wpk + (size_t)nb * nkb * U4), 0, (nkb + 1) * U4 * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < UPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Ub + W4_U0 + (UPW * wv + i) * W4_UPOS), 16, uvo, (int)((unsigned)(UPW * wv + i) * W4_POSB), 0, 0);
        dma_plane_at(0, W4_P0);
        dma_plane_at(1, W4_P1);
    };
    auto prologue_b = [&](int nb) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wpk + (size_t)nb * nkb * U4), 0, (nkb + 1) * U4 * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < UPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Ub + W4_U1 + (UPW * wv + i) * W4_UPOS), 16, uvo,
                                                     (int)((unsigned)U4 * 4 + (unsigned)(UPW * wv + i) * W4_POSB), 0, 0);
        dma_plane_at(2, W4_P2);
    };

    // ---- A operands: lane (q, col) -> channel q of the K-block, tile (ty, tx) of region rg (rows RGH rg .. of the block): wide (col >> 3, col & 7), MFMA row r = 8 ty + tx;
    // tall (col >> 2, col & 3), r = 4 ty + tx
    // PACK: MFMA row col = main: tile (col >> 2, col & 3) of the block's image, at chunk tx of row 4 ty of [0]; edge: tile (ty = col & 3, 4) of image col >> 2 = 2 p + b,
    // at chunk 3 b of row 4 ty of [p]
    const int pk_ir = pk_edge ? col >> 2 : 0, pk_ty = pk_edge ? col & 3 : col >> 2;
    const int t_ty = PACK ? 0 : (TALL ? col >> 2 : col >> 3), t_tx = PACK ? (pk_edge ? 3 * (pk_ir & 1) : col & 3) : (TALL ? col & 3 : col & 7);
    const int pk_row = PACK ? PK_ROWS * (pk_ir >> 1) + 4 * pk_ty : 0;   // (first plane row of the tile's patch in [2][18] rows)
    const float* const pbase_n = Pb + q * PS + (RGH * rg + 4 * t_ty + pk_row) * W4_ROW + 4 * t_tx;        // patch row 0, the aligned chunk that holds patch column 0 in its last float: columns 0 .. 5 = floats 3 .. 8
    const float* const pbase_u = Pb + q * PS + ((RGH / 2) * rg + 2 * t_ty) * W4_ROW + 2 * t_tx + 3;    // source row s, column s of an unpooled patch (half-resolution plane)
    const int b_off = xi * 6 * W4_UPOS + (q * 16 + col) * NI + 2 * ng;   // U[pos = 6 xi + nu][ch = q][col][0 .. NI)  (N-split: N-tiles 2 ng, 2 ng + 1)

    unsigned long long tq_setup = tq_entry;
    if (it == 0) {
        if (EIG_TIMING) tq_setup = __builtin_readcyclecounter();
        prologue_a(nb0);
        EIG4_WAITCNT(0x0F70);
    }
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wpk + (size_t)nblk * nkb * U4), 0, (nkb + 1) * U4 * 4, 0x00020000);
    unsigned u_so = (unsigned)(UPW * wv) * W4_POSB + 2u * (unsigned)U4 * 4u;   // (K-blocks 0, 1 came with the prologue)
    auto dma_u = [&](int slot_off) __attribute__((always_inline)) {   // -> the U slot at float offset slot_off
#pragma unroll
        for (int i = 0; i < UPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(Ub + slot_off + (UPW * wv + i) * W4_UPOS), 16, uvo,
                                                     (int)(u_so + (unsigned)i * W4_POSB), 0, 0);
        u_so += (unsigned)U4 * 4;
    };
    // accumulators: position (xi, nu), N-tile ni
    f32x4 acc[6][NIW];
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) acc[p][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (a fresh block waited for part A of its prologue above; a walking block's waves each waited for theirs -- vmcnt(0) -- in front of the previous finishing phase's
    // gate loads.)  Behind this barrier nobody reads the exchange area any more: part B goes into it.
    EIG4_BARRIER();
    prologue_b(nblk);
    const unsigned long long tq_k0 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
    unsigned long long tq_k1 = 0, tq_x = 0, tq_y = 0;
    // [entry, set-up done, K loop start, K loop end, first exchange barrier passed, y ready (gates start), exit, HW_ID | XCC_ID << 32] per (block, N-block of the walk, wave)
    auto timeline = [&]() __attribute__((always_inline)) {
        if (EIG_TIMING && a.dbg && lane_id() == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* dd = a.dbg + (((size_t)blockIdx.x * nwalk + it) * NWAVES + wv) * 8;
            const unsigned long long tq_end = __builtin_readcyclecounter();
            dd[0] = tq_entry; dd[1] = tq_setup; dd[2] = tq_k0; dd[3] = tq_k1; dd[4] = tq_x; dd[5] = tq_y; dd[6] = tq_end;
            dd[7] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
            tq_entry = tq_end;
        }
    };

    // The K loops exist once per xi (role_tag) behind ONE wave-uniform branch: the patch rows a wave reads and its row combination are compile-time constants of xi,
    // and so are the positions an unpooled-source K-block skips.
    auto kloops = [&](auto role_tag) __attribute__((always_inline)) {
        constexpr int XI = decltype(role_tag)::value;
        float v[6];            // A operands of the current K-block (built at the end of the previous one)
        float bq[NIW];         // B operand of the current K-block's first chunk, read before the barrier in front of it
        // Rings of three, all in the phase kb % 3, as ROTATING float offsets (one set of moves per K-block; slot counters cost an add, a compare, a select and a multiply
        // each, in every K-block's instruction stream): U slots of K-blocks kb, kb + 1 and of the fetch (kb + 2); plane slots of the fetch (kb + 3 -> slot kb % 3), of the
        // patch rows read next (kb + 1) and the third
        int uo0 = W4_U0, uo1 = W4_U1, uo2 = W4_U2;
        int po0 = W4_P0, po1 = W4_P1, po2 = W4_P2;
        // fetch cursor of the planes (K-block pj = kb + 3, clamped to the last one): the descriptor of its source as scalars (a mutable descriptor OBJECT ends in
        // scratch memory), the byte offset of this wave's channel, the slot
        int pj = 0, psz = 0, pbound = 0;
        unsigned pcoff = 0, plo = 0, phi = 0, phw4 = 0, pist = 0;
        bool pup = false;
        auto plane_source = [&](int j) __attribute__((always_inline)) {   // (re)position the cursor on K-block j: at the start and where a source begins
            const bool up = EIG4_IS_UP(j);
            const bool s1 = j >= up_hi;
            const unsigned long long mu = 0ull - (unsigned long long)up, m1 = 0ull - (unsigned long long)s1;
            const unsigned long long u = sb0 + ((sb1 - sb0) & m1) + ((sbu - sb0) & mu);
            plo = __builtin_amdgcn_readfirstlane((unsigned)u); phi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            psz = __builtin_amdgcn_readfirstlane(sz0 + ((sz1 - sz0) & (int)m1) + ((szu - sz0) & (int)mu));
            const int base = (up_lo & (int)mu) + (up_hi & (int)m1);
            phw4 = (unsigned)((HW + ((HWh - HW) & (int)mu)) * 4);
            pcoff = (unsigned)((j - base) * KC + (PACK ? 0 : pch)) * phw4;
            if constexpr (PACK) pist = (unsigned)(ist0 + ((ist1 - ist0) & (int)m1));
            pup = up; pj = j;
            pbound = j < up_lo ? up_lo : (j < up_hi ? up_hi : nkb);   // where the next source begins (or the K-blocks end)
        };
        auto dma_plane = [&]() __attribute__((always_inline)) {   // this wave's plane DMA of K-block pj -> the slot at po0, then advance the cursor
            const unsigned o = (unsigned)roff + (((unsigned)uoff - (unsigned)roff) & (0u - (unsigned)pup));
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)phi << 32) | plo), 0, psz, 0x00020000);
            if constexpr (PACK) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (k >= pk_n) break;
                    const int idx = wv + NWAVES * k;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (__attribute__((address_space(3))) void*)(Pb + po0 + (idx >> 2) * PS + (idx & 3) * 256), 16,
                                                             (int)__builtin_elementwise_add_sat((unsigned)pk_ro[k] + pk_mul[k] * pist, pcoff + (unsigned)(idx >> 2) * phw4), 0, 0, 0);
                }
            } else if (has_plane) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (__attribute__((address_space(3))) void*)(Pb + po0 + pch * PS + ppart * 256), 16,
                                                     (int)__builtin_elementwise_add_sat(o, pcoff), 0, 0, 0);
            }
            if (two_parts) {
                const unsigned o2 = (unsigned)roff2 + (((unsigned)uoff2 - (unsigned)roff2) & (0u - (unsigned)pup));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (__attribute__((address_space(3))) void*)(Pb + po0 + ch2 * PS + part2 * 256), 16,
                                                         (int)__builtin_elementwise_add_sat(o2, pcoff + (unsigned)(ch2 - pch) * phw4), 0, 0, 0);
            }
            if (__builtin_expect(pj + 1 == pbound, 0)) { if (pbound < nkb) plane_source(pj + 1); }   // (at the end the cursor stays on the last K-block)
            else { ++pj; pcoff += KC * phw4; }
        };
        plane_source(3 < nkb ? 3 : nkb - 1);   // (slot 0 = 3 % 3)
        // Row XI of B^T d of the next K-block's patch in TWO phases (half the registers in flight; the same operations in the same order as w4_row):
        //   phase 0 reads the patch rows of the INNER operation (xi 1, 2: rows 2, 4 -> p = fmaf(-4, d2, d4); xi 3, 4: r = d4 - d2; xi 0: fmaf(-5, d2, d4); xi 5: fmaf(-5, d3, d5)),
        //   phase 1 reads the others and finishes (xi 1 / 2: p +- fmaf(-4, d1, d3); xi 3 / 4: fmaf(+-2, d3 - d1, r); xi 0: fmaf(4, d0, .); xi 5: fmaf(4, d1, .)).
        float pr[2][6];        // the two patch rows of a phase
        float t[6];            // phase 0: the inner operation; phase 1: row XI of B^T d
        int rso = 0;           // LDS offset of the plane slot the current K-block reads its patch rows from
        bool rup = false;
        // k-th patch row (of w4_rowidx(XI, .)) a phase reads: phase 0 -> k = 1, 3 (xi 0, 5: 1, 2), phase 1 -> k = 0, 2 (xi 0, 5: 0)
        auto read_rows = [&](int ph) __attribute__((always_inline)) {
            if (EIG_W4_DIAG & 4) return;
            constexpr bool END = XI == 0 || XI == 5;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = ph == 0 ? (END ? 1 + j : 1 + 2 * j) : (END ? 0 : 2 * j);
                if (ph == 1 && END && j == 1) continue;
                if (rup) {   // rows s + (0, 1, 1, 2, 2, 3), columns likewise: four distinct source values per row
                    const float* const pl = pbase_u + rso + w4_uprow(w4_rowidx(XI, k)) * W4_ROW;
                    const float s0 = pl[0], s1_ = pl[1], s2 = pl[2], s3 = pl[3];
                    pr[j][0] = s0; pr[j][1] = s1_; pr[j][2] = s1_; pr[j][3] = s2; pr[j][4] = s2; pr[j][5] = s3;
                } else {
                    const f32x4* const pl = reinterpret_cast<const f32x4*>(pbase_n + rso + w4_rowidx(XI, k) * W4_ROW);
                    const f32x4 c0 = pl[0], c1 = pl[1], c2 = pl[2];   // three aligned 16-byte reads: floats 0 .. 11 of which 3 .. 8 are the patch row
                    pr[j][0] = c0[3]; pr[j][1] = c1[0]; pr[j][2] = c1[1]; pr[j][3] = c1[2]; pr[j][4] = c1[3]; pr[j][5] = c2[0];
                }
            }
        };
        auto rows_begin = [&](int slot_off, bool up) __attribute__((always_inline)) {   // the plane slot of the K-block whose patch rows are read, then phase 0's reads
            rso = slot_off; rup = up;
            read_rows(0);
        };
        auto rows_mid = [&]() __attribute__((always_inline)) {     // the inner operation from phase 0's rows, then phase 1's reads
            if (!(EIG_W4_DIAG & 4)) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    if constexpr (XI == 0 || XI == 5) t[c] = fmaf(-5.0f, pr[0][c], pr[1][c]);
                    else if constexpr (XI == 1 || XI == 2) t[c] = fmaf(-4.0f, pr[0][c], pr[1][c]);
                    else t[c] = pr[1][c] - pr[0][c];
                }
            }
            read_rows(1);
        };
        auto rows_end = [&]() __attribute__((always_inline)) {     // row XI of B^T d
            if (EIG_W4_DIAG & 4) return;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if constexpr (XI == 0 || XI == 5) t[c] = fmaf(4.0f, pr[0][c], t[c]);
                else if constexpr (XI == 1) t[c] = t[c] + fmaf(-4.0f, pr[0][c], pr[1][c]);
                else if constexpr (XI == 2) t[c] = t[c] - fmaf(-4.0f, pr[0][c], pr[1][c]);
                else if constexpr (XI == 3) t[c] = fmaf(2.0f, pr[1][c] - pr[0][c], t[c]);
                else t[c] = fmaf(-2.0f, pr[1][c] - pr[0][c], t[c]);
            }
        };
        auto build_cols = [&]() __attribute__((always_inline)) {   // ... then the 1-D transform along the row: the six A operands
            if (EIG_W4_DIAG & 4) { for (int c = 0; c < 6; ++c) v[c] = 1.0f; return; }
            w4_in1d(t[0], t[1], t[2], t[3], t[4], t[5], v);
        };
        auto read_b = [&](int slot_off, int nu, float* dst) __attribute__((always_inline)) {
            const float* const bsrc = Ub + slot_off + b_off + nu * W4_UPOS;
            if constexpr (NSPLIT) {
                const f32x2 b2 = *reinterpret_cast<const f32x2*>(bsrc);
                dst[0] = b2[0]; dst[1] = b2[1];
            } else if constexpr (NI == 4) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bsrc);
                dst[0] = b4[0]; dst[1] = b4[1]; dst[2] = b4[2]; dst[3] = b4[3];
            } else {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) dst[ni] = bsrc[ni];
            }
        };
        // One K-block.  kind_tag: 0 full, 1 unpooled source, 2 run time; nk_tag: kind of K-block kb + 1 (whose patch rows are read now), same codes; last_tag: the last
        // K-block (nothing to stage).  On entry v and bq hold the A operands and the first B operand of this K-block: the first instruction behind the barrier is an MFMA,
        // and the staging work sits in slices BETWEEN the chunks -- the U fetch behind chunk 0, the plane fetch behind chunk 1, the patch rows of K-block kb + 1 and
        // the A operands of kb + 1 behind the last chunks (see slice below).
        auto kiter = [&](const int kb, auto kind_tag, auto nk_tag, auto last_tag, auto first_tag) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_tag)::value;
            constexpr int NK = decltype(nk_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value || (DEEPU && !LAST);   // K-block 0 (DEEPU: every K-block): the U slab of K-block 1 (prologue part B) is only known to have landed at this K-block's END -- its first operand is read behind the barrier
            constexpr bool UP = KIND == 1;   // (run-time kind = the last K-block of all: an unpooled-source one there runs the full body on its exact-zero operands -- fma(0, u, M) = M)
            constexpr bool IDLE = UP && XI == 2;   // nothing to multiply
            constexpr int NCH = IDLE ? 0 : (UP ? 5 : 6);   // chunks: nu = 0, 1, (2,) 3, 4, 5
            float bv[2][NIW];
            // slices of staging work behind the chunks: the U fetch behind chunk 0, the plane fetch behind chunk 1, the patch rows of K-block kb + 1 and row XI of
            // B^T d in two phases behind chunks NCH - 4 .. NCH - 2 (their registers are needed late), the column pass behind the last chunk
            auto slice = [&](int i) __attribute__((always_inline)) {
                if constexpr (!LAST) {
                    if (i == 0) { if (!(EIG_W4_DIAG & 16)) dma_u(uo2); }
                    if (i == 1) { if (!(EIG_W4_DIAG & 8)) dma_plane(); }
                    if (i == NCH - 4) rows_begin(po1, NK == 2 ? EIG4_IS_UP(kb + 1) : NK == 1);
                    if (i == NCH - 3) rows_mid();
                    if (i == NCH - 2) rows_end();
                    if (i == NCH - 1) build_cols();
                }
            };
            if constexpr (IDLE) {
                if constexpr (!LAST) {
                    if (!(EIG_W4_DIAG & 16)) dma_u(uo2);
                    if (!(EIG_W4_DIAG & 8)) dma_plane();
                    rows_begin(po1, NK == 2 ? EIG4_IS_UP(kb + 1) : NK == 1); rows_mid(); rows_end();
                    if constexpr (!FIRST) read_b(uo1, 0, bq);
                    build_cols();
                }
            } else {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int nu = (UP && i >= 2) ? i + 1 : i;
                    if (i + 1 < NCH) {
                        const int nu1 = (UP && i + 1 >= 2) ? i + 2 : i + 1;
                        read_b(uo0, nu1, bv[(i + 1) & 1]);
                    } else if constexpr (!LAST && !FIRST) read_b(uo1, 0, bq);
                    const float* const b = i == 0 ? bq : bv[i & 1];
#pragma unroll
                    for (int ni = 0; ni < NIW; ++ni) acc[nu][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[nu], b[ni], acc[nu][ni], 0, 0, 0);
                    slice(i);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            { const int t0 = uo0; uo0 = uo1; uo1 = uo2; uo2 = t0; }
            { const int t0 = po0; po0 = po1; po1 = po2; po2 = t0; }
            // the U fetch of this K-block has landed (its plane fetch may stay in flight: it is read two K-blocks from now); after the last K-block: everything
            if (LAST || (EIG_W4_DIAG & (8 | 16))) EIG4_WAITCNT(0x0F70);
            else if (!(EIG_W4_DIAG & 2)) { if (!(EIG_W4_DIAG & 1)) wait_u(); }
            if (!(EIG_W4_DIAG & 2) || LAST) EIG4_BARRIER();
            if constexpr (FIRST) read_b(uo0, 0, bq);
        };
        const std::false_type nl{};
        const std::integral_constant<int, 2> rt{};
        int kb = 0;
        rows_begin(W4_P0, EIG4_IS_UP(0)); rows_mid(); rows_end();
        read_b(W4_U0, 0, bq);
        build_cols();
        EIG4_WAITCNT(0xC07F);
        EIG4_BARRIER();   // (every wave has read plane 0 out of its slot before anyone's K-block 0 fetches into it)
        // K-blocks [kb, end) of one kind; the LAST K-block of all is left out.  A K-block reads the patch rows of the next one: same kind except at the end of a run.
        auto run = [&](const int end, auto kind_tag) __attribute__((always_inline)) {
            for (; kb + 1 < end; ++kb) kiter(kb, kind_tag, kind_tag, nl, nl);
            if (kb + 1 == end && end < nkb) { kiter(kb, kind_tag, rt, nl, nl); ++kb; }
        };
        kiter(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, nl, std::true_type{});   // (source 0 has at least two K-blocks: Cin % 8 == 0)
        kb = 1;
        run(up_lo, std::integral_constant<int, 0>{});
        run(up_hi, std::integral_constant<int, 1>{});
        run(nkb, std::integral_constant<int, 0>{});
        kiter(nkb - 1, rt, rt, std::true_type{}, nl);
    };
    switch (xi) {
        case 0: kloops(std::integral_constant<int, 0>{}); break;
        case 1: kloops(std::integral_constant<int, 1>{}); break;
        case 2: kloops(std::integral_constant<int, 2>{}); break;
        case 3: kloops(std::integral_constant<int, 3>{}); break;
        case 4: kloops(std::integral_constant<int, 4>{}); break;
        default: kloops(std::integral_constant<int, 5>{}); break;
    }
    if (EIG_TIMING) tq_k1 = __builtin_readcyclecounter();
    // WALK: every wave is past the last barrier of the K loop, every DMA of this N-block has landed -- part A of the next N-block's prologue goes out now and lands
    // behind the finishing phase below (U slot 0, plane slots 0, 1; the exchange lives in X)
    const bool more = it + 1 < nwalk;
    if (more) prologue_a(nblk + 1);

    // ---- output transform.  Along nu in-lane: c_xi,b (b = 0..3) of every N-tile.
    f32x4 cc[4][NIW];
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) {
        f32x4 Y[4];
        w4_out1d(acc[0][ni], acc[1][ni], acc[2][ni], acc[3][ni], acc[4][ni], acc[5][ni], Y);
#pragma unroll
        for (int b = 0; b < 4; ++b) cc[b][ni] = Y[b];
    }
    // Along xi: the six waves of a region publish their c rows in two rounds (N-tiles 0, 1 then 2, 3: [12 waves][2 N-tiles][4 e][64 lanes] 16-byte vectors over b = 96 KB
    // each; U / planes are dead -- every wave is past the last barrier, every DMA has landed); the finishing lanes read what their output rows need:
    // y0 = (c0 + s) + u, y1 = fma(2, w, d), y2 = fma(4, u, s), y3 = fma(8, w, d) + c5 with s = c1 + c2, d = c1 - c2, u = c3 + c4, w = c3 - c4.
    {   // (own scope: per-lane quantities from a FRESH lane_id(), scalars from coordinates offset by a fresh opaque zero: see the top of the kernel / of the walk)
    int zf = 0;
    asm volatile("" : "+s"(zf));
    const int eb = eb_i + zf, y0 = y0_i + zf, x0 = x0_i + zf, wv = wv_o + zf, rg = rg_o + zf, xi = xi_o + zf, HWf = HW + zf, Cout = a.Cout + zf, ngf = ng_o + zf;
    const bool pk_edge_f = (pk_edge_o ? 1 : 0) + zf != 0;
    const int lane = lane_id();
    const int q = lane >> 4, col = lane & 15;
    const size_t cHW = (size_t)HWf;
    // finishing lane: chunk j of the block row (= tile column tx = j), channel 8 chh + chl (wide: 8 chunks per row, 8 channels per unit; tall: 4 chunks, all 16 channels)
    const int j = TALL ? lane & 3 : lane & 7, chl = TALL ? lane >> 2 : lane >> 3, e_r = j & 3, ql = TALL ? 0 : j >> 2;
    int woff[4];                                                           // publishing lane (q, col): its slot in plane e
#pragma unroll
    for (int e = 0; e < 4; ++e) woff[e] = (((2 * xi + (NSPLIT ? ngf : rg)) * 2 * 4 + e) * 64 + q * 16 + ((col + 4 * e + (PACK ? q : (TALL ? 0 : 8 * (q & 1)))) & 15)) * 4;
    // PACK: finishing lane L takes tile r = L & 15 of the block (writer lane group r >> 2, register r & 3) for channel 4 cq + (L >> 4); a unit = one channel quarter cq;
    // the writer's slot q 16 + ((col + 4 e + q) & 15) spreads the sixteen tiles of a reader group over all sixteen slots.  The tile's image / row / column as in the K loop.
    const int fr = lane & 15, fchl = lane >> 4;
    const int f_i = pk_edge_f ? fr >> 2 : 0, f_ty = pk_edge_f ? fr & 3 : fr >> 2, f_tx = pk_edge_f ? 4 : fr & 3;
    if constexpr (EPI == EPI_LSTM || EPI == EPI_CONVP) {
        // All twelve waves finish outputs, and they finish them in IMAGE order: the exchange is laid out so that a finishing lane reads the four pixels b = 0..3 of ONE tile
        // as a 16-byte vector, and finishing lane L takes the 16-byte chunk j = L & 7 of a 32-pixel block row (tile tx = j, writer lane q = 2 ty + (j >> 2), register e =
        // j & 3) of channel 8 chh + (L >> 3): the eight lanes j = 0..7 of a channel read / write ONE 128-byte line of c, the peepholes, h or P (lane = (tile group, channel)
        // as the accumulators lie touched 64 lines per instruction, and the 7-9 us that cost per block were half of the time between two K loops:
        // profiles/r05_f_w4_timeline.txt).  Units of 64 chunks: wave xi < 4 takes output row a = xi of (tile row ty, channel half chh) = (0,0), (0,1), (1,0); waves 4 / 5
        // take (1,1) of rows 0, 1 / 2, 3 -- 12 or 8 cells per lane.  Rounds: N-tiles (gates) 0, 1 then 2, 3: [12 waves][2][4 e][64 lanes] 16-byte vectors = 96 KB each; the
        // slot of writer lane (q, col) in plane e is q 16 + ((col + 4 e + 8 (q & 1)) & 15), which spreads every 16-lane service group of the b128 reads over all 16 slots.
        // The sixteen units of a region: (output row arow, quarter cq) -- cq = (tile row ty, channel half chh) = (cq >> 1, cq & 1) in the wide shape, the tile row in the
        // tall one, the channel quarter with packed tiles.  Six waves per region: wave xi < 4 takes row xi of quarters 0, 1, 2; waves 4 / 5 quarter 3 of rows 0, 1 / 2, 3.
        // N-split (twelve waves, one region): (0, xi < 4) row xi of quarters 0, 1; (1, xi < 4) of quarter 2; (ng, 4) / (ng, 5) quarter 3 of row ng / 2 + ng.
        const int nun = NSPLIT ? ((xi < 4 && ngf == 0) ? 2 : 1) : (xi < 4 ? 3 : 2);
        auto unit_cq = [&](int un) __attribute__((always_inline)) { return xi < 4 ? (NSPLIT ? (ngf ? 2 : un) : un) : 3; };
        auto unit_arow = [&](int un) __attribute__((always_inline)) { return xi < 4 ? xi : 2 * (xi - 4) + (NSPLIT ? ngf : un); };
        int xoff[3];
#pragma unroll
        for (int un = 0; un < 3; ++un) {   // (tall: MFMA row r = 4 ty + tx -- the writer lane group q IS the tile row, a unit = one of the four tile rows for all 16 channels)
            const int cq = unit_cq(un);
            const int ty = TALL ? cq : cq >> 1, chh = TALL ? 0 : cq & 1;
            xoff[un] = TALL ? (e_r * 64 + ty * 16 + ((chl + 4 * e_r) & 15)) * 4 : (e_r * 64 + (2 * ty + ql) * 16 + ((8 * chh + chl + 4 * e_r + 8 * ql) & 15)) * 4;
            if constexpr (PACK) xoff[un] = ((fr & 3) * 64 + (fr >> 2) * 16 + ((4 * cq + fchl + 4 * (fr & 3) + (fr >> 2)) & 15)) * 4;
        }
        auto finish = [&](int arow, int nr, int off, int xw) __attribute__((always_inline)) -> f32x4 {   // output row arow of the reader's chunk: pixels b = 0..3 (xw: the region / N-split: the N-tile pair)
            auto C = [&](int x) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(xb + ((x * 2 + xw) * 2 + nr) * 1024 + off); };
            const f32x4 c1 = C(1), c2 = C(2), c3 = C(3), c4 = C(4);
            const f32x4 s_ = c1 + c2, d_ = c1 - c2, u_ = c3 + c4, w_ = c3 - c4;
            f32x4 y;
            if (arow == 0) { const f32x4 c0 = C(0); y = (c0 + s_) + u_; }
            else if (arow == 1) { for (int b = 0; b < 4; ++b) y[b] = fmaf(2.0f, w_[b], d_[b]); }
            else if (arow == 2) { for (int b = 0; b < 4; ++b) y[b] = fmaf(4.0f, u_[b], s_[b]); }
            else { const f32x4 c5 = C(5); for (int b = 0; b < 4; ++b) y[b] = fmaf(8.0f, w_[b], d_[b]) + c5[b]; }
            return y;
        };
        f32x4 ys[NI][3];   // [N-tile][unit]: the four pixels of the lane's chunk
        auto finish_unit = [&](int ni, int nr, int xw, int un) __attribute__((always_inline)) {   // (the output row is a compile-time constant of the wave's role)
            const int sel = NSPLIT ? ngf : un;   // waves 4, 5: which of their two rows
            switch (xi) {
                case 0: ys[ni][un] = finish(0, nr, xoff[un], xw); break;
                case 1: ys[ni][un] = finish(1, nr, xoff[un], xw); break;
                case 2: ys[ni][un] = finish(2, nr, xoff[un], xw); break;
                case 3: ys[ni][un] = finish(3, nr, xoff[un], xw); break;
                case 4: ys[ni][un] = sel ? finish(1, nr, xoff[un], xw) : finish(0, nr, xoff[un], xw); break;
                default: ys[ni][un] = sel ? finish(3, nr, xoff[un], xw) : finish(2, nr, xoff[un], xw); break;
            }
        };
        if constexpr (NSPLIT) {   // ONE round: wave (ng, xi) publishes its two N-tiles, every finishing lane reads all four
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x4 t;
                    t[0] = cc[0][nr][e]; t[1] = cc[1][nr][e]; t[2] = cc[2][nr][e]; t[3] = cc[3][nr][e];
                    *reinterpret_cast<f32x4*>(xb + woff[e] + nr * 1024) = t;
                }
            EIG4_LDS_BARRIER();
            if (EIG_TIMING) tq_x = __builtin_readcyclecounter();
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int un = 0; un < 2; ++un) {
                    if (un >= nun) break;
                    finish_unit(ni, ni & 1, ni >> 1, un);
                }
        } else {
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (2 * rnd >= NI) break;
            if (rnd) EIG4_LDS_BARRIER();   // (everyone has read round 0; lgkmcnt only -- the prefetches of the next N-block stay in flight)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr) {
                const int ni = 2 * rnd + nr;
                if (ni >= NI) break;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x4 t;
                    t[0] = cc[0][ni][e]; t[1] = cc[1][ni][e]; t[2] = cc[2][ni][e]; t[3] = cc[3][ni][e];
                    *reinterpret_cast<f32x4*>(xb + woff[e] + nr * 1024) = t;
                }
            }
            EIG4_LDS_BARRIER();
            if (EIG_TIMING && rnd == 0) tq_x = __builtin_readcyclecounter();
#pragma unroll
            for (int nr = 0; nr < 2; ++nr) {
                const int ni = 2 * rnd + nr;
                if (ni >= NI) break;
#pragma unroll
                for (int un = 0; un < 3; ++un) {
                    if (un >= nun) break;
                    finish_unit(ni, nr, rg, un);
                }
            }
        }
        }
        // WALK: this wave's share of part A has landed (issued ~3 us ago) -- waited for HERE, ahead of the gate loads and the stores, so that the barrier at the top
        // of the next N-block needs no vmcnt wait of its own (which would wait for the stores below as well)
        if (more) EIG4_WAITCNT(0x0F70);
        if (EIG_TIMING) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq_y = __builtin_readcyclecounter(); }
#pragma unroll
        for (int un = 0; un < 3; ++un) {
            if (un >= nun) break;
            const int cq = unit_cq(un);
            const int ty = TALL ? cq : cq >> 1, chh = TALL ? 0 : cq & 1;
            const int arow = unit_arow(un);
            const int gy = PACK ? 4 * f_ty + arow : y0 + RGH * rg + 4 * ty + arow, gx = PACK ? 4 * f_tx : x0 + 4 * j;
            if (gy >= a.H || gx >= a.W) continue;
            const int ebo = PACK ? eb + f_i : eb;   // (PACK: the tile's image)
            if (PACK && ebo >= a.B) continue;       // (past the last tile of the launch)
            const size_t pix = (size_t)gy * a.W + gx;
            const int chq = PACK ? 4 * cq + fchl : 8 * chh + chl;
            if constexpr (EPI == EPI_LSTM) {
                const int ch = nblk * 16 + chq;
                if (ch >= Cout) continue;
                const float bi = a.bias[ch], bf = a.bias[Cout + ch], bc = a.bias[2 * Cout + ch], bo = a.bias[3 * Cout + ch];
                const size_t cbase = ((size_t)ebo * Cout + ch) * cHW, ps = (size_t)Cout * cHW, pbase = (size_t)ch * cHW;
                const f32x4 cold4 = *reinterpret_cast<const f32x4*>(a.c_state + cbase + pix);
                const f32x4 pi4 = *reinterpret_cast<const f32x4*>(a.peep + pbase + pix);
                const f32x4 pf4 = *reinterpret_cast<const f32x4*>(a.peep + ps + pbase + pix);
                const f32x4 po4 = *reinterpret_cast<const f32x4*>(a.peep + 2 * ps + pbase + pix);
                f32x4 cn4, hn4;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float cn, hn;
                    lstm_cell(ys[0][un][b], ys[1][un][b], ys[2][un][b], ys[3][un][b], bi, bf, bc, bo, cold4[b], pi4[b], pf4[b], po4[b], cn, hn);
                    cn4[b] = cn; hn4[b] = hn;
                }
                *reinterpret_cast<f32x4*>(a.c_state + cbase + pix) = cn4;
                *reinterpret_cast<f32x4*>(a.h_out + cbase + pix) = hn4;
            } else {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int ch = (nblk * NI + ni) * 16 + chq;
                    if (ch >= Cout) continue;
                    const float bb_ = a.bias[ch];
                    f32x4 v4;
#pragma unroll
                    for (int b = 0; b < 4; ++b) v4[b] = relu_f(ys[ni][un][b] + bb_);
                    *reinterpret_cast<f32x4*>(a.Pout + ((size_t)ebo * Cout + ch) * cHW + pix) = v4;
                }
            }
        }
    } else {
        // ConvA, in image order as well: a finishing lane takes the chunk j of the output-row PAIR ap (rows 2 ap, 2 ap + 1) of tile row ty for one channel -- the two 2x2
        // pooling windows of its four pixels -- i.e. two pooled pixels (8 bytes); the eight lanes of a channel cover 16 pooled pixels = 64 contiguous bytes of P and of both
        // halves of E.  Units of 64 such lanes per round (N-tiles 2 rnd, 2 rnd + 1): g = (nr, ty, ap, chh), 16 of them (8 when the round holds one N-tile) over the six waves
        // of the region as 3,3,3,3,2,2 (2,2,1,1,1,1).  (Before: lane = (tile group, channel), 8-byte accesses to 64 different lines per instruction, and only xi < 4 worked.)
        const int Ho = a.H >> 1, Wo = a.W >> 1;
        const size_t plane = (size_t)Ho * Wo;
        const int ox = (x0 >> 1) + 2 * j;
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            if (2 * rnd >= NI) break;
            const int nval = NI - 2 * rnd >= 2 ? 2 : 1;
            if (rnd) EIG4_LDS_BARRIER();
#pragma unroll
            for (int nr = 0; nr < 2; ++nr) {
                if (nr >= nval) break;
                const int ni = 2 * rnd + nr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x4 t;
                    t[0] = cc[0][ni][e]; t[1] = cc[1][ni][e]; t[2] = cc[2][ni][e]; t[3] = cc[3][ni][e];
                    *reinterpret_cast<f32x4*>(xb + woff[e] + nr * 1024) = t;
                }
            }
            EIG4_LDS_BARRIER();
            if (rnd == (NI - 1) / 2) { if (more) EIG4_WAITCNT(0x0F70); }   // (see the ConvLSTM / ConvP branch)
            if (EIG_TIMING && rnd == 0) { tq_x = __builtin_readcyclecounter(); tq_y = tq_x; }
            const int g0 = nval == 2 ? (xi < 4 ? 3 * xi : 12 + 2 * (xi - 4)) : (xi < 2 ? 2 * xi : xi + 2);
            const int cnt = nval == 2 ? (xi < 4 ? 3 : 2) : (xi < 2 ? 2 : 1);
#pragma unroll
            for (int un = 0; un < 3; ++un) {
                if (un >= cnt) break;
                const int g = g0 + un;
                const int nr = g >> 3, ty = TALL ? (g >> 1) & 3 : (g >> 2) & 1, ap = TALL ? g & 1 : (g >> 1) & 1, chh = TALL ? 0 : g & 1;
                const int off = TALL ? ((nr + 2 * rg) * 256 + e_r * 64 + ty * 16 + ((chl + 4 * e_r) & 15)) * 4
                                     : ((nr + 2 * rg) * 256 + e_r * 64 + (2 * ty + ql) * 16 + ((8 * chh + chl + 4 * e_r + 8 * ql) & 15)) * 4;
                auto C = [&](int x) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(xb + x * 4096 + off); };
                const f32x4 c1 = C(1), c2 = C(2), c3 = C(3), c4 = C(4);
                const f32x4 s_ = c1 + c2, d_ = c1 - c2, u_ = c3 + c4, w_ = c3 - c4;
                f32x4 ya, yb;   // output rows 2 ap, 2 ap + 1
                if (ap == 0) {
                    const f32x4 c0 = C(0);
                    ya = (c0 + s_) + u_;
                    for (int b = 0; b < 4; ++b) yb[b] = fmaf(2.0f, w_[b], d_[b]);
                } else {
                    const f32x4 c5 = C(5);
                    for (int b = 0; b < 4; ++b) { ya[b] = fmaf(4.0f, u_[b], s_[b]); yb[b] = fmaf(8.0f, w_[b], d_[b]) + c5[b]; }
                }
                const int ch = (nblk * NI + 2 * rnd + nr) * 16 + 8 * chh + chl;
                const int oy = (y0 >> 1) + (RGH / 2) * rg + 2 * ty + ap;
                if (oy >= Ho || ox >= Wo || ch >= Cout) continue;
                const float bb_ = a.bias[ch];
                const size_t pb = ((size_t)eb * Cout + ch) * plane + (size_t)oy * Wo + ox;
                const size_t e0 = ((size_t)eb * 2 * Cout + ch) * plane + (size_t)oy * Wo + ox, e1 = e0 + (size_t)Cout * plane;
                const f32x2 p2 = *reinterpret_cast<const f32x2*>(a.P + pb);
                f32x2 ea, eb2;
#pragma unroll
                for (int bp = 0; bp < 2; ++bp) {
                    const float v00 = relu_f(ya[2 * bp] + bb_), v01 = relu_f(ya[2 * bp + 1] + bb_);
                    const float v10 = relu_f(yb[2 * bp] + bb_), v11 = relu_f(yb[2 * bp + 1] + bb_);
                    const float A = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
                    ea[bp] = relu_f(A - p2[bp]);
                    eb2[bp] = relu_f(p2[bp] - A);
                }
                *reinterpret_cast<f32x2*>(a.E + e0) = ea;
                *reinterpret_cast<f32x2*>(a.E + e1) = eb2;
            }
        }
    }
    }
    timeline();
    };   // walk_body
    for (int it = 0; it < nwalk; ++it) walk_body(it);
#undef EIG4_WAITCNT
#undef EIG4_BARRIER
#undef EIG4_LDS_BARRIER
#undef EIG4_IS_UP
}

}  // namespace eig
