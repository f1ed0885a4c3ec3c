// conv_wino.h -- the 3x3 convolutions of PredNet layers >= 1 (ConvLSTM over E_l / unpooled R_{l+1} / h_l, ConvA, ConvP) as Winograd
// F(2x2, 3x3) on the fp32 matrix pipe.  DESIGN.md section 3.1d.
//
// Why: the direct implicit-GEMM kernel (conv_mfma.h) runs at 0.91 of the fp32 MFMA peak; at fp32 only FEWER multiply-adds make the
// roll-out faster.  F(2x2, 3x3) needs 16 multiply-adds per channel and 2x2 output pixels instead of 36 (2.25x fewer).  The parity
// half of the question was answered first (profiles/r04_c_winograd_study.json): the reference's element-wise order with Winograd
// convolutions is indistinguishable from any other fp32 re-order of it.  Default for every eligible operator since it measured faster at
// every shape (EIGEN_WINOGRAD = bit mask, eigen_engine.hip: wino_op); the canonical arithmetic of an operator that takes it is stated
// in oracle/eig_oracle.c (wino_*), operation by operation, and this kernel executes exactly those operations:
//   input transform   t = B^T d (rows), V = t B (columns)        -- fp32 additions / subtractions, fixed order
//   16 chains         M_pos[o][T] = fmaf(V_pos[c][T], U_pos[o][c], .) over (source, channel) ascending -- v_mfma_f32_16x16x4_f32
//   output transform  c = M A (columns), y = A^T c (rows)        -- fixed order
// then bias and the operator's epilogue: lstm_cell of conv_mfma.h / relu + 2x2 max-pool + error units / relu.  A ConvLSTM's unpooled
// source either rides in the same chains (up_fused below: 9 of the 16 positions are non-zero) or, where that does not apply, its
// 2x2-form chain (an EPI_UP4 launch of conv_mfma.h) is added with one fp32 addition.
//
// Block = 16 x 16 output pixels of one image (64 tiles of 2x2 = the 64 pooling windows of the class-major map) x NI 16-column N-tiles;
// 8 waves, one block per CU.  K-block = 8 channels of one source.  Per K-block and block: 128 NI MFMAs (direct kernel, NI = 4: 1152).
//   wave w, lane l TRANSFORMS channel w of the K-block for tile l: the 4x4 patch out of the wave's private LDS plane (MODE 8; filled by
//     LDS-DMA, rows / chunks outside the image are out of the buffer descriptor's range = zeros), 32 additions, 16 ds_write_b32 into
//     V[pos][channel][tile];
//   wave w COMPUTES region rg = w & 3 (tiles 16 rg .. 16 rg + 15 = rows 4 rg .. 4 rg + 3 of the tile) for positions (xi, nu) with
//     xi in {2 h, 2 h + 1}, h = w >> 2: 8 positions x NI accumulator tiles; per k-step 8 ds_read_b32 (A) + 8 ds_read_b128 (B: a lane's
//     NI columns of one channel are contiguous) for 8 NI MFMAs.
//   U (8 NI KB per K-block) travels global -> LDS by LDS-DMA, double-buffered; V is double-buffered too: the transform of K-block k+1
//     is written while K-block k is being multiplied -- ONE barrier per K-block.
//   Output transform: columns in-lane; the row transform needs xi = 0..3, so the two waves of a region exchange c-rows through LDS
//     (free after the K loop) -- see the two splits below.
#pragma once
#include "conv_mfma.h"

namespace eig {

constexpr int WINO_THREADS = 512;
constexpr int WINO_VS = 80;                          // floats per (position, channel) row of V: 64 tiles + 16 (k-slots q, q+1 on disjoint banks)
constexpr int WINO_V_FLOATS = 16 * KC * WINO_VS;     // 10240
constexpr int wino_u_floats(int NI) { return 16 * KC * 16 * NI; }   // [16 pos][8 ch][16 cols][NI]: 8192 for NI = 4
constexpr int WINO_RAW_FLOATS = 18 * 24;               // MODE 8: one channel's haloed rows y0-1 .. y0+16, aligned chunks x0-4 .. x0+19, per WAVE
constexpr int wino_lds_bytes(int NI, bool raw = false) { return (2 * (WINO_V_FLOATS + wino_u_floats(NI)) + (raw ? 8 * WINO_RAW_FLOATS : 0)) * 4 + (NI == 3 ? 16 : 0); }  // 147456 / 161280 for NI = 4
static_assert(KC == 8, "conv_wino.h: 8-channel K-blocks");

// NI 16-column N-tiles per block.  EPI_LSTM (NI = 4): N-tile = gate, column = channel 16 nblk + col; the wave pair of a region splits the
// OUTPUT ROWS (each wave needs all four gates of its pixels).  EPI_CONVA / EPI_CONVP (NI = 3, 4): column = channel 16 (NI nblk + ni) +
// col; the pair splits the N-TILES instead, so that each wave ends up with all four parity classes of its channels -- the 2x2 tile
// IS ConvA's pooling window (max in-lane), and ConvP stores whole 4 x 4-pixel patches.
// MODE (same operations on the same data in every mode -- results are identical; A/B'd in profiles/r04_g_wino_modes.txt):
//   0  every wave transforms K-block k + 1, then multiplies K-block k
//   4  software pipeline written out: 8 chunks of 8 MFMAs, each with the operand reads of the next chunk and a slice of the staging
//      work, separated by scheduling fences (a wave fills its own matrix-pipe shadows): +3.6 % on the headline
//   8  (default; also prefetches the ConvLSTM's cell state / peepholes during the last K-block: +0.4 %) mode 4 with the patch rows staged through wave-private LDS planes by LDS-DMA (see dma_raw): +3.3 % again, and every
//      input element crosses the memory system 1.7 times instead of 4 (profiles/r04_m_wino_raw_staging.txt)
//   5, 6, 7  MEASUREMENT ONLY (wrong results): mode 4 without the U DMA after the prologue / without the patch loads / without patch loads
//      and most V writes -- what the staging traffic costs (profiles/r04_k_wino_bounds.txt)
//   (tried and dropped: the waves of one half / of one parity multiplying first and transforming afterwards, -3 ... -6 %; the same
//    interleave dictated with sched_group_barrier, -1.5 %)
template <int NI, int EPI, int MODE>
__global__ void __launch_bounds__(WINO_THREADS, 1) wino_kernel(const ConvArgs a)
{
    static_assert(EPI == EPI_LSTM || EPI == EPI_CONVA || EPI == EPI_CONVP, "conv_wino.h: ConvLSTM, ConvA, ConvP");
    static_assert(EPI != EPI_LSTM || NI == 4, "ConvLSTM: the four N-tiles are the four gates");
    static_assert(NI == 3 || NI == 4, "N-blocks of 48 or 64 columns");
    constexpr int WINO_U_FLOATS = wino_u_floats(NI);
    const unsigned long long tw_entry = EIG_TIMING ? __builtin_readcyclecounter() : 0;   // measurement builds (-DEIG_TIMING=1, scripts/timeline_wino.py)
    unsigned long long tw_wait = 0, tw_work = 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Vb = lds;                              // [2][16][8][WINO_VS]
    float* const Ub = lds + 2 * WINO_V_FLOATS;          // [2][16][8][16][4]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv & 3, half = wv >> 2;
    const int q = lane >> 4, col = lane & 15;

    // ---- block -> (N-block, image, tile): the XCD-aware order of conv_mfma.h (speed only)
    const int tiles = a.tilesX * a.tilesY;
    const int ntile = a.B * tiles;
    const int xcd = blockIdx.x & 7, xi_ = blockIdx.x >> 3;
    const int nblk = xi_ % a.n_nblk;
    const int tlin = a.tile_map ? xcd * ((ntile + 7) >> 3) + xi_ / a.n_nblk : (xi_ / a.n_nblk) * 8 + xcd;
    if (tlin >= ntile) return;
    const int eb = tlin / tiles;
    const int t_ = tlin - eb * tiles;
    const int tyi = t_ / a.tilesX, txi = t_ - tyi * a.tilesX;
    const int y0 = tyi * 16, x0 = txi * 16;
    const int HW = a.H * a.W;

    // ---- the transform side of this thread: channel wv of every K-block, tile `lane` of the block.  Tile l = 16 rg' + r,
    // r = 4 q' + reg  <->  window (wy, wx) = (reg >> 1, 2 q' + (reg & 1)) of region rg' (the class-major map), so that MFMA row r of a
    // region is the window the direct kernel's epilogue expects in accumulator register `reg` of lane group q'.
    const int t_rg = lane >> 4, t_r = lane & 15;
    const int t_ty = 2 * t_rg + ((t_r & 3) >> 1), t_tx = 2 * (t_r >> 2) + (t_r & 1);
    const int py0 = y0 + 2 * t_ty - 1, px0 = x0 + 2 * t_tx;   // patch rows py0 .. py0 + 3, columns px0 - 1 .. px0 + 2
    // byte offsets inside ONE channel plane; -1 = outside the image (a saturating add keeps it out of the descriptor's range)
    int off_c[4], off_l[4], off_r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = py0 + i;
        const bool rok = y >= 0 && y < a.H && px0 < a.W;
        off_c[i] = rok ? (y * a.W + px0) * 4 : -1;
        off_l[i] = (rok && px0 >= 1) ? (y * a.W + px0 - 1) * 4 : -1;
        off_r[i] = (rok && px0 + 2 < a.W) ? (y * a.W + px0 + 2) * 4 : -1;
    }
    // K-blocks: 8 channels of one source, sources in list order (every source here has a multiple of 8 channels)
    // ConvLSTM with its unpooled source R_{l+1} INSIDE the chains (a.up_src, MODE 8): K-blocks nkb0 .. nkb0 + nkbu - 1, between E_l and h_l,
    // are 8 channels of the HALF-resolution map -- oracle/eig_oracle.c: eig_wino_fuse_up
    const bool up_fused = EPI == EPI_LSTM && MODE == 8 && a.up_src != nullptr;
    const int nkb0 = a.src[0].C >> 3;
    const int nkbu = up_fused ? (a.up_C >> 3) : 0;
    const int nkb = nkb0 + nkbu + (a.nsrc > 1 ? (a.src[1].C >> 3) : 0);
    const bool has1 = a.nsrc > 1;
    const int up_lo = nkb0, up_hi = nkb0 + nkbu;   // K-blocks [up_lo, up_hi) read the unpooled source (empty range: none)
// s_waitcnt through the builtin (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]; 0x0F70 = vmcnt(0), 0xC07F = lgkmcnt(0),
// 0x0070 = both, 0x007C = vmcnt(12) lgkmcnt(0)): the compiler's own wait-count pass sees it and stops assuming the reads in front of it
// are still in flight (behind an `asm` wait it added lgkmcnt(0) in front of the first MFMAs of a K-block, i.e. a second and third LDS
// round trip); the empty asm keeps the memory fence of the old spelling
// measurement builds only (wrong results): -DEIG_WINO_DIAG=mask leaves parts of a K-block out -- 1 the U DMA, 2 the plane DMA, 4 the transform's LDS
// writes (a sink keeps the values alive), 8 the barrier, 16 the additions of the transform, 32 the wait for the DMAs in front of the barrier, 64 the
// patch reads (opaque register definitions instead)
#ifndef EIG_WINO_DIAG
#define EIG_WINO_DIAG 0
#endif
#ifndef EIG_WINO_RAWBAR
#define EIG_WINO_RAWBAR 1
#endif
#define EIG_WAITCNT(imm) do { __builtin_amdgcn_s_waitcnt(imm); asm volatile("" ::: "memory"); } while (0)
#define EIG_IS_UP(kb) ((kb) >= up_lo && (kb) < up_hi)
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wpk + (size_t)nblk * nkb * WINO_U_FLOATS), 0, nkb * WINO_U_FLOATS * 4, 0x00020000);

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float d[4][4];
    // descriptor of the source a K-block reads, from SCALAR selects of base pointer and size (no branch, no descriptor table); a K-block
    // past the last one gets a channel offset of 2^31: every lane's saturating add lands out of range and the loads return zeros
    const unsigned long long sb0 = (unsigned long long)(a.src[0].ptr + (size_t)eb * a.src[0].Ct * HW);
    const unsigned long long sb1 = has1 ? (unsigned long long)(a.src[1].ptr + (size_t)eb * a.src[1].Ct * HW) : sb0;
    const int sz0 = a.src[0].C * HW * 4, sz1 = has1 ? a.src[1].C * HW * 4 : sz0;
    auto load_patch = [&](int kb) __attribute__((always_inline)) {
        const bool s1 = kb >= nkb0;
        const unsigned long long u = s1 ? sb1 : sb0;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(s1 ? sz1 : sz0), 0x00020000);
        const unsigned in_range = (unsigned)((kb - nkb) >> 31);   // all ones while kb < nkb (bit arithmetic: a ternary here became a branch
        const unsigned coff = ((unsigned)(((kb - (s1 ? nkb0 : 0)) * KC + wv) * HW * 4) & in_range) | (0x80000000u & ~in_range);  // that split the K-block's basic block)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned vl = __builtin_elementwise_add_sat((unsigned)off_l[i], coff), vc = __builtin_elementwise_add_sat((unsigned)off_c[i], coff),
                           vr = __builtin_elementwise_add_sat((unsigned)off_r[i], coff);
            d[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vl, 0, 0));
            const f32x2 m = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vc, 0, 0));
            d[i][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vr, 0, 0));
            d[i][1] = m[0]; d[i][2] = m[1];
        }
    };
    // MODE 8: the patch rows come through LDS.  Every wave owns a private plane [18][24] behind V and U and fills it by LDS-DMA -- 108 chunks
    // of 16 B: chunk `lane` and, for lane < 44, chunk 64 + lane -- with the rows of ITS channel of the K-block; its lanes then read their
    // 4x4 patches from it (12 LDS reads instead of 12 global loads, and every input element crosses the memory system 1.7 times
    // instead of 4).  Private planes need no barrier: the wave reads the plane (K-block k + 1), then refills it (k + 2).
    float* const rawp = lds + 2 * (WINO_V_FLOATS + WINO_U_FLOATS) + wv * WINO_RAW_FLOATS;
    int roff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int c = lane + 64 * r, row = c / 6, cx = c - row * 6;
        const int gy = y0 - 1 + row, gx = x0 - 4 + 4 * cx;
        roff[r] = (c < 108 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (gy * a.W + gx) * 4 : -1;
    }
    // an unpooled-source K-block: plane [10][24] of the half-resolution map (the same row stride as the full-resolution plane), rows
    // Y0 - 1 .. Y0 + 8, aligned chunks X0 - 4 .. X0 + 11 in the first four chunks of a row (60 lanes, 40 of them fetch)
    const int Hh = a.H >> 1, Wh = a.W >> 1, HWh = Hh * Wh;
    int uoff;
    {
        const int row = lane / 6, cx = lane - row * 6;
        const int gy = (y0 >> 1) - 1 + row, gx = (x0 >> 1) - 4 + 4 * cx;
        uoff = (lane < 60 && cx < 4 && gy >= 0 && gy < Hh && gx >= 0 && gx < Wh) ? (gy * Wh + gx) * 4 : -1;
    }
    const unsigned long long sbu = up_fused ? (unsigned long long)(a.up_src + (size_t)eb * a.up_C * HWh) : sb0;
    const int szu = up_fused ? a.up_C * HWh * 4 : sz0;
    // ONE instruction stream for both kinds of K-block (no branch, no merge copies): descriptor, channel offset and the lanes' chunk
    // offsets are selects on the wave-uniform kind; an unpooled-source K-block's second instruction fetches nothing (zeros behind its rows)
    auto dma_raw = [&](int kb, float* plane) __attribute__((always_inline)) {
        const bool up = EIG_IS_UP(kb);
        const bool s1 = kb >= nkb0 + nkbu;
        // (additive selects: a three-way ?: of the base pointers sent the kernel arguments through scratch memory)
        const unsigned long long mu = 0ull - (unsigned long long)up, m1 = 0ull - (unsigned long long)(s1 && !up);
        const unsigned long long u = sb0 + ((sb1 - sb0) & m1) + ((sbu - sb0) & mu);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        const int sz = sz0 + ((sz1 - sz0) & (int)m1) + ((szu - sz0) & (int)mu);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(sz), 0x00020000);
        const unsigned in_range = (unsigned)((kb - nkb) >> 31);
        const unsigned chan = (unsigned)((kb - (up ? nkb0 : (s1 ? nkb0 + nkbu : 0))) * KC + wv) * (unsigned)((up ? HWh : HW) * 4);
        const unsigned coff = (chan & in_range) | (0x80000000u & ~in_range);
        const unsigned o0 = up ? (unsigned)uoff : (unsigned)roff[0], o1 = up ? 0xFFFFFFFFu : (unsigned)roff[1];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)plane, 16, (int)__builtin_elementwise_add_sat(o0, coff), 0, 0, 0);
        if (lane < 44)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(plane + 64 * 4), 16, (int)__builtin_elementwise_add_sat(o1, coff), 0, 0, 0);
    };
    // the lane's 4x4 patch out of the plane.  Full resolution: plane row 0 = image row y0 - 1, plane column 0 = image column x0 - 4, the patch
    // of tile (ty, tx) at rows 2 ty + i, columns 2 tx + 3 + j.  Unpooled source: the patch of the x2 nearest-unpooled map around the tile = source
    // pixel (Y0 + ty, X0 + tx) has rows / columns s_-1, s_0, s_0, s_+1, i.e. plane rows ty + (0, 1, 1, 2), columns tx + 3 + (0, 1, 1, 2): the
    // same sixteen reads with the base of rows 2, 3 moved up one row and the base of columns 2, 3 one column to the left.
    const int rd_off = (2 * t_ty) * 24 + 2 * t_tx + 3;
    const int rd_off_u = t_ty * 24 + t_tx + 3;
    auto read_patch = [&](int kb, const float* plane) __attribute__((always_inline)) {
        if constexpr (EIG_WINO_DIAG & 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "=v"(d[i][j]));
            return;
        }
        const bool up = EIG_IS_UP(kb);
        const float* const p00 = plane + (up ? rd_off_u : rd_off);
        const float* const p10 = p00 - (up ? 24 : 0);
        const int cs = up ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* const pl = (i < 2 ? p00 : p10) + i * 24;
            const float* const pr = pl - cs;
            d[i][0] = pl[0]; d[i][1] = pl[1]; d[i][2] = pr[2]; d[i][3] = pr[3];
        }
    };
    // B^T d B of the patch in d -> V[buf][pos][wv][lane]  (oracle/eig_oracle.c: wino_accumulate, same operations in the same order)
    auto transform = [&](float* vbuf) __attribute__((always_inline)) {
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
        }
        float* dst = vbuf + wv * WINO_VS + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[(i * 4 + 0) * KC * WINO_VS] = t[i][0] - t[i][2];
            dst[(i * 4 + 1) * KC * WINO_VS] = t[i][1] + t[i][2];
            dst[(i * 4 + 2) * KC * WINO_VS] = t[i][2] - t[i][1];
            dst[(i * 4 + 3) * KC * WINO_VS] = t[i][1] - t[i][3];
        }
    };
    // the U slab of K-block kb
    auto dma_u = [&](int kb, float* ubuf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NI; ++j)   // 512 NI chunks of 16 B, lane-linear
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(ubuf + (j * WINO_THREADS + wv * 64) * 4), 16,
                                                     tid * 16, (int)((unsigned)(j * WINO_THREADS * 16) + (unsigned)kb * (WINO_U_FLOATS * 4)), 0, 0);   // (scalar offset: no VALU)
    };

    // accumulators: position p = 4 xl + nu (xl = 0, 1: xi = 2 half + xl), N-tile ni
    f32x4 acc[8][NI];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[p][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int a_off = (half * 8 * KC + q) * WINO_VS + rg * 16 + col;   // V[pos = 8 half + p][ch = 4 ks + q][tile 16 rg + col]
    const int b_off = ((half * 8 * KC + q) * 16 + col) * NI;           // U[pos][ch][col][0..NI-1]
    auto read_b = [&](const float* ucur, int p, int ks, float (&bv)[NI]) __attribute__((always_inline)) {
        const float* src = ucur + b_off + (p * KC + ks * 4) * 16 * NI;
        if constexpr (NI == 4) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(src);
            bv[0] = b4[0]; bv[1] = b4[1]; bv[2] = b4[2]; bv[3] = b4[3];
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bv[ni] = src[ni];
        }
    };

    // ---- prologue: K-block 0 transformed and staged, the patch of K-block 1 in flight
    if constexpr (MODE == 8) {
        dma_raw(0, rawp);
        dma_u(0, Ub);
        EIG_WAITCNT(0x0F70);
        read_patch(0, rawp);
        EIG_WAITCNT(0xC07F);
        dma_raw(1, rawp);
        transform(Vb);
    } else {
        load_patch(0);
        dma_u(0, Ub);
        transform(Vb);
        load_patch(1);
    }
    EIG_WAITCNT(0x0070);
    if constexpr (MODE == 8) read_patch(1, rawp);   // (MODE 8: a K-block's patch is read out of the plane BEFORE the barrier in front of it, see kiter)
    __syncthreads();

    // the chain of the unpooled source (EPI_UP4 launch at half this resolution): loaded during the LAST K-block (ConvLSTM only)
    const bool has_up = EPI == EPI_LSTM && a.acc_init != nullptr;
    f32x4 upc[2][4];
    const int Hs = a.H >> 1, Ws = a.W >> 1, up_hw = Hs * Ws, up_cstride = a.n_nblk * 64 * up_hw;
    const __amdgpu_buffer_rsrc_t rs_up = __builtin_amdgcn_make_buffer_rsrc((void*)(has_up ? a.acc_init + (size_t)eb * 4 * up_cstride : a.zeros), 0, has_up ? 4 * up_cstride * 4 : 0, 0x00020000);
    int up_off[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int gy0 = y0 + 4 * rg + 2 * v, gx0 = x0 + 4 * q;
        up_off[v] = (gy0 < a.H && gx0 < a.W) ? ((nblk * 64 + col) * up_hw + (gy0 >> 1) * Ws + (gx0 >> 1)) * 4 : 0;
    }
    auto up_loads = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int soff = (ni * 16 * up_hw + (2 * half + mi) * up_cstride) * 4;  // plane = parity class (py = half, px = mi)
                    const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_up, up_off[v], soff, 0));
                    upc[mi][ni][2 * v] = t[0]; upc[mi][ni][2 * v + 1] = t[1];
                }
    };

    // ConvLSTM: the cell state and the three peephole values of the lane's two 4-pixel segments -- fetched during the LAST K-block where no
    // separate unpooled-source chain occupies those registers (otherwise in the epilogue, as the direct kernel does): their latency is
    // then off the block's critical path (one block per CU: nothing else would cover it)
    f32x4 st4[2][4];
    const bool pre_state = EPI == EPI_LSTM && !has_up;
    auto state_loads = [&]() __attribute__((always_inline)) {
        const int ch = nblk * 16 + col;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int gy = y0 + 4 * rg + half + 2 * sl, gx = x0 + 4 * q;
            if (ch >= a.Cout || gy >= a.H || gx >= a.W) continue;
            const size_t pix = (size_t)gy * a.W + gx, cb = ((size_t)eb * a.Cout + ch) * HW + pix, pb = (size_t)ch * HW + pix, ps = (size_t)a.Cout * HW;
            st4[sl][0] = *reinterpret_cast<const f32x4*>(a.c_state + cb);
            st4[sl][1] = *reinterpret_cast<const f32x4*>(a.peep + pb);
            st4[sl][2] = *reinterpret_cast<const f32x4*>(a.peep + ps + pb);
            st4[sl][3] = *reinterpret_cast<const f32x4*>(a.peep + 2 * ps + pb);
        }
    };
    // An unpooled-source K-block skips its zero chains: positions (p = 4 xl + nu) with nu = 2, and xl = 0 of the upper half (xi = 2) -- one
    // bit per position in a scalar mask (spelled out as predicates they cost every K-block ~60 scalar instructions; two compile-time
    // bodies made the register allocator spill 500 VGPRs)
    const unsigned wave_skip = half ? 0x4Fu : 0x44u;
    auto kiter = [&](const int kb, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        const unsigned skip = EIG_IS_UP(kb) ? wave_skip : 0u;
        const unsigned long long tk0 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
        const float* const vcur = Vb + (kb & 1) * WINO_V_FLOATS;
        const float* const ucur = Ub + (kb & 1) * WINO_U_FLOATS;
        // the U slab of K-block kb + 1 first: every vector-memory instruction issued after it (the 12 patch loads) may still be in
        // flight at the barrier, the DMA may not
        if constexpr (!LAST) { if constexpr (MODE != 5 && !(EIG_WINO_DIAG & 1)) dma_u(kb + 1, Ub + ((kb + 1) & 1) * WINO_U_FLOATS); }   // (MODE 5: measurement only)
        else if (has_up) up_loads();
        else if constexpr (EPI == EPI_LSTM) state_loads();
        if constexpr (MODE >= 4 && !LAST) {
            // software pipeline written out: 8 chunks of 2 NI MFMAs (k-step ks = c >> 2, positions 2 pp, 2 pp + 1 with pp = c & 3); each
            // chunk carries the operand reads of the NEXT chunk and a slice of the staging work, fenced so that the slices stay in
            // the matrix-pipe shadow of their chunk: c = 0, 1 the column pass of B^T d, c = 2..5 one row of V each (4 subtractions +
            // 4 LDS writes), c = 6, 7 the twelve loads of the patch after next.
            float* const vnext = Vb + ((kb + 1) & 1) * WINO_V_FLOATS + wv * WINO_VS + lane;
            float t[4][4];
            float av[2][2];
            float bv[2][2][NI];
            auto fetch = [&](int c, int slot) __attribute__((always_inline)) {
                const int ks = c >> 2, pp = c & 3;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    av[slot][u] = vcur[a_off + ((2 * pp + u) * KC + ks * 4) * WINO_VS];
                    read_b(ucur, 2 * pp + u, ks, bv[slot][u]);
                }
            };
            // (the source-select scalars of load_patch, once)
            const int kb2 = kb + 2;
            const bool s1 = kb2 >= nkb0;
            const unsigned long long ub = s1 ? sb1 : sb0;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ub), hi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(s1 ? sz1 : sz0), 0x00020000);
            const unsigned in_range = (unsigned)((kb2 - nkb) >> 31);
            const unsigned coff = ((unsigned)(((kb2 - (s1 ? nkb0 : 0)) * KC + wv) * HW * 4) & in_range) | (0x80000000u & ~in_range);
            fetch(0, 0);
            if constexpr (MODE == 8) {
                // the patch of K-block kb + 1 was read out of this wave's plane before the barrier (end of the previous K-block: the plane
                // is private and its DMA had landed), so that ONE LDS round trip -- shared with the first operands -- stands between the
                // barrier and the first MFMA; then the plane is refilled for kb + 2
                EIG_WAITCNT(0xC07F);
                if constexpr (!(EIG_WINO_DIAG & 2)) dma_raw(kb + 2, rawp);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int pp = c & 3;
                if (c + 1 < 8) fetch(c + 1, (c + 1) & 1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    // an unpooled-source K-block: the positions with xi = 2 (this wave's xl = 0 when half = 1) or nu = 2 are chains of exact zeros
                    if ((skip >> (2 * pp + u)) & 1) continue;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[2 * pp + u][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][u], bv[c & 1][u][ni], acc[2 * pp + u][ni], 0, 0, 0);
                }
                if (c < 2) {
#pragma unroll
                    for (int j = 2 * c; j < 2 * c + 2; ++j) {
                        if constexpr (EIG_WINO_DIAG & 16) { t[0][j] = d[0][j]; t[1][j] = d[1][j]; t[2][j] = d[2][j]; t[3][j] = d[3][j]; continue; }
                        t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
                    }
                } else if (c < 6) {
                    const int i = c - 2;
                    float v4[4] = {t[i][0] - t[i][2], t[i][1] + t[i][2], t[i][2] - t[i][1], t[i][1] - t[i][3]};
                    if constexpr (EIG_WINO_DIAG & 16) { v4[0] = t[i][0]; v4[1] = t[i][1]; v4[2] = t[i][2]; v4[3] = t[i][3]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (EIG_WINO_DIAG & 4) asm volatile("" :: "v"(v4[j]));   // (a sink: the value stays alive, no LDS write)
                        else if (MODE != 7 || j > 0) vnext[(i * 4 + j) * KC * WINO_VS] = v4[j];
                    }
                } else if constexpr (MODE != 6 && MODE != 7 && MODE != 8) {
#pragma unroll
                    for (int i = 2 * (c - 6); i < 2 * (c - 6) + 2; ++i) {
                        const unsigned vl = __builtin_elementwise_add_sat((unsigned)off_l[i], coff), vc = __builtin_elementwise_add_sat((unsigned)off_c[i], coff),
                                       vr = __builtin_elementwise_add_sat((unsigned)off_r[i], coff);
                        d[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vl, 0, 0));
                        const f32x2 m = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vc, 0, 0));
                        d[i][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vr, 0, 0));
                        d[i][1] = m[0]; d[i][2] = m[1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            if constexpr (!LAST) {   // K-block kb + 1: transform its patch (in d), then fetch the patch of kb + 2
                transform(Vb + ((kb + 1) & 1) * WINO_V_FLOATS);
                load_patch(kb + 2);  // (past the end: out of range, zeros, never used)
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float av[8];
                float bv[8][NI];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    av[p] = vcur[a_off + (p * KC + ks * 4) * WINO_VS];
                    read_b(ucur, p, ks, bv[p]);
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    if ((skip >> p) & 1) continue;   // (zero chains of an unpooled-source K-block)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[p][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p], bv[p][ni], acc[p][ni], 0, 0, 0);
                }
            }
        }
        const unsigned long long tk1 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
        if constexpr (!LAST && MODE != 6 && MODE != 7 && MODE != 8) EIG_WAITCNT(0x007C);   // (12 = the loads of load_patch)
        else if constexpr (EIG_WINO_DIAG & 32) EIG_WAITCNT(0xC07F);
        else EIG_WAITCNT(0x0070);
        if constexpr (MODE == 8 && !LAST) read_patch(kb + 2, rawp);   // (past the end: the plane holds zeros, never used)
#if EIG_WINO_DIAG & 8
#elif EIG_WINO_RAWBAR
        // the bare barrier: every LDS write and DMA of this wave has landed (the wait above); __syncthreads() would add a release fence =
        // lgkmcnt(0), i.e. wait for the patch reads just issued, which are private to the wave and may stay in flight across the barrier
        asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
#else
        __syncthreads();
#endif
        if (EIG_TIMING) { const unsigned long long tk2 = __builtin_readcyclecounter(); tw_work += tk1 - tk0; tw_wait += tk2 - tk1; }
    };
    const unsigned long long tw_loop0 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
    for (int kb = 0; kb + 1 < nkb; ++kb) kiter(kb, std::false_type{});
    kiter(nkb - 1, std::true_type{});
    const unsigned long long tw_loop1 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
    // [entry, K loop start, K loop end, exit, HW_ID | XCC_ID << 32, cycles waiting (waitcnt + barrier) in the K loop, cycles working in it, K-blocks]
    auto timeline = [&]() {
        if (EIG_TIMING && a.dbg && lane == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* dd = a.dbg + ((size_t)blockIdx.x * 8 + wv) * 8;
            dd[0] = tw_entry; dd[1] = tw_loop0; dd[2] = tw_loop1; dd[3] = __builtin_readcyclecounter();
            dd[4] = (unsigned long long)hwid | ((unsigned long long)xcc << 32); dd[5] = tw_wait; dd[6] = tw_work; dd[7] = (unsigned long long)nkb;
        }
    };

    // ---- output transform.  Columns in-lane: c_xi0 = (M_xi0 + M_xi1) + M_xi2, c_xi1 = (M_xi1 - M_xi2) - M_xi3.
    f32x4 cc[2][2][NI];  // [xl][b][ni]
#pragma unroll
    for (int xl = 0; xl < 2; ++xl)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            cc[xl][0][ni] = (acc[xl * 4 + 0][ni] + acc[xl * 4 + 1][ni]) + acc[xl * 4 + 2][ni];
            cc[xl][1][ni] = (acc[xl * 4 + 1][ni] - acc[xl * 4 + 2][ni]) - acc[xl * 4 + 3][ni];
        }
    // Rows: y_0b = (c_0b + c_1b) + c_2b,  y_1b = c_1b - (c_2b + c_3b).  Wave (rg, 0) holds c_0, c_1, wave (rg, 1) c_2, c_3 of the region.
    float* const xb = lds;  // [8 waves][32][64 lanes]; V / U are dead (every wave is past the last barrier)
    const int ch0 = (EPI == EPI_LSTM) ? nblk * 16 + col : nblk * NI * 16 + col;   // channel of N-tile 0 (LSTM: of every gate)
    const size_t cHW = (size_t)HW;
    if constexpr (EPI == EPI_LSTM) {
        // ROW split: wave (rg, 0) finishes row parity 0 and needs c_2 of its partner, wave (rg, 1) row parity 1 and needs c_1.
        {
            const int mine = half ? 0 : 1;  // half 0 sends c_1 (its xl = 1), half 1 sends c_2 (its xl = 0)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xb[(wv * 32 + (b * 4 + ni) * 4 + r) * 64 + lane] = cc[mine][b][ni][r];
        }
        __syncthreads();
        f32x4 y[2][4];  // [mi = px][ni]: the accumulators of the eight-wave direct kernel (classes (py = half, px))
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = xb[((wv ^ 4) * 32 + (b * 4 + ni) * 4 + r) * 64 + lane];
                if (half == 0) y[b][ni] = (cc[0][b][ni] + cc[1][b][ni]) + o;
                else y[b][ni] = o - (cc[0][b][ni] + cc[1][b][ni]);
            }
        if (has_up) {  // + the chain of the unpooled source: one fp32 addition, as in the direct kernel
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) y[mi][ni] = y[mi][ni] + upc[mi][ni];
        }
        // gate epilogue: the eight-wave (W8), 16-wide, 16-byte-access path of conv_mfma.h's EPI_LSTM.  Segment sl = row
        // 4 rg + half + 2 sl of the tile, columns 4 q .. 4 q + 3; element j = register 2 sl + (j >> 1) of sub-tile j & 1.
        const int ch = ch0;
        if (ch >= a.Cout) { timeline(); return; }
        const float bi = a.bias[ch], bf = a.bias[a.Cout + ch], bc = a.bias[2 * a.Cout + ch], bo = a.bias[3 * a.Cout + ch];
        const size_t cbase = ((size_t)eb * a.Cout + ch) * cHW;
        const size_t pbase = (size_t)ch * cHW;
        const size_t pstride = (size_t)a.Cout * cHW;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int gy = y0 + 4 * rg + half + 2 * sl, gx = x0 + 4 * q;
            if (gy >= a.H || gx >= a.W) continue;
            const int pix = gy * a.W + gx;
            f32x4 cold4, pi4, pf4, po4;
            if (pre_state) { cold4 = st4[sl][0]; pi4 = st4[sl][1]; pf4 = st4[sl][2]; po4 = st4[sl][3]; }
            else {
                cold4 = *reinterpret_cast<const f32x4*>(a.c_state + cbase + pix);
                pi4 = *reinterpret_cast<const f32x4*>(a.peep + pbase + pix);
                pf4 = *reinterpret_cast<const f32x4*>(a.peep + pstride + pbase + pix);
                po4 = *reinterpret_cast<const f32x4*>(a.peep + 2 * pstride + pbase + pix);
            }
            f32x4 cn4, hn4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float cn, hn;
                lstm_cell(y[j & 1][0][2 * sl + (j >> 1)], y[j & 1][1][2 * sl + (j >> 1)], y[j & 1][2][2 * sl + (j >> 1)], y[j & 1][3][2 * sl + (j >> 1)],
                          bi, bf, bc, bo, cold4[j], pi4[j], pf4[j], po4[j], cn, hn);
                cn4[j] = cn; hn4[j] = hn;
            }
            *reinterpret_cast<f32x4*>(a.c_state + cbase + pix) = cn4;
            *reinterpret_cast<f32x4*>(a.h_out + cbase + pix) = hn4;
        }
    } else {
        // N-TILE split: wave (rg, 0) finishes N-tiles 0, 1, wave (rg, 1) N-tiles 2 (, 3) -- for BOTH row parities, so each wave needs the
        // partner's two c rows of its own N-tiles: 2 (xl) x 2 (b) x 2 (ni) x 4 floats per lane each way.
        constexpr int N0 = 2;                      // N-tiles of half 0; half 1 owns NI - 2
        const int my0 = half ? N0 : 0;             // first own N-tile
        const int nmine = half ? NI - N0 : N0;     // own N-tiles (1 or 2)
        {
            const int their0 = half ? 0 : N0;
#pragma unroll
            for (int xl = 0; xl < 2; ++xl)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        // (compile-time register indices: select between the two candidate N-tiles)
                        const f32x4 v = half ? cc[xl][b][k] : cc[xl][b][(N0 + k < NI) ? N0 + k : NI - 1];
                        (void)their0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) xb[(wv * 32 + ((xl * 2 + b) * 2 + k) * 4 + r) * 64 + lane] = v[r];
                    }
        }
        __syncthreads();
        // y[py][px][k] of own N-tile k (register r = window (wy, wx) = (r >> 1, 2 q + (r & 1)))
        f32x4 y[2][2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x4 o0, o1;  // the partner's c rows (its xl = 0, 1) of my N-tile k
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o0[r] = xb[((wv ^ 4) * 32 + ((0 * 2 + b) * 2 + k) * 4 + r) * 64 + lane];
                    o1[r] = xb[((wv ^ 4) * 32 + ((1 * 2 + b) * 2 + k) * 4 + r) * 64 + lane];
                }
                const f32x4 m0 = half ? cc[0][b][(N0 + k < NI) ? N0 + k : NI - 1] : cc[0][b][k];   // my own rows of N-tile k
                const f32x4 m1 = half ? cc[1][b][(N0 + k < NI) ? N0 + k : NI - 1] : cc[1][b][k];
                // half 0 holds c_0, c_1 (partner: c_2, c_3); half 1 holds c_2, c_3 (partner: c_0, c_1)
                const f32x4 c0 = half ? o0 : m0, c1 = half ? o1 : m1, c2 = half ? m0 : o0, c3 = half ? m1 : o1;
                y[0][b][k] = (c0 + c1) + c2;
                y[1][b][k] = c1 - (c2 + c3);
            }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= nmine) continue;
            const int ch = ch0 + (my0 + k) * 16;
            if (ch >= a.Cout) continue;
            const float bb = a.bias[ch];
            if constexpr (EPI == EPI_CONVP) {
                // P = relu(y + b): the lane's 4 x 4 pixels (rows 4 rg + 2 wy + py, columns 4 q + 2 (r & 1) + px), one 16-byte store per row
                const size_t base = ((size_t)eb * a.Cout + ch) * cHW;
#pragma unroll
                for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        const int gy = y0 + 4 * rg + 2 * wy + py, gx = x0 + 4 * q;
                        if (gy >= a.H || gx >= a.W) continue;
                        f32x4 v4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v4[j] = relu_f(y[py][j & 1][k][2 * wy + (j >> 1)] + bb);
                        *reinterpret_cast<f32x4*>(a.Pout + base + gy * a.W + gx) = v4;
                    }
            } else {
                // A = max-pool of relu(y + b) over the tile = over the four classes; E = [relu(A - P), relu(P - A)] at half resolution:
                // pooled rows (y0 >> 1) + 2 rg + wy, pooled columns (x0 >> 1) + 2 q, + 1 (registers 2 wy, 2 wy + 1): 8-byte accesses
                const int Ho = a.H >> 1, Wo = a.W >> 1;
                const size_t plane = (size_t)Ho * Wo;
                const size_t pb = ((size_t)eb * a.Cout + ch) * plane;
                const size_t e0 = ((size_t)eb * 2 * a.Cout + ch) * plane, e1 = e0 + (size_t)a.Cout * plane;
#pragma unroll
                for (int wy = 0; wy < 2; ++wy) {
                    const int oy = (y0 >> 1) + 2 * rg + wy, ox = (x0 >> 1) + 2 * q;
                    if (oy >= Ho || ox >= Wo) continue;
                    const f32x2 p2 = *reinterpret_cast<const f32x2*>(a.P + pb + oy * Wo + ox);
                    f32x2 ea, eb2;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int r = 2 * wy + j;
                        const float v00 = relu_f(y[0][0][k][r] + bb), v01 = relu_f(y[0][1][k][r] + bb), v10 = relu_f(y[1][0][k][r] + bb), v11 = relu_f(y[1][1][k][r] + bb);
                        const float A = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
                        ea[j] = relu_f(A - p2[j]);
                        eb2[j] = relu_f(p2[j] - A);
                    }
                    *reinterpret_cast<f32x2*>(a.E + e0 + oy * Wo + ox) = ea;
                    *reinterpret_cast<f32x2*>(a.E + e1 + oy * Wo + ox) = eb2;
                }
            }
        }
    }
    timeline();
}

#undef EIG_IS_UP
#undef EIG_WAITCNT
}  // namespace eig
