// conv_winoh.h -- the Winograd F(2x2, 3x3) operators of conv_wino16.h WITHOUT a transformed-input buffer: every wave builds its A operands itself,
// straight from the DMA'd input planes.  Two block shapes of ONE kernel.  DESIGN.md section 3.1.
//
// Why (profiles/r05_a_w16_timeline.txt, r05_b_winoh_versions.txt): in conv_wino16.h eight of the sixteen waves transform the input (16 LDS reads, 32 additions,
// 16 LDS writes per channel and tile) for all sixteen to read back after a barrier -- the transform side costs 5.7 % and makes the transforming waves the slow ones at
// every barrier; and a block owns its CU alone (157 KB of LDS), so the matrix pipe idles 7.1-7.5 us between the K loops of successive blocks (gate epilogue 3.9 us,
// argument / descriptor set-up 1.2, prologue DMA round trip + first transform 1.7-2.0, dispatcher 0.25): 12.6 / 6.9 / 4.6 % of a ConvLSTM block at layers 1 / 2 / 3
// and 30-45 % of a ConvA / ConvP block.
//
// Same arithmetic in the same order as wino16_kernel / wino_kernel (oracle/eig_oracle.c: wino_accumulate / wino_finish): every accumulator is the fma chain over
// (source, channel) ascending, the transforms are the same additions on the same operands -- results are identical bit for bit.  What changes:
//   A operands in-wave: wave (rg, xi) multiplies region rg (16 tiles = pixel rows 4 rg .. 4 rg + 3 of the block) for the four positions (xi, nu = 0..3).  Lane (q, col)
//     holds channel q (+ 4 per k-step) of the K-block for tile col of the region; it reads the two patch rows that row xi of B^T d needs (xi 0: d0 - d2, 1: d1 + d2,
//     2: d2 - d1, 3: d1 - d3) out of the channel's plane -- three aligned 8-byte reads per row --, 4 additions for the row, 4 for the column pass: exactly the values the
//     transforming waves of conv_wino16.h wrote to V[4 xi + nu][channel][tile].  No V round trip (LDS write -> barrier -> read), no transforming role: every wave runs the
//     same K-block.
//   Block shapes <RG regions, KS k-steps per K-block>:
//     <2, 1>  HALF tiles, 8 x 16 output pixels, eight waves, K-blocks of four channels, 78 KB of LDS: TWO co-resident blocks per CU -- one block's epilogue, set-up and
//             prologue are covered by the other's MFMAs.  U ring 4 x 16 KB, plane ring 3; the fetches of a K-block are waited for one K-block LATER (vmcnt(n)).
//     <4, 2>  16 x 16 output pixels, sixteen waves, K-blocks of eight channels, one block per CU (the shape of conv_wino16.h, half the U traffic of <2, 1>).  U ring
//             3 x 32 KB, plane ring 2; a K-block's fetches are waited for at its end (its 32 MFMAs per wave last as long as a round trip).
//   Ring discipline: one barrier per K-block that never drains the matrix pipe -- the U slab and the planes of K-block j are visible to everyone DURING K-block j - 1:
//     plane(j)'s patch rows go into registers behind its first chunk of MFMAs, the A operands of j are built and its first B operand is read at its end, in front of
//     the barrier; the first instruction behind a barrier is an MFMA, and the staging work sits in slices between the chunks.  Past the end the fetch cursors stay on
//     the last K-block (a harmless duplicate), so every wave issues the same number of DMAs in every K-block.
//   The packed weights are wino16's ([K-blocks of 8][16 pos][8 ch][16 cols][NI]): the 4-channel half of a position is one contiguous KB = one LDS-DMA instruction.
//   An unpooled-source K-block (rows / columns s-1, s0, s0, s+1 of the half-resolution plane): nu = 2 is a chain of zeros for every xi, xi = 2 entirely -- not built,
//     not read, not multiplied (the oracle computes them: fma(0, u, M) = M, c + 0 = c).
//   Output transform / epilogues: conv_wino16.h's.
#pragma once
#include "conv_wino16.h"

namespace eig {

#ifndef EIG_WH_DIAG
#define EIG_WH_DIAG 0   // measurement builds only (WRONG RESULTS): 1 no wait for the K loop's DMAs, 2 no barrier in the K loop, 4 no A-operand build (patch reads + additions), 8 no plane DMA, 16 no U DMA
#endif
constexpr int WH_UPOS = 256;   // floats per (position, k-step) of a U slot: 4 ch x 16 cols x NI (NI = 3: 192 used)

template <int RG, int KS> struct WinohGeom {
    static constexpr int NW = 4 * RG;                 // waves per block: (region, xi)
    static constexpr int THREADS = 64 * NW;
    static constexpr int KC = 4 * KS;                 // channels per K-block
    static constexpr int U_FLOATS = 16 * KS * WH_UPOS;        // one U slot: [16 pos][KS][4 ch][16 cols][NI]
    static constexpr bool DEFER = KS == 1;            // the fetches of a K-block are waited for at the end of the NEXT one
    static constexpr int NUS = DEFER ? 4 : 3;         // U slots (K-block j lives in slot j % NUS)
    static constexpr int FD = NUS - 1;                // fetch distance in K-blocks
    static constexpr int NPS = FD;                    // plane slots (K-block j lives in slot j % NPS)
    static constexpr int PH = 4 * RG + 2;             // haloed rows y0-1 .. y0 + 4 RG of a channel's plane, 24 floats each: aligned chunks x0-4 .. x0+19
    static constexpr int PI = (PH * 6 + 63) / 64;     // LDS-DMA instructions per plane (64 chunks of 16 bytes each)
    static constexpr int PS = PI * 256 + 32;          // plane stride in floats, = 32 (mod 64): the 8-byte patch reads of the two channels of a 32-lane group fall on disjoint banks
    static constexpr int NPL = KC * PI;               // plane DMA instructions per K-block (wave w < NPL: channel w / PI, part w % PI)
    static constexpr int STAGE = NUS * U_FLOATS + NPS * KC * PS;
    static constexpr int LDS_FLOATS = STAGE > NW * 2048 ? STAGE : NW * 2048;   // (the output exchange: NW x 8 KB)
};
template <int RG, int KS> constexpr int winoh_lds_bytes() { return WinohGeom<RG, KS>::LDS_FLOATS * 4; }   // <2, 1>: 79360, <4, 2>: 133120

// rows a, b of the 4 x 4 patch with  t = d[a] -/+ d[b]  = row xi of B^T d (xi = 1 adds, the others subtract) ...
constexpr int wh_row_a(int xi) { return xi == 0 ? 0 : xi == 2 ? 2 : 1; }
constexpr int wh_row_b(int xi) { return xi == 2 ? 1 : xi == 3 ? 3 : 2; }
// ... and the same for an unpooled-source patch (rows s-1, s0, s0, s+1 = rows 0, 1, 1, 2 of the half-resolution plane; xi = 2 is d2 - d1 = 0)
constexpr int wh_urow_a(int xi) { return xi == 0 ? 0 : 1; }
constexpr int wh_urow_b(int xi) { return xi == 3 ? 2 : 1; }

template <int NI, int EPI, int RG, int KS>
__global__ void __launch_bounds__((WinohGeom<RG, KS>::THREADS), 4) winoh_kernel(const ConvArgs a)
{
    static_assert(EPI == EPI_LSTM || EPI == EPI_CONVA || EPI == EPI_CONVP, "conv_winoh.h: ConvLSTM, ConvA, ConvP");
    static_assert(EPI != EPI_LSTM || NI == 4, "ConvLSTM: the four N-tiles are the four gates");
    static_assert(NI == 3 || NI == 4, "N-blocks of 48 or 64 columns");
    static_assert((RG == 2 && KS == 1) || (RG == 4 && KS == 2), "block shapes: half tiles x 4 channels, full tiles x 8 channels");
    using G = WinohGeom<RG, KS>;
    constexpr int NW = G::NW, KC = G::KC, NUS = G::NUS, NPS = G::NPS, FD = G::FD, PS = G::PS, PI = G::PI;
    constexpr int U8 = wino_u_floats(NI);         // floats of one 8-channel K-block of the packed weights
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned long long tq_entry = EIG_TIMING ? __builtin_readcyclecounter() : 0;   // measurement builds (-DEIG_TIMING=1, scripts/timeline_w16.py)
    float* const Ub = lds;
    float* const Pb = lds + NUS * G::U_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv & (RG - 1), xi = wv / RG;
    const int q = lane >> 4, col = lane & 15;

    const int tiles = a.tilesX * a.tilesY;
    const int ntile = a.B * tiles;
    const int xcd = blockIdx.x & 7, xi_ = blockIdx.x >> 3;
    const int nblk = xi_ % a.n_nblk;
    const int tlin = a.tile_map ? xcd * ((ntile + 7) >> 3) + xi_ / a.n_nblk : (xi_ / a.n_nblk) * 8 + xcd;
    if (tlin >= ntile) return;
    const int eb = tlin / tiles;
    const int t_ = tlin - eb * tiles;
    const int tyi = t_ / a.tilesX, txi = t_ - tyi * a.tilesX;
    const int y0 = tyi * (4 * RG), x0 = txi * 16;
    const int HW = a.H * a.W;

    // K-blocks of KC channels; every source is a multiple of 8 channels
    const bool up_fused = EPI == EPI_LSTM && a.up_src != nullptr;
    const int nkb0 = a.src[0].C / KC;
    const int nkbu = up_fused ? (a.up_C / KC) : 0;
    const bool has1 = a.nsrc > 1;
    const int nkb = nkb0 + nkbu + (has1 ? (a.src[1].C / KC) : 0);
    const int up_lo = nkb0, up_hi = nkb0 + nkbu;
#define EIGH_WAITCNT(imm) do { __builtin_amdgcn_s_waitcnt(imm); asm volatile("" ::: "memory"); } while (0)
#define EIGH_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define EIGH_IS_UP(kb) ((kb) >= up_lo && (kb) < up_hi)
    const int nkb8 = nkb * KC / 8;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wpk + (size_t)nblk * nkb8 * U8), 0, nkb8 * U8 * 4, 0x00020000);

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // ---- plane fetch (waves w < NPL: channel w / PI of the K-block, part w % PI of its plane): lane = 16-byte chunk of the PH x 6-chunk haloed plane (unpooled source:
    // 2 RG + 2 rows x 4 chunks at half resolution, row stride 24, all in part 0); rows / chunks outside the image are out of the descriptor's range through a
    // saturating add = zeros
    const unsigned long long sb0 = (unsigned long long)(a.src[0].ptr + (size_t)eb * a.src[0].Ct * HW);
    const unsigned long long sb1 = has1 ? (unsigned long long)(a.src[1].ptr + (size_t)eb * a.src[1].Ct * HW) : sb0;
    const int sz0 = a.src[0].C * HW * 4, sz1 = has1 ? a.src[1].C * HW * 4 : sz0;
    const int Hh = a.H >> 1, Wh = a.W >> 1, HWh = Hh * Wh;
    const unsigned long long sbu = up_fused ? (unsigned long long)(a.up_src + (size_t)eb * a.up_C * HWh) : sb0;
    const int szu = up_fused ? a.up_C * HWh * 4 : sz0;
    const int pch = wv / PI, ppart = wv - pch * PI;   // this wave's plane DMA
    int roff, uoff;
    {
        const int c = lane + 64 * ppart;
        const int row = c / 6, cx = c - row * 6;
        const int gy = y0 - 1 + row, gx = x0 - 4 + 4 * cx;
        roff = (c < G::PH * 6 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (gy * a.W + gx) * 4 : -1;
        const int hy = (y0 >> 1) - 1 + row, hx = (x0 >> 1) - 4 + 4 * cx;
        uoff = (c < (2 * RG + 2) * 6 && cx < 4 && hy >= 0 && hy < Hh && hx >= 0 && hx < Wh) ? (hy * Wh + hx) * 4 : -1;
    }
    auto dma_plane_at = [&](int jj, int slot) __attribute__((always_inline)) {   // (prologue) this wave's plane DMA of K-block min(jj, nkb - 1) -> slot
        const int j = jj < nkb ? jj : nkb - 1;
        const bool up = EIGH_IS_UP(j);
        const bool s1 = j >= up_hi;
        // (arithmetic selection: a select between captured variables becomes a select between their ADDRESSES -- a table in scratch memory)
        const unsigned long long mu = 0ull - (unsigned long long)up, m1 = 0ull - (unsigned long long)s1;
        const unsigned long long u = sb0 + ((sb1 - sb0) & m1) + ((sbu - sb0) & mu);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        const int sz = sz0 + ((sz1 - sz0) & (int)m1) + ((szu - sz0) & (int)mu);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(sz), 0x00020000);
        const int base = (up_lo & (int)mu) + (up_hi & (int)m1), hw = HW + ((HWh - HW) & (int)mu);
        const unsigned coff = (unsigned)((j - base) * KC + pch) * (unsigned)(hw * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Pb + (slot * KC + pch) * PS + ppart * 256), 16,
                                                 (int)__builtin_elementwise_add_sat((unsigned)roff + (((unsigned)uoff - (unsigned)roff) & (unsigned)mu), coff), 0, 0, 0);
    };
    // ---- U fetch (every wave: two of the 16 KS (position, k-step) KBs of the slab): the 4-channel half of a position of the packed 8-channel K-block is contiguous
    const int uvo = (NI == 4 || lane < 48) ? lane * 16 : -1;
    auto dma_u = [&](int jj, int slot) __attribute__((always_inline)) {   // K-block min(jj, nkb - 1) -> slot
        const int j = jj < nkb ? jj : nkb - 1;
        constexpr unsigned HALF = 4 * 16 * NI * 4, POS = 8 * 16 * NI * 4;   // bytes
        const unsigned g8 = KS == 2 ? (unsigned)j : (unsigned)(j >> 1);
        const unsigned jh = KS == 2 ? 0u : (unsigned)(j & 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = 2 * wv + i, pos = f / KS, ks = f - pos * KS;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(Ub + slot * G::U_FLOATS + f * WH_UPOS), 16, uvo,
                                                     (int)(g8 * (U8 * 4) + (jh + (unsigned)ks) * HALF + (unsigned)pos * POS), 0, 0);
        }
    };

    // ---- A operands: lane (q, col) -> channel q (+ 4 per k-step) of the K-block, tile col of region rg (class-major map: r = 4 q' + reg <-> window (reg >> 1, 2 q' + (reg & 1)))
    const int t_ty = 2 * rg + ((col & 3) >> 1), t_tx = 2 * (col >> 2) + (col & 1);
    const float* const pbase_n = Pb + q * PS + (2 * t_ty) * 24 + 2 * t_tx + 2;   // patch row 0, ONE column left of the patch (8-byte aligned: three ds_read_b64 per row)
    const float* const pbase_u = Pb + q * PS + t_ty * 24 + t_tx + 3;             // row s-1, column s-1 of a half-resolution plane

    // accumulators: position (xi, nu), N-tile ni
    f32x4 acc[4][NI];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[p][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int b_off = xi * 4 * KS * WH_UPOS + (q * 16 + col) * NI;   // U[pos = 4 xi + nu][ks][ch = q][col][0 .. NI)

    // ---- prologue: the U slabs and planes of K-blocks 0 .. FD - 1
    const unsigned long long tq_setup = EIG_TIMING ? __builtin_readcyclecounter() : 0;
#pragma unroll
    for (int j = 0; j < FD; ++j) {
        dma_u(j, j);
        if (wv < G::NPL) dma_plane_at(j, j % NPS);
    }
    if (G::DEFER) { if (wv < G::NPL) EIGH_WAITCNT(0x0F73); else EIGH_WAITCNT(0x0F72); }   // all but the fetches of K-block FD - 1
    else EIGH_WAITCNT(0x0F70);
    EIGH_BARRIER();
    const unsigned long long tq_k0 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
    unsigned long long tq_k1 = 0, tq_x = 0, tq_y = 0;
    // [entry, set-up done, K loop start, K loop end, exchange barrier passed, y ready (gates start), exit, HW_ID | XCC_ID << 32]
    auto timeline = [&]() __attribute__((always_inline)) {
        if (EIG_TIMING && a.dbg && lane == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* dd = a.dbg + ((size_t)blockIdx.x * NW + wv) * 8;
            dd[0] = tq_entry; dd[1] = tq_setup; dd[2] = tq_k0; dd[3] = tq_k1; dd[4] = tq_x; dd[5] = tq_y; dd[6] = __builtin_readcyclecounter();
            dd[7] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
        }
    };

    const int ra = xi >> 1, seg = xi & 1;   // this wave finishes output row parity ra of segment seg
    // The K loops exist once per xi (role_tag) behind ONE wave-uniform branch: the patch rows a wave reads and the sign of its row combination are
    // compile-time constants of xi, and so are the positions an unpooled-source K-block skips.
    auto kloops = [&](auto role_tag) __attribute__((always_inline)) {
        constexpr int XI = decltype(role_tag)::value;
        constexpr bool PLW = G::NPL >= NW || XI < G::NPL / RG;   // this wave fetches planes
        float pr[KS][2][4];    // [k-step][row a / b][column]: the two patch rows of the NEXT K-block
        float v[KS][4];        // A operands of the current K-block (built at the end of the previous one)
        float bq[NI];          // B operand of the current K-block's first chunk, read before the barrier in front of it
        int rslot = 0;         // plane slot of the K-block whose patch rows are read next
        int uslot = 0;         // U slot of the current K-block
        int fu = FD % NUS;     // U slot the next fetch goes to
        // fetch cursor of the planes (K-block pj = kb + FD, clamped to the last one): the descriptor of its source as scalars (a mutable descriptor OBJECT ends in
        // scratch memory), the byte offset of this wave's channel, the slot
        int pj = 0, pslot = 0, psz = 0;
        unsigned pcoff = 0, plo = 0, phi = 0, phw4 = 0;
        bool pup = false;
        auto plane_source = [&](int j) __attribute__((always_inline)) {   // (re)position the cursor on K-block j: at the start and where a source begins
            const bool up = EIGH_IS_UP(j);
            const bool s1 = j >= up_hi;
            const unsigned long long mu = 0ull - (unsigned long long)up, m1 = 0ull - (unsigned long long)s1;
            const unsigned long long u = sb0 + ((sb1 - sb0) & m1) + ((sbu - sb0) & mu);
            plo = __builtin_amdgcn_readfirstlane((unsigned)u); phi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            psz = __builtin_amdgcn_readfirstlane(sz0 + ((sz1 - sz0) & (int)m1) + ((szu - sz0) & (int)mu));
            const int base = (up_lo & (int)mu) + (up_hi & (int)m1);
            phw4 = (unsigned)((HW + ((HWh - HW) & (int)mu)) * 4);
            pcoff = (unsigned)((j - base) * KC + pch) * phw4;
            pup = up; pj = j;
        };
        auto dma_plane = [&]() __attribute__((always_inline)) {   // this wave's plane DMA of K-block pj -> slot pslot, then advance the cursor
            const unsigned o = (unsigned)roff + (((unsigned)uoff - (unsigned)roff) & (0u - (unsigned)pup));
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)phi << 32) | plo), 0, psz, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (__attribute__((address_space(3))) void*)(Pb + (pslot * KC + pch) * PS + ppart * 256), 16,
                                                     (int)__builtin_elementwise_add_sat(o, pcoff), 0, 0, 0);
            pslot = pslot == NPS - 1 ? 0 : pslot + 1;
            if (pj + 1 < nkb) {
                if (pj + 1 == up_lo || pj + 1 == up_hi) plane_source(pj + 1);
                else { ++pj; pcoff += KC * phw4; }
            }
        };
        if (PLW) { plane_source(FD < nkb ? FD : nkb - 1); pslot = FD % NPS; }
        // patch rows of the K-block in slot rslot (up: an unpooled-source K-block), then advance rslot
        auto read_rows = [&](bool up) __attribute__((always_inline)) {
            const int so = rslot * (KC * PS);
            rslot = rslot == NPS - 1 ? 0 : rslot + 1;
            if (EIG_WH_DIAG & 4) return;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (up) {   // (xi = 2: row s0 twice -> d2 - d1 = an exact zero row, for the run-time-kind K-block below, which multiplies it; a compile-time one skips it)
                    const float* const p = pbase_u + so + ks * 4 * PS;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        if (r == 1 && wh_urow_b(XI) == wh_urow_a(XI)) { for (int c = 0; c < 4; ++c) pr[ks][1][c] = pr[ks][0][c]; continue; }
                        const float* const pl = p + (r ? wh_urow_b(XI) : wh_urow_a(XI)) * 24;
                        pr[ks][r][0] = pl[0]; pr[ks][r][1] = pl[1]; pr[ks][r][2] = pr[ks][r][1]; pr[ks][r][3] = pl[2];
                    }
                } else {
                    const float* const p = pbase_n + so + ks * 4 * PS;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {   // columns -1 .. 4 of the patch as three aligned 8-byte reads (conflict-free: plane stride = 32 mod 64); columns 0 .. 3 are used
                        const f32x2* const pl = reinterpret_cast<const f32x2*>(p + (r ? wh_row_b(XI) : wh_row_a(XI)) * 24);
                        const f32x2 x0_ = pl[0], x1_ = pl[1], x2_ = pl[2];
                        pr[ks][r][0] = x0_[1]; pr[ks][r][1] = x1_[0]; pr[ks][r][2] = x1_[1]; pr[ks][r][3] = x2_[0];
                    }
                }
            }
        };
        auto build_a = [&]() __attribute__((always_inline)) {   // row xi of B^T d, then the column pass, from the patch rows in pr
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (EIG_WH_DIAG & 4) { v[ks][0] = v[ks][1] = v[ks][2] = v[ks][3] = 1.0f; continue; }
                float t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) t[c] = (XI == 1) ? pr[ks][0][c] + pr[ks][1][c] : pr[ks][0][c] - pr[ks][1][c];
                v[ks][0] = t[0] - t[2]; v[ks][1] = t[1] + t[2]; v[ks][2] = t[2] - t[1]; v[ks][3] = t[1] - t[3];
            }
        };
        auto read_b = [&](int slot, int ks, int nu, float* dst) __attribute__((always_inline)) {
            const float* const bsrc = Ub + slot * G::U_FLOATS + b_off + (nu * KS + ks) * WH_UPOS;
            if constexpr (NI == 4) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bsrc);
                dst[0] = b4[0]; dst[1] = b4[1]; dst[2] = b4[2]; dst[3] = b4[3];
            } else {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) dst[ni] = bsrc[ni];
            }
        };
        // One K-block.  kind_tag: 0 full, 1 unpooled source, 2 run time; nk_tag: kind of K-block kb + 1 (whose patch rows are read now), same codes;
        // last_tag: the last K-block (nothing to stage).  On entry v and bq hold the A operands and the first B operand of this K-block: the first
        // instruction behind the barrier is an MFMA, and the staging work sits in slices BETWEEN the chunks of MFMAs -- patch rows of K-block kb + 1 behind
        // chunk 0, the U fetch behind chunk 1, the plane fetch behind chunk 2, the A operands of kb + 1 behind the last chunk.
        auto kiter = [&](const int kb, auto kind_tag, auto nk_tag, auto last_tag) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_tag)::value;
            constexpr int NK = decltype(nk_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr bool UP = KIND == 1;   // (run-time kind = the last K-block of all: an unpooled-source one there runs the full body on its exact-zero operands -- fma(0, u, M) = M)
            constexpr bool IDLE = UP && XI == 2;   // nothing to multiply
            constexpr int NPK = UP ? 3 : 4;        // positions per k-step: nu = 0, 1, (2,) 3
            constexpr int NCH = IDLE ? 0 : NPK * KS;
            const int nslot = uslot == NUS - 1 ? 0 : uslot + 1;
            float bv[2][NI];
            auto slice = [&](int i) __attribute__((always_inline)) {
                if constexpr (!LAST) {
                    if (i == 0) read_rows(NK == 2 ? EIGH_IS_UP(kb + 1) : NK == 1);
                    if (i == 1) { if (!(EIG_WH_DIAG & 16)) dma_u(kb + FD, fu); fu = fu == NUS - 1 ? 0 : fu + 1; }
                    if (i == 2) { if (PLW && !(EIG_WH_DIAG & 8)) dma_plane(); }
                }
            };
            if constexpr (IDLE) {
                slice(0); slice(1); slice(2);
                if constexpr (!LAST) { read_b(nslot, 0, 0, bq); build_a(); }
            } else {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int ks = i / NPK, pn = i - ks * NPK, nu = (UP && pn == 2) ? 3 : pn;
                    if (i + 1 < NCH) {
                        const int ks1 = (i + 1) / NPK, pn1 = (i + 1) - ks1 * NPK, nu1 = (UP && pn1 == 2) ? 3 : pn1;
                        read_b(uslot, ks1, nu1, bv[(i + 1) & 1]);
                    } else if constexpr (!LAST) read_b(nslot, 0, 0, bq);
                    const float* const b = i == 0 ? bq : bv[i & 1];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[nu][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[ks][nu], b[ni], acc[nu][ni], 0, 0, 0);
                    slice(i);
                    if (i == NCH - 1) { if constexpr (!LAST) build_a(); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            uslot = nslot;
            // DEFER: the fetches of the PREVIOUS K-block have landed (this K-block's n = 2 or 3 may stay in flight); otherwise, and after the last K-block: everything
            if (LAST || !G::DEFER || (EIG_WH_DIAG & (8 | 16))) { if (!(EIG_WH_DIAG & 1) || LAST) EIGH_WAITCNT(0x0F70); }
            else if (!(EIG_WH_DIAG & 1)) { if (PLW) EIGH_WAITCNT(0x0F73); else EIGH_WAITCNT(0x0F72); }
            if (!(EIG_WH_DIAG & 2) || LAST) EIGH_BARRIER();
        };
        const std::false_type nl{};
        const std::integral_constant<int, 2> rt{};
        int kb = 0;
        read_rows(EIGH_IS_UP(0));
        read_b(0, 0, 0, bq);
        build_a();
        EIGH_WAITCNT(0xC07F);
        EIGH_BARRIER();   // (every wave has read plane 0 out of its slot before anyone's K-block 0 fetches into it)
        // K-blocks [kb, end) of one kind; the LAST K-block of all is left out.  A K-block reads the patch rows of the next one: same kind except at the end
        // of a run (run time there).
        auto run = [&](const int end, auto kind_tag) __attribute__((always_inline)) {
            for (; kb + 1 < end; ++kb) kiter(kb, kind_tag, kind_tag, nl);
            if (kb + 1 == end && end < nkb) { kiter(kb, kind_tag, rt, nl); ++kb; }
        };
        run(up_lo, std::integral_constant<int, 0>{});
        run(up_hi, std::integral_constant<int, 1>{});
        run(nkb, std::integral_constant<int, 0>{});
        kiter(nkb - 1, rt, rt, std::true_type{});
    };
    if (wv < RG) kloops(std::integral_constant<int, 0>{});
    else if (wv < 2 * RG) kloops(std::integral_constant<int, 1>{});
    else if (wv < 3 * RG) kloops(std::integral_constant<int, 2>{});
    else kloops(std::integral_constant<int, 3>{});
    if (EIG_TIMING) tq_k1 = __builtin_readcyclecounter();

    // ---- output transform (conv_wino16.h).  Columns in-lane: c_xi,0 = (M_xi0 + M_xi1) + M_xi2, c_xi,1 = (M_xi1 - M_xi2) - M_xi3.
    f32x4 cc[2][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        cc[0][ni] = (acc[0][ni] + acc[1][ni]) + acc[2][ni];
        cc[1][ni] = (acc[1][ni] - acc[2][ni]) - acc[3][ni];
    }
    // Rows: y_0b = (c_0b + c_1b) + c_2b, y_1b = c_1b - (c_2b + c_3b).  Every wave publishes its c row; wave (rg, xi = 2 ra + seg) then finishes row
    // parity ra of segment seg = accumulator registers 2 seg, 2 seg + 1.
    float* const xb = lds;   // [NW waves = (xi, rg)][8 (b, ni)][64 lanes][4 registers]: U / planes are dead (every wave is past the last barrier, every DMA has landed)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *reinterpret_cast<f32x4*>(xb + ((wv * 8 + b * 4 + ni) * 64 + lane) * 4) = cc[b][ni];
    __syncthreads();
    if (EIG_TIMING) tq_x = __builtin_readcyclecounter();
#define EIGH_ROW(x) (xb + (RG * (x) + rg) * 2048)   // the c row published by wave (xi = x, rg)
    if constexpr (EPI == EPI_LSTM) {
        float y[2][NI][2];   // [b = px][ni][window 2 seg + k]
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int e = ((b * 4 + ni) * 64 + lane) * 4 + 2 * seg;   // registers 2 seg, 2 seg + 1 of the publishing wave's c row
                const f32x2 c1 = *reinterpret_cast<const f32x2*>(EIGH_ROW(1) + e), c2 = *reinterpret_cast<const f32x2*>(EIGH_ROW(2) + e);
                const f32x2 c03 = *reinterpret_cast<const f32x2*>(EIGH_ROW(ra ? 3 : 0) + e);
#pragma unroll
                for (int k = 0; k < 2; ++k) y[b][ni][k] = ra ? c1[k] - (c2[k] + c03[k]) : (c03[k] + c1[k]) + c2[k];
            }
        if (EIG_TIMING) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq_y = __builtin_readcyclecounter(); }
        const int ch = nblk * 16 + col;
        if (ch >= a.Cout) { timeline(); return; }
        const float bi = a.bias[ch], bf = a.bias[a.Cout + ch], bc = a.bias[2 * a.Cout + ch], bo = a.bias[3 * a.Cout + ch];
        const size_t cbase = ((size_t)eb * a.Cout + ch) * (size_t)HW;
        {
            const int gy = y0 + 4 * rg + ra + 2 * seg, gx = x0 + 4 * q;
            if (gy >= a.H || gx >= a.W) { timeline(); return; }
            const int pix = gy * a.W + gx;
            // the cell state and the three peephole values of the lane's 4-pixel segment
            const size_t pb = (size_t)ch * HW + pix, ps = (size_t)a.Cout * HW;
            f32x4 st4[4];
            st4[0] = *reinterpret_cast<const f32x4*>(a.c_state + cbase + pix);
            st4[1] = *reinterpret_cast<const f32x4*>(a.peep + pb);
            st4[2] = *reinterpret_cast<const f32x4*>(a.peep + ps + pb);
            st4[3] = *reinterpret_cast<const f32x4*>(a.peep + 2 * ps + pb);
            f32x4 cn4, hn4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // element j = sub-tile px = j & 1 of window 2 seg + (j >> 1)
                float cn, hn;
                lstm_cell(y[j & 1][0][j >> 1], y[j & 1][1][j >> 1], y[j & 1][2][j >> 1], y[j & 1][3][j >> 1],
                          bi, bf, bc, bo, st4[0][j], st4[1][j], st4[2][j], st4[3][j], cn, hn);
                cn4[j] = cn; hn4[j] = hn;
            }
            *reinterpret_cast<f32x4*>(a.c_state + cbase + pix) = cn4;
            *reinterpret_cast<f32x4*>(a.h_out + cbase + pix) = hn4;
        }
        timeline();
    } else {
        // N-TILE split: wave (rg, xi) finishes N-tile xi of its region for both row parities (xi = 3 rests where NI = 3): all four parity classes of
        // a window in one lane -- ConvA's max_pooling_2d is the max over a lane's four values, ConvP stores whole 4 x 4-pixel patches (conv_wino.h)
        const int k = xi;
        if (k >= NI) return;
        f32x4 y[2][2];   // [py][px]
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int e = ((b * 4 + k) * 64 + lane) * 4;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(EIGH_ROW(0) + e), c1 = *reinterpret_cast<const f32x4*>(EIGH_ROW(1) + e);
            const f32x4 c2 = *reinterpret_cast<const f32x4*>(EIGH_ROW(2) + e), c3 = *reinterpret_cast<const f32x4*>(EIGH_ROW(3) + e);
            y[0][b] = (c0 + c1) + c2;
            y[1][b] = c1 - (c2 + c3);
        }
        const int ch = nblk * NI * 16 + k * 16 + col;
        if (ch >= a.Cout) return;
        const float bb = a.bias[ch];
        const size_t cHW = (size_t)HW;
        if constexpr (EPI == EPI_CONVP) {
            const size_t base = ((size_t)eb * a.Cout + ch) * cHW;
#pragma unroll
            for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int gy = y0 + 4 * rg + 2 * wy + py, gx = x0 + 4 * q;
                    if (gy >= a.H || gx >= a.W) continue;
                    f32x4 v4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = relu_f(y[py][j & 1][2 * wy + (j >> 1)] + bb);
                    *reinterpret_cast<f32x4*>(a.Pout + base + gy * a.W + gx) = v4;
                }
        } else {
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
                    const float v00 = relu_f(y[0][0][r] + bb), v01 = relu_f(y[0][1][r] + bb), v10 = relu_f(y[1][0][r] + bb), v11 = relu_f(y[1][1][r] + bb);
                    const float A = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
                    ea[j] = relu_f(A - p2[j]);
                    eb2[j] = relu_f(p2[j] - A);
                }
                *reinterpret_cast<f32x2*>(a.E + e0 + oy * Wo + ox) = ea;
                *reinterpret_cast<f32x2*>(a.E + e1 + oy * Wo + ox) = eb2;
            }
        }
    }
#undef EIGH_ROW
#undef EIGH_WAITCNT
#undef EIGH_BARRIER
#undef EIGH_IS_UP
}

}  // namespace eig
