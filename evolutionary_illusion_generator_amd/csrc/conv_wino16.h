// conv_wino16.h -- the Winograd F(2x2, 3x3) operators on SIXTEEN waves per block (four per SIMD).  DESIGN.md section 3.1d.
//
// Why: the model of the round-4 eight-wave K-block (scripts/wino_loop_model.hip, profiles/r04_w_wino_loop_model.txt) shows every piece of the staging work
// costing the matrix pipe several times its instruction count when only ONE partner wave per SIMD can fill in, and the barrier 9 points once
// the waves drift; the same K-block on sixteen waves of half the accumulators each loses 4 points to all of it instead of 17.
//
// Same block (16 x 16 output pixels of one image x 64 columns = 16 channels x 4 gates), same K-blocks, same LDS image (V, U double-buffered,
// eight planes), same arithmetic in the same order as the eight-wave kernel of round 4 -- the results are identical bit for bit.  What changes:
//   wave w COMPUTES region rg = w & 3 for the FOUR positions (xi, nu = 0..3) with xi = w >> 2: 4 x 4 accumulator tiles (64 VGPRs), 32 MFMAs per
//     K-block in 8 chunks of 4, each with the operand reads of the next chunk;
//   waves 0-7 TRANSFORM channel w of the K-block (plane DMA, patch reads before the barrier, column pass in chunk 0, one row in each of the chunks 1-4),
//     waves 8-15 fetch the U slab (4 LDS-DMA instructions each): on every SIMD two waves of each kind;
//   an unpooled-source K-block (xi = 2 and nu = 2 are chains of exact zeros): waves 8-11 have nothing to multiply, the others 3 positions;
//   output transform: columns in-lane (c_xi,b), every wave publishes its 32 values per lane in LDS, then wave (rg, 2 a + s) finishes output row
//     parity a of segment s (tile rows 4 rg + a + 2 s: window row s) from c_0..c_2 or c_1..c_3 -- the pixel ownership of the eight-wave
//     kernel's gate epilogue split in two, whose code (lstm_cell, 16-byte accesses) is reused.
#pragma once
#include "wino_launch.h"

namespace eig {
// (the eight-wave F(2x2) kernel of rounds 4-5, conv_wino.h, is gone: round 6 -- docs/HISTORY.md section 3.1d keeps its description)
constexpr int WINO_RAW_FLOATS = 18 * 24;                            // one channel's haloed rows y0-1 .. y0+16, aligned chunks x0-4 .. x0+19

#ifndef EIG_W16_TRIM_UP
#define EIG_W16_TRIM_UP 1      // (0: every K-block's patch through the full 4 x 4 transform -- A/B builds)
#endif
#ifndef EIG_W16_CONST_SKIP
#define EIG_W16_CONST_SKIP 1   // (0: the skipped positions of an unpooled-source K-block as a scalar bit mask tested at run time -- A/B builds)
#endif
// LDS image of this kernel: V [2][16 pos][8 ch][64 tiles] WITHOUT padding -- tile index XOR 16 on odd channels keeps the k-slots q, q + 1 of an
// operand read on disjoint banks -- (64 KB), U [2][16][8][16][4] (64 KB), and TWO planes per channel (K-blocks of even / odd index: 27 KB)
constexpr int W16_VS = 64;
constexpr int W16_V_FLOATS = 16 * KC * W16_VS;   // 8192
constexpr int wino16_lds_bytes(int NI) { return (2 * (W16_V_FLOATS + wino_u_floats(NI)) + 16 * WINO_RAW_FLOATS) * 4; }   // 158720 for NI = 4

template <int NI, int EPI>
__global__ void __launch_bounds__(WINO16_THREADS, 1) wino16_kernel(const ConvArgs a)
{
    static_assert(EPI == EPI_LSTM || EPI == EPI_CONVA || EPI == EPI_CONVP, "conv_wino16.h: ConvLSTM, ConvA, ConvP");
    static_assert(EPI != EPI_LSTM || NI == 4, "ConvLSTM: the four N-tiles are the four gates");
    static_assert(NI == 3 || NI == 4, "N-blocks of 48 or 64 columns");
    constexpr int WINO_U_FLOATS = wino_u_floats(NI);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned long long tq_entry = EIG_TIMING ? __builtin_readcyclecounter() : 0;   // measurement builds (-DEIG_TIMING=1, scripts/timeline_w16.py)
    float* const Vb = lds;
    float* const Ub = lds + 2 * W16_V_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv & 3, xi = wv >> 2;
    const int q = lane >> 4, col = lane & 15;
    const bool xf = wv < 8;            // transforming wave (channel wv); the others fetch U
    const int tch = wv & 7;

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

    // transform side: tile `lane` of the block
    const int t_rg = lane >> 4, t_r = lane & 15;
    const int t_ty = 2 * t_rg + ((t_r & 3) >> 1), t_tx = 2 * (t_r >> 2) + (t_r & 1);
    const bool up_fused = EPI == EPI_LSTM && a.up_src != nullptr;
    const int nkb0 = a.src[0].C >> 3;
    const int nkbu = up_fused ? (a.up_C >> 3) : 0;
    const int nkb = nkb0 + nkbu + (a.nsrc > 1 ? (a.src[1].C >> 3) : 0);
    const bool has1 = a.nsrc > 1;
    const int up_lo = nkb0, up_hi = nkb0 + nkbu;
#define EIG16_WAITCNT(imm) do { __builtin_amdgcn_s_waitcnt(imm); asm volatile("" ::: "memory"); } while (0)
#define EIG16_IS_UP(kb) ((kb) >= up_lo && (kb) < up_hi)
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wpk + (size_t)nblk * nkb * WINO_U_FLOATS), 0, nkb * WINO_U_FLOATS * 4, 0x00020000);

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float d[4][4];
    const unsigned long long sb0 = (unsigned long long)(a.src[0].ptr + (size_t)eb * a.src[0].Ct * HW);
    const unsigned long long sb1 = has1 ? (unsigned long long)(a.src[1].ptr + (size_t)eb * a.src[1].Ct * HW) : sb0;
    const int sz0 = a.src[0].C * HW * 4, sz1 = has1 ? a.src[1].C * HW * 4 : sz0;
    // plane (channel tch, parity of the K-block); fetched by wave 8 + tch, read by wave tch
    float* const planes = lds + 2 * (W16_V_FLOATS + WINO_U_FLOATS) + tch * WINO_RAW_FLOATS;
    constexpr int PLANE_PAR = 8 * WINO_RAW_FLOATS;   // planes of odd K-blocks behind those of even ones
    int roff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int c = lane + 64 * r, row = c / 6, cx = c - row * 6;
        const int gy = y0 - 1 + row, gx = x0 - 4 + 4 * cx;
        roff[r] = (c < 108 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (gy * a.W + gx) * 4 : -1;
    }
    const int Hh = a.H >> 1, Wh = a.W >> 1, HWh = Hh * Wh;
    int uoff;
    {
        const int row = lane / 6, cx = lane - row * 6;
        const int gy = (y0 >> 1) - 1 + row, gx = (x0 >> 1) - 4 + 4 * cx;
        uoff = (lane < 60 && cx < 4 && gy >= 0 && gy < Hh && gx >= 0 && gx < Wh) ? (gy * Wh + gx) * 4 : -1;
    }
    const unsigned long long sbu = up_fused ? (unsigned long long)(a.up_src + (size_t)eb * a.up_C * HWh) : sb0;
    const int szu = up_fused ? a.up_C * HWh * 4 : sz0;
    auto dma_raw = [&](int kb, int par) __attribute__((always_inline)) {   // (round 4: dma_raw) -> the plane of K-block kb's parity (par = kb & 1)
        float* const rawp = planes + par * PLANE_PAR;
        const bool up = EIG16_IS_UP(kb);
        const bool s1 = kb >= nkb0 + nkbu;
        const unsigned long long mu = 0ull - (unsigned long long)up, m1 = 0ull - (unsigned long long)(s1 && !up);
        const unsigned long long u = sb0 + ((sb1 - sb0) & m1) + ((sbu - sb0) & mu);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        const int sz = sz0 + ((sz1 - sz0) & (int)m1) + ((szu - sz0) & (int)mu);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(sz), 0x00020000);
        const unsigned in_range = (unsigned)((kb - nkb) >> 31);
        const unsigned chan = (unsigned)((kb - (up ? nkb0 : (s1 ? nkb0 + nkbu : 0))) * KC + tch) * (unsigned)((up ? HWh : HW) * 4);
        const unsigned coff = (chan & in_range) | (0x80000000u & ~in_range);
        const unsigned o0 = up ? (unsigned)uoff : (unsigned)roff[0], o1 = up ? 0xFFFFFFFFu : (unsigned)roff[1];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)rawp, 16, (int)__builtin_elementwise_add_sat(o0, coff), 0, 0, 0);
        if (lane < 44)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(rawp + 64 * 4), 16, (int)__builtin_elementwise_add_sat(o1, coff), 0, 0, 0);
    };
    const int rd_off = (2 * t_ty) * 24 + 2 * t_tx + 3;
    const int rd_off_u = t_ty * 24 + t_tx + 3;
    const float* const pbase_n = planes + rd_off;
    const float* const pbase_u = planes + rd_off_u;
    auto read_patch = [&](int kb, int par) __attribute__((always_inline)) {   // (round 4: read_patch) <- the plane of K-block kb's parity (par = kb & 1)
        const bool up = EIG16_IS_UP(kb);
        const float* const p00 = (up ? pbase_u : pbase_n) + par * PLANE_PAR;
        const float* const p10 = p00 - (up ? 24 : 0);
        const int cs = up ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* const pl = (i < 2 ? p00 : p10) + i * 24;
            const float* const pr = pl - cs;
            d[i][0] = pl[0]; d[i][1] = pl[1]; d[i][2] = pr[2]; d[i][3] = pr[3];
        }
    };
    auto read_patch_up = [&](int par) __attribute__((always_inline)) {   // read_patch of a K-block known to be an unpooled-source one: d[2] = d[1], d[.][2] = d[.][1]
        const float* const p00 = pbase_u + par * PLANE_PAR;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i == 2) continue;
            const float* const pl = p00 + (i == 3 ? 2 : i) * 24;
            d[i][0] = pl[0]; d[i][1] = pl[1]; d[i][2] = d[i][1]; d[i][3] = pl[2];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) d[2][j] = d[1][j];
    };
    auto transform = [&](float* vbuf) __attribute__((always_inline)) {   // (round 4: transform)
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
        }
        float* dst = vbuf + tch * W16_VS + (lane ^ ((tch & 1) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[(i * 4 + 0) * KC * W16_VS] = t[i][0] - t[i][2];
            dst[(i * 4 + 1) * KC * W16_VS] = t[i][1] + t[i][2];
            dst[(i * 4 + 2) * KC * W16_VS] = t[i][2] - t[i][1];
            dst[(i * 4 + 3) * KC * W16_VS] = t[i][1] - t[i][3];
        }
    };
    // the U slab of K-block kb: 512 NI chunks of 16 B, lane-linear -- NI instructions on each of the waves 8-15
    const int ut = tid - 512;
    auto dma_u = [&](int kb, float* ubuf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(ubuf + (j * 512 + (wv - 8) * 64) * 4), 16,
                                                     ut * 16, (int)((unsigned)(j * 512 * 16) + (unsigned)kb * (WINO_U_FLOATS * 4)), 0, 0);   // (scalar offset: no VALU)
    };

    // accumulators: position (xi, nu), N-tile ni
    f32x4 acc[4][NI];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[p][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int a_off = (xi * 4 * KC + q) * W16_VS + ((rg * 16 + col) ^ ((q & 1) << 4));   // V[pos = 4 xi + nu][ch = 4 ks + q][tile 16 rg + col, swizzled]
    const int b_off = ((xi * 4 * KC + q) * 16 + col) * NI;           // U[pos][ch][col][0..3]

    // ---- prologue: the planes of K-blocks 0 and 1 and the U slab of K-block 0 (waves 8-15); K-block 0 transformed (waves 0-7)
    const unsigned long long tq_setup = EIG_TIMING ? __builtin_readcyclecounter() : 0;
    if (!xf) {
        dma_raw(0, 0);
        dma_raw(1, 1);
        dma_u(0, Ub);
        EIG16_WAITCNT(0x0F70);
    }
    __syncthreads();
    if (xf) {
        read_patch(0, 0);
        EIG16_WAITCNT(0xC07F);
        transform(Vb);
    }
    EIG16_WAITCNT(0x0070);
    __syncthreads();
    const unsigned long long tq_k0 = EIG_TIMING ? __builtin_readcyclecounter() : 0;
    unsigned long long tq_k1 = 0, tq_x = 0, tq_y = 0;
    // [entry, set-up done, K loop start, K loop end, exchange barrier passed, y ready (gates start), exit, HW_ID | XCC_ID << 32]
    auto timeline = [&]() __attribute__((always_inline)) {
        if (EIG_TIMING && a.dbg && lane == 0) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* dd = a.dbg + ((size_t)blockIdx.x * 16 + wv) * 8;
            dd[0] = tq_entry; dd[1] = tq_setup; dd[2] = tq_k0; dd[3] = tq_k1; dd[4] = tq_x; dd[5] = tq_y; dd[6] = __builtin_readcyclecounter();
            dd[7] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
        }
    };

    // the cell state and the three peephole values of the lane's 4-pixel segment, fetched during the LAST K-block
    const int ra = xi >> 1, seg = xi & 1;   // this wave finishes output row parity ra of segment seg
    f32x4 st4[4];
    auto state_loads = [&]() __attribute__((always_inline)) {
        const int ch = nblk * 16 + col;
        const int gy = y0 + 4 * rg + ra + 2 * seg, gx = x0 + 4 * q;
        if (ch >= a.Cout || gy >= a.H || gx >= a.W) return;
        const size_t pix = (size_t)gy * a.W + gx, cb = ((size_t)eb * a.Cout + ch) * HW + pix, pb = (size_t)ch * HW + pix, ps = (size_t)a.Cout * HW;
        st4[0] = *reinterpret_cast<const f32x4*>(a.c_state + cb);
        st4[1] = *reinterpret_cast<const f32x4*>(a.peep + pb);
        st4[2] = *reinterpret_cast<const f32x4*>(a.peep + ps + pb);
        st4[3] = *reinterpret_cast<const f32x4*>(a.peep + 2 * ps + pb);
    };
    // an unpooled-source K-block: nu = 2 is a chain of zeros for every xi, xi = 2 entirely
    const unsigned wave_skip = xi == 2 ? 0xFu : 0x4u;
    // kind_tag: 0 = a full K-block (no skip tests in the instruction stream: they cost this kernel 4 %), 1 = an unpooled-source K-block, 2 = run time
    // role_tag: the wave's role as a compile-time constant (1: transforming wave, xi = 0 / 1; 0: U-fetching wave; 2: U-fetching wave with xi = 2, whose
    // positions an unpooled-source K-block skips entirely; 3: not known at compile time) -- the K loops exist once per role behind ONE wave-uniform
    // branch, so that no value defined on one role's path only (the transform's registers) needs a definition on the other's
    // par_tag: kb & 1 as a compile-time constant (0, 1: the LDS addresses of the K-block's buffers become instruction offsets) or 2 = run time
    // trim_tag: K-block kb + 1 is KNOWN to be an unpooled-source K-block -- its patch has rows 1 = 2 and columns 1 = 2, so the transform's row 2 and
    // column 2 are exact zeros that no wave reads (the skipped positions): 9 patch reads, 18 additions and 9 LDS writes instead of 16 / 32 / 16
    auto kiter = [&](const int kb, auto last_tag, auto kind_tag, auto role_tag, auto par_tag, auto trim_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr bool TRIM = EIG_W16_TRIM_UP && decltype(trim_tag)::value;
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr int ROLE = decltype(role_tag)::value;
        constexpr bool XF = ROLE == 1;
        constexpr int PARC = decltype(par_tag)::value;
        const int par = PARC == 2 ? (kb & 1) : PARC;
        // the positions an unpooled-source K-block skips are a compile-time constant of the role (no scalar tests, no operand reads for them)
        constexpr bool CSKIP = EIG_W16_CONST_SKIP && KIND == 1 && ROLE != 3;
        const unsigned skip = KIND == 0 ? 0u : CSKIP ? (ROLE == 2 ? 0xFu : 0x4u) : KIND == 1 ? wave_skip : (EIG16_IS_UP(kb) ? wave_skip : 0u);
        const float* const vcur = Vb + par * W16_V_FLOATS;
        const float* const ucur = Ub + par * WINO_U_FLOATS;
        float* const vnext = Vb + (1 - par) * W16_V_FLOATS + tch * W16_VS + (lane ^ ((tch & 1) << 4));
        // waves 8-15: the U slab of K-block kb + 1 and the plane of K-block kb + 2 (its buffer held K-block kb: read out by wave tch at the top of
        // K-block kb - 1, a barrier ago); waves 0-7: the patch of K-block kb + 1 (its plane landed before the barrier in front of this K-block)
        float t[4][4];
        float av[2][2];
        float bv[2][2][NI];
        auto fetch = [&](int c, int slot) __attribute__((always_inline)) {   // chunk c: k-step c >> 1, positions nu = 2 (c & 1), 2 (c & 1) + 1
            const int ks = c >> 1, pp = c & 1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (CSKIP && ((skip >> (2 * pp + u)) & 1)) continue;
                av[slot][u] = vcur[a_off + ((2 * pp + u) * KC + ks * 4) * W16_VS];
                const float* const bsrc = ucur + b_off + ((2 * pp + u) * KC + ks * 4) * 16 * NI;
                if constexpr (NI == 4) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bsrc);
                    bv[slot][u][0] = b4[0]; bv[slot][u][1] = b4[1]; bv[slot][u][2] = b4[2]; bv[slot][u][3] = b4[3];
                } else {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) bv[slot][u][ni] = bsrc[ni];
                }
            }
        };
        fetch(0, 0);   // (first: the operands of chunk 0 are in flight while the DMAs below are issued)
        if constexpr (!LAST) { if constexpr (!XF) { dma_u(kb + 1, Ub + (1 - par) * WINO_U_FLOATS); dma_raw(kb + 2, par); } else if constexpr (TRIM) read_patch_up(1 - par); else read_patch(kb + 1, 1 - par); }
        else if constexpr (EPI == EPI_LSTM) state_loads();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int pp = c & 1;
            if (c + 1 < 4) fetch(c + 1, (c + 1) & 1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if ((skip >> (2 * pp + u)) & 1) continue;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[2 * pp + u][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][u], bv[c & 1][u][ni], acc[2 * pp + u][ni], 0, 0, 0);
            }
            if constexpr (!LAST && XF) {
                if (c == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
                    }
                } else if (c < 3) {
#pragma unroll
                    for (int i = 2 * (c - 1); i < 2 * c; ++i) {
                        if (TRIM && i == 2) continue;
                        vnext[(i * 4 + 0) * KC * W16_VS] = t[i][0] - t[i][2];
                        vnext[(i * 4 + 1) * KC * W16_VS] = t[i][1] + t[i][2];
                        if (!TRIM) vnext[(i * 4 + 2) * KC * W16_VS] = t[i][2] - t[i][1];
                        vnext[(i * 4 + 3) * KC * W16_VS] = t[i][1] - t[i][3];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        EIG16_WAITCNT(0x0070);
        asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    };
    auto kloops = [&](auto role_tag) __attribute__((always_inline)) {
        int kb = 0;
        const std::false_type nl{};
        const std::integral_constant<int, 0> p0{};
        const std::integral_constant<int, 1> p1{};
        auto run = [&](const int end, auto kind_tag, auto trim_tag) __attribute__((always_inline)) {   // K-blocks [kb, end) of one kind, in (even, odd) pairs
            if (kb < end && (kb & 1)) { kiter(kb, nl, kind_tag, role_tag, p1, trim_tag); ++kb; }
            for (; kb + 1 < end; kb += 2) { kiter(kb, nl, kind_tag, role_tag, p0, trim_tag); kiter(kb + 1, nl, kind_tag, role_tag, p1, trim_tag); }
            if (kb < end) { kiter(kb, nl, kind_tag, role_tag, p0, trim_tag); ++kb; }
        };
        const int e1 = up_lo < nkb - 1 ? up_lo : nkb - 1, e2 = up_hi < nkb - 1 ? up_hi : nkb - 1;
        run(e1, std::integral_constant<int, 0>{}, std::false_type{});
        // unpooled-source K-blocks: all but the last of them transform the patch of another unpooled-source K-block (kb + 1 < up_hi)
        if constexpr (EIG_W16_TRIM_UP && decltype(role_tag)::value == 1) run(up_hi - 1 < e2 ? up_hi - 1 : e2, std::integral_constant<int, 1>{}, std::true_type{});
        run(e2, std::integral_constant<int, 1>{}, std::false_type{});
        run(nkb - 1, std::integral_constant<int, 0>{}, std::false_type{});
    };
    if (xf) kloops(std::integral_constant<int, 1>{});
    else if (EIG_W16_CONST_SKIP && EPI == EPI_LSTM && xi == 2) kloops(std::integral_constant<int, 2>{});
    else kloops(std::integral_constant<int, EIG_W16_CONST_SKIP ? 0 : 3>{});
    kiter(nkb - 1, std::true_type{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}, std::false_type{});
    if (EIG_TIMING) tq_k1 = __builtin_readcyclecounter();

    // ---- output transform.  Columns in-lane: c_xi,0 = (M_xi0 + M_xi1) + M_xi2, c_xi,1 = (M_xi1 - M_xi2) - M_xi3.
    f32x4 cc[2][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        cc[0][ni] = (acc[0][ni] + acc[1][ni]) + acc[2][ni];
        cc[1][ni] = (acc[1][ni] - acc[2][ni]) - acc[3][ni];
    }
    // Rows: y_0b = (c_0b + c_1b) + c_2b, y_1b = c_1b - (c_2b + c_3b).  Every wave publishes its c row; wave (rg, xi = 2 ra + seg) then finishes row
    // parity ra of segment seg = accumulator registers 2 seg, 2 seg + 1.
    float* const xb = lds;   // [16 waves][8 (b, ni)][64 lanes][4 registers] = 128 KB: V / U are dead (every wave is past the last barrier)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *reinterpret_cast<f32x4*>(xb + ((wv * 8 + b * 4 + ni) * 64 + lane) * 4) = cc[b][ni];
    __syncthreads();
    if (EIG_TIMING) tq_x = __builtin_readcyclecounter();
    if constexpr (EPI == EPI_LSTM) {
        float y[2][NI][2];   // [b = px][ni][window 2 seg + k]
    #pragma unroll
        for (int b = 0; b < 2; ++b)
    #pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int e = ((b * 4 + ni) * 64 + lane) * 4 + 2 * seg;   // registers 2 seg, 2 seg + 1 of the publishing wave's c row
                const f32x2 c1 = *reinterpret_cast<const f32x2*>(xb + (4 + rg) * 2048 + e), c2 = *reinterpret_cast<const f32x2*>(xb + (8 + rg) * 2048 + e);
                const f32x2 c03 = *reinterpret_cast<const f32x2*>(xb + ((ra ? 12 : 0) + rg) * 2048 + e);
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
        // a window in one lane -- ConvA's max_pooling_2d is the max over a lane's four values, ConvP stores whole 4 x 4-pixel patches
        const int k = xi;
        if (k >= NI) return;
        f32x4 y[2][2];   // [py][px]
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int e = ((b * 4 + k) * 64 + lane) * 4;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(xb + (0 + rg) * 2048 + e), c1 = *reinterpret_cast<const f32x4*>(xb + (4 + rg) * 2048 + e);
            const f32x4 c2 = *reinterpret_cast<const f32x4*>(xb + (8 + rg) * 2048 + e), c3 = *reinterpret_cast<const f32x4*>(xb + (12 + rg) * 2048 + e);
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
#undef EIG16_WAITCNT
#undef EIG16_IS_UP
}

}  // namespace eig
