// Micro-benchmark (measurement only, not product): the K-block of conv_wino.h's wino_kernel<4, EPI_LSTM, 8> rebuilt piece by piece -- 8 waves,
// ONE block per CU, 8 positions x 4 N-tiles of accumulators per wave, 8 chunks of 8 v_mfma_f32_16x16x4_f32 per K-block with the operand reads of
// the next chunk -- to see which piece costs the matrix pipe what when the pieces are added one at a time to the bare loop.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wino_loop_model scripts/wino_loop_model.hip && /tmp/wino_loop_model
// FEAT bits: 1 operands from LDS (else registers)   2 barrier per K-block         4 the transform's 8 ds_write2st64_b32 (chunks 2-5)
//            8 the transform's 32 additions          16 the 8 patch reads (ds_read2_b32) in front of the barrier
//            32 six LDS-DMA instructions per K-block (U slab 4, plane 2) + vmcnt(0) in front of the barrier
//            256 / 512 the staging slice of a chunk in front of / between its MFMA groups (default: behind them)
//            1024 instead of s_barrier: two LDS counters -- `written` signalled behind chunk 5, waited for at the top of the next K-block; `read`
//                 signalled at the end of a K-block, waited for in front of chunk 2 of the next (U DMA moved there)
//            64 operand reads TWO chunks ahead       128 A operands as one ds_read_b64 per chunk (pair-contiguous layout) instead of ds_read2st64_b32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int VS = 80, KC = 8, V_FLOATS = 16 * KC * VS, U_FLOATS = 16 * KC * 16 * 4, RAW = 18 * 24;

template <int FEAT>
__global__ void __launch_bounds__(512, 1) k(float* out, const float* __restrict__ src, int nkb)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Vb = lds;
    float* const Ub = lds + 2 * V_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv & 3, half = wv >> 2, q = lane >> 4, col = lane & 15;
    float* const rawp = lds + 2 * (V_FLOATS + U_FLOATS) + wv * RAW;
    // bit 1024: the barrier split into two LDS counters (written / read), signalled early and waited for late -- see main()
    unsigned* const cnt = reinterpret_cast<unsigned*>(lds + 2 * (V_FLOATS + U_FLOATS) + 8 * RAW);   // [0] K-blocks written, [1] K-blocks read (x 8 waves)
    auto signal = [&](int which) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(cnt + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
    };
    auto wait_for = [&](int which, unsigned target) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        for (int spin = 0; spin < (1 << 16); ++spin) {   // (bounded: a wrong protocol ends with wrong numbers, not with a hung GPU)
            const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (v >= target) break;
        }
        asm volatile("" ::: "memory");
    };
    for (int i = tid; i < 2 * (V_FLOATS + U_FLOATS) + 8 * RAW + 2; i += 512) lds[i] = i >= 2 * (V_FLOATS + U_FLOATS) + 8 * RAW ? 0.0f : (float)(i & 7) * 0.125f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 26, 0x00020000);
    f32x4 acc[8][4];
    for (int p = 0; p < 8; ++p) for (int n = 0; n < 4; ++n) acc[p][n] = (f32x4){0, 0, 0, 0};
    const int a_off = (half * 8 * KC + q) * VS + rg * 16 + col;
    const int a_off2 = ((half * 4 * KC + q) * VS + rg * 16 + col) * 2;   // (bit 128: V[half][pair][ch][tile][2])
    const int b_off = ((half * 8 * KC + q) * 16 + col) * 4;
    float d[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) d[i][j] = (float)(lane + i * 4 + j);
    const int rd_off = (2 * (lane >> 3)) * 24 + 2 * (lane & 7) + 3;
    constexpr int PF = (FEAT & 64) ? 2 : 1;
    float junk = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        const float* const vcur = Vb + (kb & 1) * V_FLOATS;
        const float* const ucur = Ub + (kb & 1) * U_FLOATS;
        float* const vnext = Vb + ((kb + 1) & 1) * V_FLOATS + wv * VS + lane;
        float av[PF + 1][2], bv[PF + 1][2][4];
        float t[4][4];
        auto fetch = [&](int c, int slot) __attribute__((always_inline)) {
            const int ks = c >> 2, pp = c & 3;
            if constexpr (!(FEAT & 1)) {
                for (int u = 0; u < 2; ++u) { av[slot][u] = 1.0f + c; for (int n = 0; n < 4; ++n) bv[slot][u][n] = 0.5f + n; }
                return;
            }
            if constexpr (FEAT & 128) {
                const f32x2 a2 = *reinterpret_cast<const f32x2*>(vcur + a_off2 + (pp * KC + ks * 4) * VS * 2);
                av[slot][0] = a2[0]; av[slot][1] = a2[1];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if constexpr (!(FEAT & 128)) av[slot][u] = vcur[a_off + ((2 * pp + u) * KC + ks * 4) * VS];
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(ucur + b_off + ((2 * pp + u) * KC + ks * 4) * 16 * 4);
                bv[slot][u][0] = b4[0]; bv[slot][u][1] = b4[1]; bv[slot][u][2] = b4[2]; bv[slot][u][3] = b4[3];
            }
        };
        if constexpr (FEAT & 1024) wait_for(0, 8u * (unsigned)kb);   // V / U of this K-block written by every wave
        if constexpr ((FEAT & 32) && !(FEAT & 1024)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Ub + ((kb + 1) & 1) * U_FLOATS + (j * 512 + wv * 64) * 4), 16,
                                                         (int)((unsigned)(tid * 16 + j * 8192) + (unsigned)(kb & 63) * 32768u), 0, 0, 0);
        }
        fetch(0, 0);
        if constexpr (PF == 2) fetch(1, 1);
        if constexpr (FEAT & 32) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)rawp, 16, (int)((unsigned)(lane * 16) + (unsigned)((kb & 63) * 8 + wv) * 65536u), 0, 0, 0);
            if (lane < 44)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(rawp + 256), 16, (int)((unsigned)(lane * 16 + 1024) + (unsigned)((kb & 63) * 8 + wv) * 65536u), 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int pp = c & 3;
            if (c + PF < 8) fetch(c + PF, (c + PF) % (PF + 1));
            auto staging = [&]() __attribute__((always_inline)) {
            if (c < 2) {
#pragma unroll
                for (int j = 2 * c; j < 2 * c + 2; ++j) {
                    if constexpr (FEAT & 8) { t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j]; }
                    else { t[0][j] = d[0][j]; t[1][j] = d[1][j]; t[2][j] = d[2][j]; t[3][j] = d[3][j]; }
                }
            } else if (c < 6) {
                const int i = c - 2;
                float v4[4] = {t[i][0], t[i][1], t[i][2], t[i][3]};
                if constexpr (FEAT & 8) { v4[0] = t[i][0] - t[i][2]; v4[1] = t[i][1] + t[i][2]; v4[2] = t[i][2] - t[i][1]; v4[3] = t[i][1] - t[i][3]; }
                if constexpr ((FEAT & 4) && (FEAT & 128)) {   // pair-contiguous V: two ds_write_b64 per row
                    float* const vp = Vb + ((kb + 1) & 1) * V_FLOATS + (wv * VS + lane) * 2;
                    *reinterpret_cast<f32x2*>(vp + (i * 2 + 0) * KC * VS * 2) = (f32x2){v4[0], v4[1]};
                    *reinterpret_cast<f32x2*>(vp + (i * 2 + 1) * KC * VS * 2) = (f32x2){v4[2], v4[3]};
                } else
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (FEAT & 4) vnext[(i * 4 + j) * KC * VS] = v4[j];
                    else asm volatile("" :: "v"(v4[j]));
                }
            }
            };
            if constexpr (FEAT & 1024) if (c == 2) {
                wait_for(1, 8u * (unsigned)kb);   // every wave has read K-block kb - 1: its buffers may be overwritten
                if constexpr (FEAT & 32) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Ub + ((kb + 1) & 1) * U_FLOATS + (j * 512 + wv * 64) * 4), 16,
                                                                 (int)((unsigned)(tid * 16 + j * 8192) + (unsigned)(kb & 63) * 32768u), 0, 0, 0);
                }
            }
            if constexpr (FEAT & 256) staging();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[2 * pp + u][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c % (PF + 1)][u], bv[c % (PF + 1)][u][n], acc[2 * pp + u][n], 0, 0, 0);
                if constexpr (FEAT & 512) if (u == 0) staging();
            }
            if constexpr (!(FEAT & (256 | 512))) staging();
            if constexpr (FEAT & 1024) if (c == 5) { __builtin_amdgcn_s_waitcnt(0x0070); signal(0); }   // my V rows and U pieces of K-block kb + 1 are in LDS
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (FEAT & 32) __builtin_amdgcn_s_waitcnt(0x0070);
        else if constexpr (FEAT & (4 | 2 | 1024)) __builtin_amdgcn_s_waitcnt(0xC07F);
        if constexpr (FEAT & 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { d[i][0] = rawp[rd_off + i * 24]; d[i][1] = rawp[rd_off + i * 24 + 1]; d[i][2] = rawp[rd_off + i * 24 + 2]; d[i][3] = rawp[rd_off + i * 24 + 3]; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(d[i][j]));
        }
        if constexpr (FEAT & 1024) signal(1);   // (the operands of this K-block's last chunk have arrived: lgkmcnt(0) above)
        else if constexpr (FEAT & 2) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    }
    float s = junk;
    for (int p = 0; p < 8; ++p) for (int n = 0; n < 4; ++n) s += acc[p][n][0] + acc[p][n][1] + acc[p][n][2] + acc[p][n][3];
    out[blockIdx.x * 512 + tid] = s + d[0][0];
}

// The same K-block on SIXTEEN waves (four per SIMD, <= 128 VGPRs each): every wave multiplies 4 positions x 4 N-tiles (32 MFMAs per K-block, 4 chunks
// of 8), waves 0-7 transform one channel each and fetch its plane, waves 8-15 fetch the U slab.  FEAT bits as above (1, 2, 4, 8, 16, 32).
template <int FEAT>
__global__ void __launch_bounds__(1024, 1) k16(float* out, const float* __restrict__ src, int nkb)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Vb = lds;
    float* const Ub = lds + 2 * V_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv & 3, quarter = wv >> 2, q = lane >> 4, col = lane & 15;
    const bool xf = wv < 8;
    float* const rawp = lds + 2 * (V_FLOATS + U_FLOATS) + (wv & 7) * RAW;
    for (int i = tid; i < 2 * (V_FLOATS + U_FLOATS) + 8 * RAW; i += 1024) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 26, 0x00020000);
    f32x4 acc[4][4];
    for (int p = 0; p < 4; ++p) for (int n = 0; n < 4; ++n) acc[p][n] = (f32x4){0, 0, 0, 0};
    const int a_off = (quarter * 4 * KC + q) * VS + rg * 16 + col;
    const int b_off = ((quarter * 4 * KC + q) * 16 + col) * 4;
    float d[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) d[i][j] = (float)(lane + i * 4 + j);
    const int rd_off = (2 * (lane >> 3)) * 24 + 2 * (lane & 7) + 3;
    for (int kb = 0; kb < nkb; ++kb) {
        const float* const vcur = Vb + (kb & 1) * V_FLOATS;
        const float* const ucur = Ub + (kb & 1) * U_FLOATS;
        float* const vnext = Vb + ((kb + 1) & 1) * V_FLOATS + (wv & 7) * VS + lane;
        float av[2][2], bv[2][2][4];
        float t[4][4];
        auto fetch = [&](int c, int slot) __attribute__((always_inline)) {
            const int ks = c >> 1, pp = c & 1;
            if constexpr (!(FEAT & 1)) {
                for (int u = 0; u < 2; ++u) { av[slot][u] = 1.0f + c; for (int n = 0; n < 4; ++n) bv[slot][u][n] = 0.5f + n; }
                return;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                av[slot][u] = vcur[a_off + ((2 * pp + u) * KC + ks * 4) * VS];
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(ucur + b_off + ((2 * pp + u) * KC + ks * 4) * 16 * 4);
                bv[slot][u][0] = b4[0]; bv[slot][u][1] = b4[1]; bv[slot][u][2] = b4[2]; bv[slot][u][3] = b4[3];
            }
        };
        if constexpr (FEAT & 32) if (!xf) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(Ub + ((kb + 1) & 1) * U_FLOATS + (j * 512 + (wv - 8) * 64) * 4), 16,
                                                         (int)((unsigned)((tid - 512) * 16 + j * 8192) + (unsigned)(kb & 63) * 32768u), 0, 0, 0);
        }
        fetch(0, 0);
        if constexpr (FEAT & 32) if (xf) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)rawp, 16, (int)((unsigned)(lane * 16) + (unsigned)((kb & 63) * 8 + wv) * 65536u), 0, 0, 0);
            if (lane < 44)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(rawp + 256), 16, (int)((unsigned)(lane * 16 + 1024) + (unsigned)((kb & 63) * 8 + wv) * 65536u), 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int pp = c & 1;
            if (c + 1 < 4) fetch(c + 1, (c + 1) & 1);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[2 * pp + u][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][u], bv[c & 1][u][n], acc[2 * pp + u][n], 0, 0, 0);
            if (xf) {
                if (c == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (FEAT & 8) { t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j]; }
                        else { t[0][j] = d[0][j]; t[1][j] = d[1][j]; t[2][j] = d[2][j]; t[3][j] = d[3][j]; }
                    }
                } else if (c < 3) {
#pragma unroll
                    for (int i = 2 * (c - 1); i < 2 * c; ++i) {
                        float v4[4] = {t[i][0], t[i][1], t[i][2], t[i][3]};
                        if constexpr (FEAT & 8) { v4[0] = t[i][0] - t[i][2]; v4[1] = t[i][1] + t[i][2]; v4[2] = t[i][2] - t[i][1]; v4[3] = t[i][1] - t[i][3]; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if constexpr (FEAT & 4) vnext[(i * 4 + j) * KC * VS] = v4[j];
                            else asm volatile("" :: "v"(v4[j]));
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (FEAT & 32) __builtin_amdgcn_s_waitcnt(0x0070);
        else if constexpr (FEAT & (4 | 2)) __builtin_amdgcn_s_waitcnt(0xC07F);
        if (xf) {
            if constexpr (FEAT & 16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { d[i][0] = rawp[rd_off + i * 24]; d[i][1] = rawp[rd_off + i * 24 + 1]; d[i][2] = rawp[rd_off + i * 24 + 2]; d[i][3] = rawp[rd_off + i * 24 + 3]; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(d[i][j]));
            }
        }
        if constexpr (FEAT & 2) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    }
    float s = 0;
    for (int p = 0; p < 4; ++p) for (int n = 0; n < 4; ++n) s += acc[p][n][0] + acc[p][n][1] + acc[p][n][2] + acc[p][n][3];
    out[blockIdx.x * 1024 + tid] = s + d[0][0];
}

template <int FEAT> void run16(const char* name)
{
    const int blocks = 256 * 4, nkb = 600;
    const int lds = (2 * (V_FLOATS + U_FLOATS) + 8 * RAW) * 4 + 16;
    hipFuncSetAttribute((const void*)k16<FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float *d, *src; hipMalloc(&d, (size_t)blocks * 1024 * 4); hipMalloc(&src, 1 << 26); hipMemset(src, 0, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k16<FEAT><<<blocks, 1024, lds>>>(d, src, 20);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k16<FEAT><<<blocks, 1024, lds>>>(d, src, nkb);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 16 * nkb * 32 * 2048.0;
    printf("16w %3d %-74s %7.2f ms  %6.1f TFLOP/s  %.3f of 157.3\n", FEAT, name, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 157.3);
    fflush(stdout);
    hipFree(d); hipFree(src);
}

template <int FEAT> void run(const char* name)
{
    const int blocks = 256 * 4, nkb = 600;
    const int lds = (2 * (V_FLOATS + U_FLOATS) + 8 * RAW) * 4 + 16;
    hipFuncSetAttribute((const void*)k<FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float *d, *src; hipMalloc(&d, (size_t)blocks * 512 * 4); hipMalloc(&src, 1 << 26); hipMemset(src, 0, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<FEAT><<<blocks, 512, lds>>>(d, src, 20);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<FEAT><<<blocks, 512, lds>>>(d, src, nkb);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 * nkb * 64 * 2048.0;
    printf("%3d %-78s %7.2f ms  %6.1f TFLOP/s  %.3f of 157.3\n", FEAT, name, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 157.3);
    fflush(stdout);
    hipFree(d); hipFree(src);
}

int main()
{
    run16<0>("sixteen waves: operands in registers");
    run16<1>("sixteen waves: operands from LDS");
    run16<1 | 2>("sixteen waves: LDS + barrier");
    run16<1 | 2 | 4 | 8 | 16>("sixteen waves: LDS + barrier + whole transform");
    run16<1 | 2 | 32>("sixteen waves: LDS + barrier + DMA");
    run16<1 | 2 | 4 | 8 | 16 | 32>("sixteen waves: everything");
    run16<1 | 4 | 8 | 16 | 32>("sixteen waves: everything but the barrier");
    run<0>("operands in registers, nothing else");
    run<2>("registers + barrier");
    run<1>("operands from LDS, one chunk ahead");
    run<1 | 64>("operands from LDS, two chunks ahead");
    run<1 | 128>("LDS, A as one ds_read_b64 per chunk");
    run<1 | 2>("LDS + barrier");
    run<1 | 2 | 4>("LDS + barrier + V writes");
    run<1 | 2 | 8>("LDS + barrier + additions");
    run<1 | 2 | 4 | 8>("LDS + barrier + V writes + additions");
    run<1 | 2 | 16>("LDS + barrier + patch reads");
    run<1 | 2 | 4 | 8 | 16>("LDS + barrier + whole transform");
    run<1 | 2 | 32>("LDS + barrier + DMA");
    run<1 | 2 | 4 | 8 | 16 | 32>("everything (the kernel's full K-block)");
    run<1 | 4 | 8 | 16 | 32>("everything but the barrier");
    run<1 | 2 | 4 | 8 | 16 | 32 | 64>("everything, operands two chunks ahead");
    run<1 | 4 | 8 | 16 | 32 | 1024>("everything, the barrier split into written / read counters in LDS");
    run<1 | 1024>("LDS + split barrier");
    run<1 | 2 | 8 | 256>("LDS + barrier + additions, in FRONT of the chunk's MFMAs");
    run<1 | 2 | 8 | 512>("LDS + barrier + additions, BETWEEN the chunk's two MFMA groups");
    run<1 | 2 | 4 | 8 | 16 | 32 | 256>("everything, staging slice in front of the chunk's MFMAs");
    run<1 | 2 | 4 | 8 | 16 | 32 | 512>("everything, staging slice between the two MFMA groups");
    run<1 | 2 | 4 | 8 | 16 | 32 | 128>("everything, A pairs as ds_read_b64 / V rows as ds_write_b64");
    run<1 | 2 | 4 | 128>("LDS + barrier + V writes, A pairs b64 / writes b64");
    run<1 | 2 | 4 | 8 | 16 | 128>("LDS + barrier + whole transform, b64 pairs");
    return 0;
}
