// Micro-benchmark: rate of v_mfma_f32_16x16x4_f32 (16 independent accumulators per wave, as in the conv kernel) as a
// function of the number of waves resident per SIMD (1..4, set through the dynamic LDS size of 4-wave blocks).
// Answers whether ONE wave can keep the matrix pipe of its SIMD busy on its own.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: operands in registers; 1: operands re-read from LDS every step (conv-like gather);
                     // 2: same reads issued one step ahead (two register sets, scheduling barriers); 3: reads but operands unused
__global__ void __launch_bounds__(256, 4) k(float* out, int iters)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
    float a[4] = {1.f, 2.f, 3.f, 4.f}, b[4] = {.5f, .25f, .125f, 1.f};
    float junk = 0;
    const int base = (lane & 15) * 18 + (lane >> 4) * 3;
    if (MODE == 2) {
        float av[2][4], bv[2][4];
        auto ld = [&](int it, int st, float* a_, float* b_) {
#pragma unroll
            for (int m = 0; m < 4; ++m) a_[m] = lds[base + st * 64 + m * 8 + (it & 3) * 1024];
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + 4096 + lane * 4 + st * 256 + (it & 3) * 1024 - (lane * 4 & ~1023));
#pragma unroll
            for (int n = 0; n < 4; ++n) b_[n] = b4[n];
        };
        ld(0, 0, av[0], bv[0]);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < 9; ++st) {
                ld(st == 8 ? it + 1 : it, (st + 1) % 9, av[(st + 1) & 1], bv[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st & 1][m], bv[st & 1][n], acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // 9 steps per iteration is odd: swap the sets so that the parity keeps matching
            for (int m = 0; m < 4; ++m) { float t = av[0][m]; av[0][m] = av[1][m]; av[1][m] = t; t = bv[0][m]; bv[0][m] = bv[1][m]; bv[1][m] = t; }
        }
    } else
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            if (MODE == 3) {  // the LDS traffic without the dependency: operands stay constant, the loaded values are summed on the side
                float z = 0;
#pragma unroll
                for (int m = 0; m < 4; ++m) z += lds[base + st * 64 + m * 8 + (it & 3) * 1024];
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + 4096 + lane * 4 + st * 256 + (it & 3) * 1024 - (lane * 4 & ~1023));
                junk += z + b4[0] + b4[1] + b4[2] + b4[3];
            }
            if (MODE == 1) {
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = lds[base + st * 64 + m * 8 + (it & 3) * 1024];
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds + 4096 + lane * 4 + st * 256 + (it & 3) * 1024 - (lane * 4 & ~1023));
#pragma unroll
                for (int n = 0; n < 4; ++n) b[n] = b4[n];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = s + junk;
}

template <int MODE> void run(const char* name, int occ)
{
    const int blocks = 256 * occ;
    const int lds = (160 * 1024) / occ - 1024;   // exactly `occ` blocks fit per CU
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    const int iters = 8000 / occ;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256, lds>>>(d, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256, lds>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 9 * 16 * 2048.0;
    printf("%-26s waves/SIMD=%d  %.2f ms  %.1f TFLOP/s  (%.3f of 157.3)\n", name, occ, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 157.3);
    hipFree(d);
}

int main()
{
    for (int occ = 1; occ <= 4; ++occ) run<0>("operands in registers", occ);
    for (int occ = 1; occ <= 4; ++occ) run<1>("operands from LDS", occ);
    for (int occ = 1; occ <= 4; ++occ) run<2>("LDS, one step ahead", occ);
    for (int occ = 1; occ <= 4; ++occ) run<3>("LDS reads, unused", occ);
    return 0;
}
