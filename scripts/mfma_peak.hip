// Micro-benchmark: attainable rate of v_mfma_f32_16x16x4_f32 in the shape the conv kernel uses it
// (16 independent accumulators per wave, 4 A + 4 B operands per step), with and without the LDS operand reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: registers only, 1: operands from LDS (conflict-free), 2: conv-like addresses, 3: + runtime offset adds, 4: + 12 VALU every other step
__global__ void __launch_bounds__(256, 2) k(float* out, int iters, int stride)
{
    int mo[4]; for (int m = 0; m < 4; ++m) mo[m] = (stride * m * 8) & 1023;  // runtime values: v_add per read
    unsigned long long junk = (unsigned long long)out;
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 12288; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
    float a[4] = {1.f, 2.f, 3.f, 4.f}, b[4] = {.5f, .25f, .125f, 1.f};
    int base = (MODE == 2) ? ((lane & 15) * stride + (lane >> 4) * 3) : lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            if (MODE >= 1) {
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = lds[base + st * 64 + (MODE >= 3 ? mo[m] : m * 8) + (it & 3) * 1024];
#pragma unroll
                for (int n = 0; n < 4; ++n) b[n] = lds[4096 + lane + st * 256 + n * 16 + (it & 3) * 1024];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
            if (MODE == 4 && (st & 1) == 0) {  // 64-bit address arithmetic of one DMA instruction
#pragma unroll
                for (int z = 0; z < 6; ++z) { junk = junk * 2654435761ull + (unsigned)lane * (unsigned)(z + 3); asm volatile("" : "+v"(junk)); }
            }
        }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = s + (float)(junk & 1);
}

template <int MODE> void run(const char* name, int blocks, int stride)
{
    float* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256, 49152>>>(d, 100, stride);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256, 49152>>>(d, iters, stride);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 9 * 16 * 2048.0;
    printf("%-34s blocks=%5d stride=%2d  %.2f ms  %.1f TFLOP/s\n", name, blocks, stride, ms, flops / ms * 1e-9);
    hipFree(d);
}

int main()
{
    for (int blocks : {512, 2048}) {
        run<0>("regs only", blocks, 0);
        run<1>("LDS operands, conflict-free", blocks, 0);
        run<2>("LDS operands, stride 18 gather", blocks, 18);
        run<3>("+ runtime offset adds", blocks, 18);
        run<4>("+ 64-bit VALU every other step", blocks, 18);
    }
    return 0;
}
