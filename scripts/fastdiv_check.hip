// fastdiv_check.hip -- EXHAUSTIVE check (measurement / verification tool, not product) of the division the gate epilogue's tanh uses (csrc/conv_mfma.h: det_tanhf,
// r = 1 - 2 / (t + 1) with t = det_expf(2 |x|), |x| in [0.625, 10]): is   q = 2 rcp(d);  q = fmaf(fmaf(-d, q, 2), rcp(d), q)   (v_rcp_f32 + two fused multiply-adds)
// the SAME fp32 number as the IEEE division 2.0f / d for every d the kernel can see -- and is det_tanhf built on it the same function, bit for bit, for ALL 2^32 inputs?
// Only then may the kernels use it while the oracle (oracle/eig_oracle.c) keeps the plain division.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o scripts/_timing/fastdiv_check scripts/fastdiv_check.hip && scripts/_timing/fastdiv_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "../evolutionary_illusion_generator_amd/csrc/conv_mfma.h"

__device__ __forceinline__ float div2_fast(float d, int steps)
{
    const float r = __builtin_amdgcn_rcpf(d);
    float q = 2.0f * r;
    for (int i = 0; i < steps; ++i) q = fmaf(fmaf(-d, q, 2.0f), r, q);
    return q;
}
__device__ __forceinline__ float tanh_with(float x, int steps)   // det_tanhf with the candidate division
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
    const float t = eig::det_expf(2.0f * ax);
    const float r = 1.0f - div2_fast(t + 1.0f, steps);
    return x < 0.0f ? -r : r;
}
__global__ void check_div(unsigned long long* bad, unsigned* first_bad, uint32_t lo, uint32_t hi, int steps)
{
    for (uint64_t b = lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b <= hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float d = __uint_as_float((uint32_t)b);
        const float q0 = 2.0f / d, q1 = div2_fast(d, steps);
        if (__float_as_uint(q0) != __float_as_uint(q1)) { if (atomicAdd(bad, 1ull) == 0) *first_bad = (uint32_t)b; }
    }
}
__global__ void check_tanh(unsigned long long* bad, unsigned* first_bad, int steps)
{
    for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const float t0 = eig::det_tanhf(x), t1 = tanh_with(x, steps);
        if (__float_as_uint(t0) != __float_as_uint(t1) && !(t0 != t0 && t1 != t1)) { if (atomicAdd(bad, 1ull) == 0) *first_bad = (uint32_t)b; }
    }
}
int main()
{
    unsigned long long* bad; unsigned* fb;
    hipMalloc(&bad, 8); hipMalloc(&fb, 4);
    const float dlo = 1.0f, dhi = 1e9f;   // t + 1 lies in [1 + e^1.25, 1 + e^20] = [4.49, 4.9e8]; checked on a superset
    uint32_t lo, hi; memcpy(&lo, &dlo, 4); memcpy(&hi, &dhi, 4);
    for (int steps = 1; steps <= 2; ++steps) {
        unsigned long long h = 0; unsigned f = 0;
        hipMemset(bad, 0, 8); hipMemset(fb, 0, 4);
        hipLaunchKernelGGL(check_div, dim3(4096), dim3(256), 0, 0, bad, fb, lo, hi, steps);
        hipDeviceSynchronize();
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost);
        printf("2 / d, every fp32 d in [%g, %g] (%u values), rcp + %d refinement step(s): %llu differ from the IEEE division (first: 0x%08x)\n", dlo, dhi, hi - lo + 1, steps, h, f);
        hipMemset(bad, 0, 8); hipMemset(fb, 0, 4);
        hipLaunchKernelGGL(check_tanh, dim3(8192), dim3(256), 0, 0, bad, fb, steps);
        hipDeviceSynchronize();
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost);
        printf("det_tanhf, ALL 2^32 inputs, division as rcp + %d step(s): %llu outputs differ from the shipped det_tanhf (first input: 0x%08x)\n", steps, h, f);
    }
    return 0;
}
