// wino4h_kernels.hip -- translation unit of the Winograd F(4x4, 3x3) kernels on 8 x 32-pixel half blocks (conv_wino4.h: HALF); called through launch_wino4
#include "conv_wino4.h"

#include <unordered_set>

#ifndef EIG_W4_NSPLIT
#define EIG_W4_NSPLIT 1
#endif

namespace eig {

hipError_t launch_wino4_half(int NI, int epi, const ConvArgs& a, int grid, hipStream_t st)
{
    // 64-column ConvLSTM / ConvP: twelve waves, each multiplying two of the four N-tiles (conv_wino4.h: NSPLIT); the others six waves.  -DEIG_W4_NSPLIT=0: six waves
    // everywhere (measurement builds: profiles/r06_y_nsplit_ab.txt)
    auto go = [&](auto kern, int threads) -> hipError_t {
        static std::unordered_set<const void*> attr_done;
        if (attr_done.insert((const void*)kern).second) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, wino4_lds_bytes());
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), wino4_lds_bytes(), st, a);
        return hipGetLastError();
    };
    constexpr bool NS = EIG_W4_NSPLIT != 0;
    constexpr int T12 = NS ? W4_THREADS : W4_THREADS / 2, T6 = W4_THREADS / 2;
    if (epi == EPI_LSTM) return NI == 4 ? go(wino4_kernel<4, EPI_LSTM, false, true, false, NS>, T12) : hipErrorInvalidConfiguration;
    if (epi == EPI_CONVA) return NI == 4 ? go(wino4_kernel<4, EPI_CONVA, false, true>, T6) : go(wino4_kernel<3, EPI_CONVA, false, true>, T6);
    if (epi == EPI_CONVP) return NI == 4 ? go(wino4_kernel<4, EPI_CONVP, false, true, false, NS>, T12) : go(wino4_kernel<3, EPI_CONVP, false, true, false>, T6);
    return hipErrorInvalidConfiguration;
}

}  // namespace eig
