// wino4_kernels.hip -- translation unit of the Winograd F(4x4, 3x3) kernels (conv_wino4.h), 16 x 32-pixel blocks, and the launcher (wino_launch.h); the 32 x 16-pixel
// (tall) instantiations are a unit of their own (wino4t_kernels.hip)
#include "conv_wino4.h"

#include <unordered_set>

namespace eig {

hipError_t launch_wino4_tall(int NI, int epi, const ConvArgs& a, int grid, hipStream_t st);   // wino4t_kernels.hip
hipError_t launch_wino4_half(int NI, int epi, const ConvArgs& a, int grid, hipStream_t st);   // wino4h_kernels.hip
hipError_t launch_wino4_pack(int NI, int epi, const ConvArgs& a, int grid, hipStream_t st);   // wino4p_kernels.hip

hipError_t launch_wino4(int NI, int epi, int shape, const ConvArgs& a, int grid, hipStream_t st)
{
    auto go = [&](auto kern) -> hipError_t {
        static std::unordered_set<const void*> attr_done;   // (an engine handle is not thread-safe anyway: one rank, one host thread)
        if (attr_done.insert((const void*)kern).second) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, wino4_lds_bytes());
        hipLaunchKernelGGL(kern, dim3(grid), dim3(W4_THREADS), wino4_lds_bytes(), st, a);
        return hipGetLastError();
    };
    if (NI != 3 && NI != 4) return hipErrorInvalidConfiguration;
    if (shape == W4_TALL) return launch_wino4_tall(NI, epi, a, grid, st);
    if (shape == W4_HALF) return launch_wino4_half(NI, epi, a, grid, st);
    if (shape == W4_PACK) return launch_wino4_pack(NI, epi, a, grid, st);
    {
        if (epi == EPI_LSTM) return NI == 4 ? go(wino4_kernel<4, EPI_LSTM>) : hipErrorInvalidConfiguration;
        if (epi == EPI_CONVA) return NI == 4 ? go(wino4_kernel<4, EPI_CONVA>) : go(wino4_kernel<3, EPI_CONVA>);
        if (epi == EPI_CONVP) return NI == 4 ? go(wino4_kernel<4, EPI_CONVP>) : go(wino4_kernel<3, EPI_CONVP>);
    }
    return hipErrorInvalidConfiguration;
}

}  // namespace eig
