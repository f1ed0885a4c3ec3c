// wino16_kernels.hip -- translation unit of the Winograd F(2x2, 3x3) kernels (conv_wino16.h) and their launcher (wino_launch.h)
#include "conv_wino16.h"

#include <unordered_set>

namespace eig {

hipError_t launch_wino16(int NI, int epi, const ConvArgs& a, int grid, hipStream_t st)
{
    auto go = [&](auto kern, int ni) -> hipError_t {
        const int lds = wino16_lds_bytes(ni);
        static std::unordered_set<const void*> attr_done;
        if (attr_done.insert((const void*)kern).second) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WINO16_THREADS), lds, st, a);
        return hipGetLastError();
    };
    if (NI != 3 && NI != 4) return hipErrorInvalidConfiguration;
    if (epi == EPI_LSTM) return NI == 4 ? go(wino16_kernel<4, EPI_LSTM>, 4) : hipErrorInvalidConfiguration;
    if (epi == EPI_CONVA) return NI == 4 ? go(wino16_kernel<4, EPI_CONVA>, 4) : go(wino16_kernel<3, EPI_CONVA>, 3);
    if (epi == EPI_CONVP) return NI == 4 ? go(wino16_kernel<4, EPI_CONVP>, 4) : go(wino16_kernel<3, EPI_CONVP>, 3);
    return hipErrorInvalidConfiguration;
}

}  // namespace eig
