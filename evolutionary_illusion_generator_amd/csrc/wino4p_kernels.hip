// wino4p_kernels.hip -- translation unit of the Winograd F(4x4, 3x3) kernels on half blocks of sixteen packed tiles (conv_wino4.h: PACK); called through launch_wino4
#include "conv_wino4.h"

#include <unordered_set>

namespace eig {

hipError_t launch_wino4_pack(int NI, int epi, const ConvArgs& a, int grid, hipStream_t st)
{
    auto go = [&](auto kern) -> hipError_t {
        static std::unordered_set<const void*> attr_done;
        if (attr_done.insert((const void*)kern).second) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, wino4_lds_bytes());
        hipLaunchKernelGGL(kern, dim3(grid), dim3(W4_THREADS / 2), wino4_lds_bytes(), st, a);
        return hipGetLastError();
    };
    if (epi == EPI_LSTM) return NI == 4 ? go(wino4_kernel<4, EPI_LSTM, false, true, true>) : hipErrorInvalidConfiguration;
    if (epi == EPI_CONVP) return NI == 4 ? go(wino4_kernel<4, EPI_CONVP, false, true, true>) : go(wino4_kernel<3, EPI_CONVP, false, true, true>);
    return hipErrorInvalidConfiguration;
}

}  // namespace eig
