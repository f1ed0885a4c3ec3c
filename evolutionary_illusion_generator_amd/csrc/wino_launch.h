// wino_launch.h -- what the engine's translation unit (eigen_engine.hip) needs to know of the Winograd kernels: their packed-weight geometry and two launchers.
// The kernels themselves are compiled in translation units of their own (wino4_kernels.hip: conv_wino4.h; wino16_kernels.hip: conv_wino16.h), so that the three hipcc
// runs of a build go side by side (__graft_entry__.build(): 55 s instead of 137 s in one unit).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_mfma.h"   // ConvArgs, KC, EPI_*

namespace eig {

// F(4x4, 3x3), conv_wino4.h: twelve waves per block
constexpr int W4_WAVES = 12;
constexpr int W4_THREADS = 64 * W4_WAVES;
constexpr int W4_KC = 4;
constexpr int W4_NPOS = 36;
constexpr int wino4_u_floats(int NI) { return W4_NPOS * 4 * 16 * NI; }   // one 4-channel K-block of the packed weights: [36 pos][4 ch][16 cols][NI] (the buffer ends in one K-block of padding: the fetch runs one K-block past the end)
// F(2x2, 3x3), conv_wino16.h: sixteen waves per block
constexpr int WINO16_THREADS = 1024;
constexpr int W16_WAVES = 16;
constexpr int wino_u_floats(int NI) { return 16 * KC * 16 * NI; }   // one packed K-block of F(2x2) weights: [16 pos][8 ch][16 cols][NI]: 8192 floats for NI = 4
static_assert(KC == 8, "F(2x2) operators: 8-channel K-blocks");

// grid blocks of wino4_kernel<NI, epi, shape> / wino16_kernel<NI, epi> on stream st (NI = 3 or 4; epi = EPI_LSTM (NI = 4 only), EPI_CONVA, EPI_CONVP); the first launch of an
// instantiation sets its dynamic-LDS attribute.  shape: W4_WIDE 16 x 32-pixel blocks, W4_TALL 32 x 16, W4_HALF 8 x 32 (six waves)
enum { W4_WIDE = 0, W4_TALL = 1, W4_HALF = 2 };
hipError_t launch_wino4(int NI, int epi, int shape, const ConvArgs& a, int grid, hipStream_t st);
hipError_t launch_wino16(int NI, int epi, const ConvArgs& a, int grid, hipStream_t st);

}  // namespace eig
