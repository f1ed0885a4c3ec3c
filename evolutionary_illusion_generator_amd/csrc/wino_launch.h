// wino_launch.h -- what the engine's translation unit (eigen_engine.hip) needs to know of the Winograd kernels: their packed-weight geometry and the launcher.
// The kernels themselves are compiled in translation units of their own (wino4_kernels.hip: the wide blocks; wino4t_kernels.hip: the tall ones; wino4h_kernels.hip: the half blocks; wino4p_kernels.hip: half blocks of packed tiles), so that
// the hipcc runs of a build go side by side (__graft_entry__.build()).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_mfma.h"   // ConvArgs, EPI_*

namespace eig {

// F(4x4, 3x3), conv_wino4.h: twelve waves per block
constexpr int W4_WAVES = 12;
constexpr int W4_THREADS = 64 * W4_WAVES;
constexpr int W4_KC = 4;
constexpr int W4_NPOS = 36;
constexpr int wino4_u_floats(int NI) { return W4_NPOS * 4 * 16 * NI; }   // one 4-channel K-block of the packed weights: [36 pos][4 ch][16 cols][NI] (the buffer ends in one K-block of padding: the fetch runs one K-block past the end)

// grid blocks of wino4_kernel<NI, epi, shape> on stream st (NI = 3 or 4; epi = EPI_LSTM (NI = 4 only), EPI_CONVA, EPI_CONVP); the first launch of an
// instantiation sets its dynamic-LDS attribute.  shape: W4_WIDE 16 x 32-pixel blocks, W4_TALL 32 x 16, W4_HALF 8 x 32 (one region; six waves, twelve for 64-column ConvLSTMs / ConvPs),
// W4_PACK half blocks of packed tiles for 16- / 20-column maps (ConvLSTM / ConvP only)
enum { W4_WIDE = 0, W4_TALL = 1, W4_HALF = 2, W4_PACK = 3 };
hipError_t launch_wino4(int NI, int epi, int shape, const ConvArgs& a, int grid, hipStream_t st);

}  // namespace eig
