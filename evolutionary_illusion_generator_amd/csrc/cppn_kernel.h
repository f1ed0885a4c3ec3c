// cppn_kernel.h -- batched CPPN render: one thread per pixel, one genome per blockIdx.y.
//
// Replaces get_image_from_cppn (/root/reference/generate_illusion.py:372-460) including the
// pytorch_neat create_cppn evaluation it calls (:384-397, :436-445), the background fill where
// x_mat == -1 (:398-401, :448-451) and the uint8 quantisation (:403, :412, :457).
//
// Arithmetic is float64 exactly as the reference's torch.float64 tensors: products and sums are separate
// roundings in connection order (the translation unit is compiled with -ffp-contract=off),
// value = act(response * sum + bias); the transcendental activations are the canonical kernels of det_math64.h.  HBM traffic is the algorithmic minimum: the float64 coordinate planes
// are read once per genome (coalesced, one pixel per lane) and C uint8 planes are written.
// The genome "program" (nodes in topological order + CSR edges) sits in LDS; node values live in LDS as
// [node][thread] columns so the data-dependent gather src -> value is a conflict-free ds_read_b64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "det_math64.h"

namespace eig {

constexpr int CPPN_THREADS = 128;

struct CppnArgs {
    const int32_t* node_off;  // [G+1]
    const int32_t* edge_off;  // [total_nodes+1]
    const uint8_t* node_act;
    const double* node_bias;
    const double* node_resp;
    const int32_t* edge_src;
    const double* edge_w;
    const int32_t* out_node;  // [G*c_out]
    const double* planes;     // [n_planes][N]
    int n_planes;
    int N;                    // H*W
    int c_out;                // outputs evaluated per genome
    int c_dim;                // channels written per genome
    int bg;                   // 0 / 1
    int mode;                 // 0: gradient colour/gray; 1: gray rounded (gradient==0, c_dim==1); 2: 5-colour palette; 4: h,s,v nodes -> RGB;
                              // 3: raw float64 node values to out_f64 (the create_cppn node call itself)
    int max_nodes;            // LDS column count
    uint8_t* out;             // [G][c_dim][N]
    double* out_f64;          // mode 3: [G][c_out][N]
};

// np.array(float64, dtype=np.uint8) on x86-64: truncate toward zero to int32, keep the low byte;
// NaN and |v| >= 2^31 give 0 (cvttsd2si "integer indefinite" 0x80000000).  Pinned by tests/golden/postprocess.npz.
__device__ __forceinline__ uint8_t quant_u8(double v)
{
    const double t = trunc(v);
    if (!(fabs(t) < 2147483648.0)) return 0;
    return (uint8_t)((int)t & 0xFF);
}

__device__ __forceinline__ double cppn_act(int act, double x)
{
    switch (act) {
        case 0: return det_sigmoid64(5.0 * x);  // sigmoid_activation: sigmoid(5x)
        case 1: return det_tanh64(2.5 * x);
        case 2: return fabs(x);
        case 3: return det_exp64(-5.0 * (x * x));
        case 4: return x;
        case 5: return det_sin64(x);
        default: return (x > 0.0 || x != x) ? x : 0.0;  // relu; NaN propagates as in torch.relu / np.maximum
    }
}

__global__ void __launch_bounds__(CPPN_THREADS) cppn_render_kernel(const CppnArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int g = blockIdx.y;
    const int n0 = a.node_off[g], n1 = a.node_off[g + 1];
    const int nn = n1 - n0;
    const int e0 = a.edge_off[n0], e1 = a.edge_off[n1];
    const int ne = e1 - e0;
    // LDS carve: values [max_nodes][T] f64 | bias[nn] f64 | resp[nn] f64 | ew[ne] f64 | eoff[nn+1] i32 | esrc[ne] i32 | act[nn] u8
    double* vals = reinterpret_cast<double*>(smem);
    double* s_bias = vals + (size_t)a.max_nodes * CPPN_THREADS;
    double* s_resp = s_bias + nn;
    double* s_ew = s_resp + nn;
    int32_t* s_eoff = reinterpret_cast<int32_t*>(s_ew + ne);
    int32_t* s_esrc = s_eoff + nn + 1;
    uint8_t* s_act = reinterpret_cast<uint8_t*>(s_esrc + ne);
    const int tid = threadIdx.x;
    for (int i = tid; i < nn; i += CPPN_THREADS) {
        s_bias[i] = a.node_bias[n0 + i];
        s_resp[i] = a.node_resp[n0 + i];
        s_act[i] = a.node_act[n0 + i];
    }
    for (int i = tid; i <= nn; i += CPPN_THREADS) s_eoff[i] = a.edge_off[n0 + i] - e0;
    for (int i = tid; i < ne; i += CPPN_THREADS) {
        s_ew[i] = a.edge_w[e0 + i];
        s_esrc[i] = a.edge_src[e0 + i];
    }
    __syncthreads();

    const int p = blockIdx.x * CPPN_THREADS + tid;
    if (p >= a.N) return;
    double leaf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) leaf[i] = (i < a.n_planes) ? a.planes[(size_t)i * a.N + p] : 0.0;
    const bool is_bg = (leaf[0] == -1.0);

    for (int n = 0; n < nn; ++n) {
        const int b = s_eoff[n], e = s_eoff[n + 1];
        double sum = 0.0;
        for (int k = b; k < e; ++k) {
            const int src = s_esrc[k];
            double x;
            if (src >= 0) x = vals[(size_t)src * CPPN_THREADS + tid];
            else {
                const int li = -src - 1;
                x = (li >= a.n_planes) ? 1.0 : (li == 0 ? leaf[0] : li == 1 ? leaf[1] : li == 2 ? leaf[2] : leaf[3]);
            }
            const double term = s_ew[k] * x;
            sum = (k == b) ? term : sum + term;   // Python sum(): 0 + t0 == t0
        }
        const double pre = s_resp[n] * sum;
        vals[(size_t)n * CPPN_THREADS + tid] = cppn_act(s_act[n], pre + s_bias[n]);
    }

    if (a.mode == 3) {  // node_func(x=inp_x, y=inp_y) of generate_illusion.py:395,406,443: the float64 plane itself
        for (int c = 0; c < a.c_out; ++c)
            a.out_f64[((size_t)g * a.c_out + c) * a.N + p] = vals[(size_t)a.out_node[g * a.c_out + c] * CPPN_THREADS + tid];
        return;
    }
    uint8_t* out = a.out + (size_t)g * a.c_dim * a.N + p;
    if (a.mode == 2) {  // colour, gradient == 0: palette from node 0 (generate_illusion.py:405-431)
        const double v = vals[(size_t)a.out_node[g * a.c_out] * CPPN_THREADS + tid];
        const uint8_t code = quant_u8(v * 4.0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint8_t o = (code == 0 || code == c + 1) ? 255 : 0;
            if (is_bg) o = (uint8_t)(a.bg * 255);
            out[(size_t)c * a.N] = o;
        }
        return;
    }
    if (a.mode == 4) {  // get_equilum_image_from_cppn (generate_illusion.py:333-367): the three nodes are h, s, v; the reference hands
        // the whole array to colorsys.hsv_to_rgb (TypeError) -- here the conversion it names runs per pixel, operation by
        // operation as CPython's colorsys does, then the usual uint8(v * 255).  bg fills h, s AND v, as the reference does.
        double hsv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = vals[(size_t)a.out_node[g * a.c_out + c] * CPPN_THREADS + tid];
            if (is_bg) v = (double)a.bg;
            hsv[c] = v;
        }
        const double h = hsv[0], sat = hsv[1], v = hsv[2];
        double r, gg, b;
        if (sat == 0.0) { r = gg = b = v; }
        else {
            const double h6 = h * 6.0;
            double i = trunc(h6);                        // int(): toward zero
            if (!(fabs(i) < 9007199254740992.0)) i = 0.0;  // int() raises for NaN / inf, and f = 0 beyond 2^53 anyway: build-defined, sextant 0
            const double f = h6 - i;
            const double pp = v * (1.0 - sat);
            const double qq = v * (1.0 - sat * f);
            const double tt = v * (1.0 - sat * (1.0 - f));
            double im = fmod(i, 6.0);                    // exact; Python's % is non-negative
            if (im < 0.0) im += 6.0;
            if (im == 0.0) { r = v; gg = tt; b = pp; }
            else if (im == 1.0) { r = qq; gg = v; b = pp; }
            else if (im == 2.0) { r = pp; gg = v; b = tt; }
            else if (im == 3.0) { r = pp; gg = qq; b = v; }
            else if (im == 4.0) { r = tt; gg = pp; b = v; }
            else { r = v; gg = pp; b = qq; }
        }
        out[0] = quant_u8(r * 255.0);
        out[(size_t)a.N] = quant_u8(gg * 255.0);
        out[(size_t)2 * a.N] = quant_u8(b * 255.0);
        return;
    }
    for (int c = 0; c < a.c_dim; ++c) {
        double v = vals[(size_t)a.out_node[g * a.c_out + c] * CPPN_THREADS + tid];
        if (is_bg) v = (double)a.bg;
        if (a.mode == 1) v = rint(v);  // np.round: half to even
        out[(size_t)c * a.N] = quant_u8(v * 255.0);
    }
}

}  // namespace eig
