// det_math64.h -- float64 exp / tanh / sigmoid / sin for the CPPN render, canonical arithmetic.
//
// The reference evaluates CPPN activations in torch float64 (pytorch_neat/activations.py reached from
// /root/reference/generate_illusion.py:395).  libms differ in the last ulp and a saturating output (tanh = 1 - 1e-16
// vs exactly 1) quantises to 254 vs 255 along a whole contour, so device and oracle both use ONE published algorithm
// per function, spelled with IEEE + - * / floor rint ldexp only (the translation unit is built with
// -ffp-contract=off; no fma, no ocml transcendental): fdlibm exp, Cephes tanh, Cephes sin.
// oracle/detmath64.py is the same sequence of operations in numpy.
#pragma once
#include <hip/hip_runtime.h>

namespace eig {

__device__ __forceinline__ double det_exp64(double x)
{
    const bool nan = (x != x);
    double xc = fmin(fmax(x, -746.0), 710.0);
    if (nan) xc = 0.0;
    const double k = rint(xc * 1.44269504088896338700e+00);
    const double hi = xc - k * 6.93147180369123816490e-01;
    const double lo = k * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (1.66666666666666019037e-01 + t * (-2.77777777770155933842e-03 + t * (6.61375632143793436117e-05 +
                     t * (-1.65339022054652515390e-06 + t * 4.13813679705723846039e-08))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const double out = ldexp(y, (int)k);
    return nan ? x : out;
}

__device__ __forceinline__ double det_tanh64(double x)
{
    if (x != x) return x;
    const double ax = fabs(x);
    if (ax < 0.625) {
        const double s = x * x;
        const double num = (-9.64399179425052238628e-1 * s + -9.92877231001918586564e1) * s + -1.61468768441708447952e3;
        const double den = ((s + 1.12811678491632931402e2) * s + 2.23548839060100448583e3) * s + 4.84406305325125486048e3;
        return x + x * s * (num / den);
    }
    const double e = det_exp64(2.0 * fmin(ax, 40.0));
    const double big = 1.0 - 2.0 / (e + 1.0);
    return x < 0 ? -big : big;
}

__device__ __forceinline__ double det_sigmoid64(double z) { return 1.0 / (1.0 + det_exp64(-z)); }

__device__ __forceinline__ double polevl6(double z, const double* c)
{
    double r = c[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) r = r * z + c[i];
    return r;
}

__device__ __forceinline__ double det_sin64(double x)
{
    const double SINCOF[6] = {1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
                              -1.98412698295895385996e-4, 8.33333333332211858878e-3, -1.66666666666666307295e-1};
    const double COSCOF[6] = {-1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
                              2.48015872888517045348e-5, -1.38888888888730564116e-3, 4.16666666666665929218e-2};
    if (!(fabs(x) <= 1.79769313486231570815e308)) return x - x;  // NaN / inf -> NaN
    double ax = fabs(x);
    if (ax > 1.073741824e9) return 0.0;
    double y = floor(ax * 1.27323954473516268615);
    double z = ldexp(y, -4);
    z = floor(z);
    z = y - ldexp(z, 4);
    int j = (int)z;
    if (j & 1) { j += 1; y += 1.0; }
    j &= 7;
    bool flip = false;
    if (j > 3) { flip = true; j -= 4; }
    const double zr = ((ax - y * 7.85398125648498535156e-1) - y * 3.77489470793079817668e-8) - y * 2.69515142907905952645e-15;
    const double zz = zr * zr;
    double r;
    if (j == 1 || j == 2) r = 1.0 - ldexp(zz, -1) + zz * zz * polevl6(zz, COSCOF);
    else r = zr + zr * zz * polevl6(zz, SINCOF);
    const bool neg = (x < 0) != flip;
    return neg ? -r : r;
}

}  // namespace eig
