"""ORACLE (test infrastructure): the float64 transcendental kernels of the CPPN render, restated in numpy.

The reference evaluates CPPN activations with torch float64 (sin / exp / tanh / sigmoid from torch's vectorised
libm, /root/reference/generate_illusion.py:395 via pytorch_neat/activations.py).  Different libms disagree in the
last ulp, and a CPPN output that saturates (tanh -> 1 - 1e-16 vs exactly 1.0) then quantises to 254 vs 255 on a
whole contour band.  To make the GPU render byte-exact against this oracle, both use ONE published algorithm per
function, written with IEEE +, -, *, /, floor, rint, ldexp only (no fused multiply-add, no library call):
  exp  -- fdlibm __ieee754_exp (Cody-Waite reduction by ln2 hi/lo, degree-5 rational correction)
  tanh -- Cephes tanh (rational P/Q for |x| < 0.625, 1 - 2/(exp(2|x|)+1) above)
  sin  -- Cephes sin (octant reduction with the 3-part pi/4, degree-6 sine / cosine polynomials)
Each is within ~1-2 ulp of the true function for the argument range a CPPN produces, i.e. in the same accuracy
class as the reference's libm.  csrc/det_math64.h spells the same operations for the device.
"""
import numpy as np

LN2_HI = 6.93147180369123816490e-01
LN2_LO = 1.90821492927058770002e-10
INV_LN2 = 1.44269504088896338700e+00
P1, P2, P3, P4, P5 = (1.66666666666666019037e-01, -2.77777777770155933842e-03, 6.61375632143793436117e-05,
                      -1.65339022054652515390e-06, 4.13813679705723846039e-08)


def det_exp(x):
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(all="ignore"):
        nan = np.isnan(x)
        xc = np.where(nan, 0.0, np.minimum(np.maximum(x, -746.0), 710.0))
        k = np.rint(xc * INV_LN2)
        hi = xc - k * LN2_HI
        lo = k * LN2_LO
        r = hi - lo
        t = r * r
        c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))))
        y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi)
        out = np.ldexp(y, k.astype(np.int32))
        return np.where(nan, np.nan, out)


TP = (-9.64399179425052238628e-1, -9.92877231001918586564e1, -1.61468768441708447952e3)
TQ = (1.12811678491632931402e2, 2.23548839060100448583e3, 4.84406305325125486048e3)


def det_tanh(x):
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(all="ignore"):
        ax = np.abs(x)
        s = x * x
        num = (TP[0] * s + TP[1]) * s + TP[2]
        den = ((s + TQ[0]) * s + TQ[1]) * s + TQ[2]
        small = x + x * s * (num / den)
        e = det_exp(2.0 * np.minimum(ax, 40.0))
        big = 1.0 - 2.0 / (e + 1.0)
        big = np.where(x < 0, -big, big)
        out = np.where(ax < 0.625, small, big)
        return np.where(np.isnan(x), np.nan, out)


def det_sigmoid(z):
    """1 / (1 + exp(-z))"""
    with np.errstate(all="ignore"):
        return 1.0 / (1.0 + det_exp(-np.asarray(z, dtype=np.float64)))


DP1, DP2, DP3 = 7.85398125648498535156e-1, 3.77489470793079817668e-8, 2.69515142907905952645e-15
FOPI = 1.27323954473516268615
SINCOF = (1.58962301576546568060e-10, -2.50507477628578072866e-8, 2.75573136213857245213e-6,
          -1.98412698295895385996e-4, 8.33333333332211858878e-3, -1.66666666666666307295e-1)
COSCOF = (-1.13585365213876817300e-11, 2.08757008419747316778e-9, -2.75573141792967388112e-7,
          2.48015872888517045348e-5, -1.38888888888730564116e-3, 4.16666666666665929218e-2)


def _polevl(z, c):
    r = c[0]
    for k in c[1:]:
        r = r * z + k
    return r


def det_sin(x):
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(all="ignore"):
        bad = ~np.isfinite(x)
        ax = np.where(bad, 0.0, np.abs(x))
        huge = ax > 1.073741824e9          # Cephes: total loss of precision -> 0
        ax = np.where(huge, 0.0, ax)
        y = np.floor(ax * FOPI)
        z = np.ldexp(y, -4)
        z = np.floor(z)
        z = y - np.ldexp(z, 4)             # y mod 16
        j = z.astype(np.int32)
        odd = (j & 1) == 1
        j = np.where(odd, j + 1, j)
        y = np.where(odd, y + 1.0, y)
        j = j & 7
        flip = j > 3
        j = np.where(flip, j - 4, j)
        zr = ((ax - y * DP1) - y * DP2) - y * DP3
        zz = zr * zr
        cosv = 1.0 - np.ldexp(zz, -1) + zz * zz * _polevl(zz, COSCOF)
        sinv = zr + zr * zz * _polevl(zz, SINCOF)
        r = np.where((j == 1) | (j == 2), cosv, sinv)
        neg = (x < 0) ^ flip
        r = np.where(neg, -r, r)
        r = np.where(huge, 0.0, r)
        return np.where(bad, np.nan, r)
