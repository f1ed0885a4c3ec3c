"""PredNet weight tables: chainer-npz import and seeded synthetic weights.

Tensor names are the chainer ``serializers.save_npz`` keys of chainer_prednet's ``L.Classifier(PredNet)``
with the ``predictor/`` prefix stripped:
``ConvA{l}/W,b`` ``ConvP{l}/W,b`` ``ConvLSTM{l}/x_{g}{n}/W`` ``ConvLSTM{l}/h_{g}/W,b`` ``ConvLSTM{l}/c_{g}/W``
(gates g in i,f,c,o; peepholes only i,f,o with shape (1, C, H_l, W_l)); conv weights are OIHW float32,
cross-correlation.  The trained files (fpsi_500000_20v.model ...) are external downloads
(/root/reference/illusion_generation.ipynb:140-157); their peepholes fix the resolution to 160x120.
"""
import numpy as np

GATES = ("i", "f", "c", "o")


def tensor_names(n_layers):
    """Order of the tensor table handed to eigen_set_prednet_weights (include/eigen_engine.h)."""
    names = []
    for l in range(n_layers):
        if l > 0:
            names += ["ConvA%d/W" % l, "ConvA%d/b" % l]
        names += ["ConvP%d/W" % l, "ConvP%d/b" % l]
        for g in GATES:
            names.append("ConvLSTM%d/x_%s0/W" % (l, g))
            if l < n_layers - 1:
                names.append("ConvLSTM%d/x_%s1/W" % (l, g))
            names.append("ConvLSTM%d/h_%s/W" % (l, g))
            names.append("ConvLSTM%d/h_%s/b" % (l, g))
        for g in ("i", "f", "o"):
            names.append("ConvLSTM%d/c_%s/W" % (l, g))
    return names


def tensor_shapes(channels, w, h):
    L = len(channels)
    if w % (1 << (L - 1)) or h % (1 << (L - 1)):
        raise ValueError("PredNet needs W and H divisible by 2^(layers-1) (2x2 pooling per layer)")
    shapes = {}
    for l, C in enumerate(channels):
        hl, wl = h >> l, w >> l
        if l > 0:
            shapes["ConvA%d/W" % l] = (C, 2 * channels[l - 1], 3, 3)
            shapes["ConvA%d/b" % l] = (C,)
        shapes["ConvP%d/W" % l] = (C, C, 3, 3)
        shapes["ConvP%d/b" % l] = (C,)
        for g in GATES:
            shapes["ConvLSTM%d/x_%s0/W" % (l, g)] = (C, 2 * C, 3, 3)
            if l < L - 1:
                shapes["ConvLSTM%d/x_%s1/W" % (l, g)] = (C, channels[l + 1], 3, 3)
            shapes["ConvLSTM%d/h_%s/W" % (l, g)] = (C, C, 3, 3)
            shapes["ConvLSTM%d/h_%s/b" % (l, g)] = (C,)
        for g in ("i", "f", "o"):
            shapes["ConvLSTM%d/c_%s/W" % (l, g)] = (1, C, hl, wl)
    return shapes


def load_chainer_npz(path, channels, w, h):
    """Read a chainer npz model file into {name: float32 array}, checking every shape."""
    shapes = tensor_shapes(channels, w, h)
    out = {}
    with np.load(path) as z:
        keys = {k[len("predictor/"):] if k.startswith("predictor/") else k: k for k in z.files}
        for name, shp in shapes.items():
            if name not in keys:
                raise KeyError("weight file %s lacks tensor %r" % (path, name))
            a = np.asarray(z[keys[name]], dtype=np.float32)
            if a.shape != shp:
                raise ValueError("%s: tensor %r has shape %s, expected %s for %dx%d channels %s"
                                 % (path, name, a.shape, shp, w, h, list(channels)))
            out[name] = np.ascontiguousarray(a)
    return out


def synthetic_prednet_weights(channels, w, h, seed=0, gain=0.6, leak=0.25):
    """Seeded stand-in for trained weights (no network access for the real files).

    Random N(0, gain/sqrt(fan_in)) convolutions and N(0, 0.1) peepholes everywhere, plus a hand-built
    "error integrator" in layer 0 so that the prediction tracks the input frame the way a trained PredNet
    does on a static image: the cell of channel k accumulates tanh(leak-scaled (E+_k - E-_k)) with the
    forget/input/output gates biased open, and ConvP0 reads the cell back with a centre tap.  The random
    part leaves sub-pixel frame-to-frame changes, so Lucas-Kanade finds non-degenerate flow.
    """
    rng = np.random.default_rng(seed)
    shapes = tensor_shapes(channels, w, h)
    out = {}
    for name, shp in shapes.items():
        if name.endswith("/b"):
            a = np.zeros(shp)
        elif "/c_" in name:
            a = rng.normal(0.0, 0.1, shp)
        else:
            fan_in = shp[1] * 9
            a = rng.normal(0.0, gain / np.sqrt(fan_in), shp)
        out[name] = a.astype(np.float32)
    C0 = channels[0]
    small = 0.15
    for g in GATES:
        for n in ("x_%s0" % g, "x_%s1" % g, "h_%s" % g):
            key = "ConvLSTM0/%s/W" % n
            if key in out:
                out[key] *= small
    for g in ("i", "f", "o"):
        out["ConvLSTM0/c_%s/W" % g] *= small
    out["ConvLSTM0/h_i/b"][:] = 3.0
    out["ConvLSTM0/h_f/b"][:] = 4.0
    out["ConvLSTM0/h_o/b"][:] = 4.0
    for k in range(C0):
        out["ConvLSTM0/x_c0/W"][k, k, 1, 1] += leak * 4.0        # + relu(x - P)
        out["ConvLSTM0/x_c0/W"][k, C0 + k, 1, 1] -= leak * 4.0   # - relu(P - x)
    out["ConvP0/W"] *= small
    for k in range(C0):
        out["ConvP0/W"][k, k, 1, 1] += 1.35
    return out
