"""ORACLE (test infrastructure): independent torch-CPU fp32 restatement of the PredNet roll-out.

Written against the same published algorithm as oracle/eig_oracle.c (chainer_prednet PredNet/net.py,
call_prednet.py; PARITY UNPINNED, see oracle/__init__.py) but with library convolutions
(torch.nn.functional.conv2d -> oneDNN) and library sigmoid/tanh, i.e. a DIFFERENT fp32 evaluation order.
It serves two purposes: (1) cross-check that the C restatement's semantics are right (tests compare the
float predictions to ~1e-4), (2) the fastest CPU path available here, timed as bench.py's cpu_baseline.
"""
import numpy as np
import torch
import torch.nn.functional as F

GATES = ("i", "f", "c", "o")


def _conv_matmul(x, w, b, padding=1):
    """3x3 'same' convolution as im2col + ONE fp32 matmul: no library gets to pick a Winograd / FFT algorithm (used on a GPU
    device, where MIOpen would choose freely)."""
    B, C, H, W = x.shape
    y = torch.matmul(w.reshape(w.shape[0], -1), F.unfold(x, 3, padding=padding)).view(B, -1, H, W)
    return y if b is None else y + b.view(1, -1, 1, 1)


_WINO = {}


# Winograd F(m x m, 3 x 3) transform matrices (Lavin & Gray 2015 / the wincnn construction; interpolation points 0, +-1, [2 | +-2], inf): (B^T, G, A^T)
_WINO_MATS = {
    2: ([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
        [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]],
        [[1, 1, 1, 0], [0, 1, -1, -1]]),
    3: ([[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]],
        [[.5, 0, 0], [-.5, -.5, -.5], [-1 / 6., 1 / 6., -1 / 6.], [1 / 6., 1 / 3., 2 / 3.], [0, 0, 1]],
        [[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]]),
    4: ([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
        [[.25, 0, 0], [-1 / 6., -1 / 6., -1 / 6.], [-1 / 6., 1 / 6., -1 / 6.], [1 / 24., 1 / 12., 1 / 6.], [1 / 24., -1 / 12., 1 / 6.], [0, 0, 1]],
        [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]),
}


def _conv_winograd(x, w, b, padding=1, m=2):
    """3x3 'same' convolution as Winograd F(m x m, 3 x 3), m = 2, 3 or 4, in fp32 throughout (a STUDY of what
    2.25x / 3.24x / 4x fewer multiply-adds would do to the results; nothing in the product uses it).  Fixed transform order: V = B^T d B per
    (m + 2)^2 input tile (rows first, then columns), U = G g G^T per filter, M = sum_c U * V as one fp32 matmul per tile position,
    Y = A^T M A (rows first).  Maps whose size is not a multiple of m are zero-padded at the bottom / right and cropped."""
    B, C, H, W = x.shape
    assert padding == 1
    key = (x.device, m)
    if key not in _WINO:
        _WINO[key] = tuple(torch.tensor(t, dtype=torch.float32, device=x.device) for t in _WINO_MATS[m])
    Bt, G, At = _WINO[key]
    a = m + 2
    O = w.shape[0]
    U = torch.matmul(torch.matmul(G, w), G.t())                               # [O, C, a, a]
    Hp, Wp = -(-H // m) * m, -(-W // m) * m
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    d = xp.unfold(2, a, m).unfold(3, a, m)                                    # [B, C, Hp/m, Wp/m, a, a]
    V = torch.matmul(torch.matmul(Bt, d), Bt.t())                             # [B, C, th, tw, a, a]
    th, tw = Hp // m, Wp // m
    Vm = V.permute(4, 5, 1, 0, 2, 3).reshape(a * a, C, B * th * tw)           # [a^2, C, tiles]
    Um = U.permute(2, 3, 0, 1).reshape(a * a, O, C)                           # [a^2, O, C]
    M = torch.bmm(Um, Vm).view(a, a, O, B, th, tw).permute(3, 2, 4, 5, 0, 1)  # [B, O, th, tw, a, a]
    Y = torch.matmul(torch.matmul(At, M), At.t())                             # [B, O, th, tw, m, m]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, O, Hp, Wp)[:, :, :H, :W]
    return y if b is None else y + b.view(1, -1, 1, 1)


def _conv_winograd3(x, w, b, padding=1):
    return _conv_winograd(x, w, b, padding, m=3)


def _conv_winograd4(x, w, b, padding=1):
    return _conv_winograd(x, w, b, padding, m=4)


CONVS = {"library": F.conv2d, "matmul": _conv_matmul, "winograd": _conv_winograd, "winograd3": _conv_winograd3, "winograd4": _conv_winograd4}


class PredNetTorch:
    def __init__(self, weights, channels, w, h, device="cpu", conv="library", order="stacked", conv_lstm=None, wino_min_layer=0):
        """device: "cpu" (oneDNN) or a cuda device (rocBLAS): two more fp32 summation orders, both independent of the build's
        canonical chain.  conv: "library" = F.conv2d, "matmul" = im2col + matmul (what chainer's CPU Convolution2D does:
        im2col + tensordot -> BLAS sgemm), "winograd" = F(2x2, 3x3) in fp32 (a study, tests/studies/winograd_study.py).
        order: "stacked" = the sources' convolutions summed, bias, then library sigmoid / tanh (round 1-2 cross-check);
               "chainer" = the element-wise order of the reference's own ConvLSTM, as far as it is knowable without its BLAS
               (see _lstm_chainer)."""
        if order not in ("stacked", "chainer"):
            raise ValueError("order must be 'stacked' or 'chainer'")
        self.order = order
        self.ch, self.w, self.h, self.L = list(channels), w, h, len(channels)
        self.dev = torch.device(device)
        # (studies: conv_lstm = another convolution for the ConvLSTM sources only; wino_min_layer = l: the Winograd variants apply to layers >= l, the
        # layers below run as "matmul" -- the HIP path's own choice is l = 1, the image layer stays direct)
        self.wino_min_layer = wino_min_layer
        self._conv_plain = CONVS[conv]
        self._conv_lstm = CONVS[conv_lstm or conv]
        self.p = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(self.dev) for k, v in weights.items()}
        # one conv per (layer, source): the 4 gates stacked along the output channels
        self.lstm = []
        for l in range(self.L):
            srcs = ["x_%s0", "x_%s1", "h_%s"] if l < self.L - 1 else ["x_%s0", "h_%s"]
            ws = [torch.cat([self.p["ConvLSTM%d/%s/W" % (l, s % g)] for g in GATES], 0) for s in srcs]
            b = torch.cat([self.p["ConvLSTM%d/h_%s/b" % (l, g)] for g in GATES], 0)
            self.lstm.append((ws, b))
        self.reset(1)

    def _cp(self, l):
        """convolution of a ConvA / ConvP running at the resolution of layer l"""
        return self._conv_plain if l >= self.wino_min_layer else (_conv_matmul if self._conv_plain not in (F.conv2d, _conv_matmul) else self._conv_plain)

    def _cl(self, l):
        """convolution of the ConvLSTM sources of layer l"""
        return self._conv_lstm if l >= self.wino_min_layer else (_conv_matmul if self._conv_lstm not in (F.conv2d, _conv_matmul) else self._conv_lstm)

    def _lstm_chainer(self, l, srcs):
        """One ConvLSTM step in the order chainer_prednet's PredNet/net.py ConvLSTM.__call__ evaluates it (quadjr/PredNet
        lineage, UPSTREAM-RECALL: SURVEY.md B.2 -- the submodule is absent from /root/reference, .gitmodules:1-3):

            ii = self.x_i0(x[0]); ii += self.x_i1(x[1]); ii += self.h_i(self.h); ii += self.c_i(self.c); ii = F.sigmoid(ii)
            ff = ... the same with x_f*, h_f, c_f ...
            cc = self.x_c0(x[0]); cc += self.x_c1(x[1]); cc += self.h_c(self.h); cc = F.tanh(cc); cc *= ii; cc += (ff * self.c)
            oo = ... x_o*, h_o, c_o(self.c) -- the OLD c ...; oo = F.sigmoid(oo)
            self.c = cc; self.h = oo * F.tanh(self.c)

        i.e. every convolution is a tensor of its own (x_* without bias, h_* = Convolution2D WITH bias, added to its own
        output), the tensors are added left to right with one fp32 rounding each, the peephole EltFilter `c_g(c) = W * c`
        is a rounded product added last, the cell update is two rounded products and one addition (no fma), and chainer's
        CPU F.sigmoid is `tanh(x * 0.5) * 0.5 + 0.5` (chainer/functions/activation/sigmoid.py, forward_cpu).  The four
        gates' convolutions of one source are stacked along the output channels here: each output channel is still its
        own dot product, so the element-wise order per gate is exactly the one above.  What stays unknowable is the
        summation order INSIDE a convolution (chainer: im2col + the host's BLAS)."""
        p = self.p
        ws, b = self.lstm[l]
        names = (["x0", "x1", "h"] if len(srcs) == 3 else ["x0", "h"])
        z = None
        for s, w_, nm in zip(srcs, ws, names):
            y = self._cl(l)(s, w_, b if nm == "h" else None, padding=1)  # h_*: the bias is added to that convolution's output
            z = y if z is None else z + y
        zi, zf, zc, zo = torch.chunk(z, 4, 1)
        c = self.cs[l]
        sig = lambda x: torch.tanh(x * 0.5) * 0.5 + 0.5
        ii = sig(zi + p["ConvLSTM%d/c_i/W" % l] * c)
        ff = sig(zf + p["ConvLSTM%d/c_f/W" % l] * c)
        cc = torch.tanh(zc)
        cc = cc * ii
        cc = cc + ff * c
        oo = sig(zo + p["ConvLSTM%d/c_o/W" % l] * c)
        self.cs[l] = cc
        self.hs[l] = oo * torch.tanh(cc)

    def reset(self, B):
        z = lambda l, m=1: torch.zeros(B, m * self.ch[l], self.h >> l, self.w >> l, device=self.dev)
        self.hs = [z(l) for l in range(self.L)]
        self.cs = [z(l) for l in range(self.L)]
        self.P = [z(l) for l in range(self.L)]

    @torch.no_grad()
    def step(self, x):
        L, p = self.L, self.p
        E = [None] * L
        E[0] = torch.cat((F.relu(x - self.P[0]), F.relu(self.P[0] - x)), 1)
        for l in range(1, L):
            A = F.max_pool2d(F.relu(self._cp(l - 1)(E[l - 1], p["ConvA%d/W" % l], p["ConvA%d/b" % l], padding=1)), 2, 2)   # (runs at the resolution of layer l - 1)
            E[l] = torch.cat((F.relu(A - self.P[l]), F.relu(self.P[l] - A)), 1)
        for l in reversed(range(L)):
            ws, b = self.lstm[l]
            srcs = [E[l]] + ([F.interpolate(self.hs[l + 1], scale_factor=2, mode="nearest")] if l < L - 1 else []) + [self.hs[l]]
            if self.order == "chainer":
                self._lstm_chainer(l, srcs)
                v = self._cp(l)(self.hs[l], p["ConvP%d/W" % l], p["ConvP%d/b" % l], padding=1)
                self.P[l] = v.clamp(0.0, 1.0) if l == 0 else F.relu(v)
                continue
            z = sum(self._cl(l)(s, w_, None, padding=1) for s, w_ in zip(srcs, ws)) + b.view(1, -1, 1, 1)
            zi, zf, zc, zo = torch.chunk(z, 4, 1)
            c = self.cs[l]
            i = torch.sigmoid(zi + p["ConvLSTM%d/c_i/W" % l] * c)
            f = torch.sigmoid(zf + p["ConvLSTM%d/c_f/W" % l] * c)
            o = torch.sigmoid(zo + p["ConvLSTM%d/c_o/W" % l] * c)
            cn = torch.tanh(zc) * i + f * c
            self.cs[l] = cn
            self.hs[l] = o * torch.tanh(cn)
            v = self._cp(l)(self.hs[l], p["ConvP%d/W" % l], p["ConvP%d/b" % l], padding=1)
            self.P[l] = v.clamp(0.0, 1.0) if l == 0 else F.relu(v)
        return self.P[0]

    @torch.no_grad()
    def rollout(self, imgs, n_repeat=20, n_ext=2, requant=False):
        """imgs: uint8 [B,C0,H,W] -> (uint8 frames [B,T,C0,H,W], float32 P0 [B,T,C0,H,W])."""
        imgs = np.asarray(imgs)
        B = imgs.shape[0]
        self.reset(B)
        x = torch.from_numpy(imgs.astype(np.float32)).to(self.dev) / 255.0
        fr, fl = [], []
        for t in range(n_repeat + n_ext):
            if t >= n_repeat:
                x = (self.P[0] * 255.0).to(torch.uint8).float() / 255.0 if requant else self.P[0]
            p0 = self.step(x)
            fl.append(p0.cpu().numpy().copy())
            fr.append((p0 * 255.0).to(torch.uint8).cpu().numpy())
        return np.stack(fr, 1), np.stack(fl, 1)
