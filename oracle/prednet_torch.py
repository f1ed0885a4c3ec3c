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


class PredNetTorch:
    def __init__(self, weights, channels, w, h, device="cpu", conv="library"):
        """device: "cpu" (oneDNN) or a cuda device (rocBLAS): two more fp32 summation orders, both independent of the build's
        canonical chain.  conv: "library" = F.conv2d, "matmul" = im2col + matmul."""
        self.ch, self.w, self.h, self.L = list(channels), w, h, len(channels)
        self.dev = torch.device(device)
        self.conv = F.conv2d if conv == "library" else _conv_matmul
        self.p = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(self.dev) for k, v in weights.items()}
        # one conv per (layer, source): the 4 gates stacked along the output channels
        self.lstm = []
        for l in range(self.L):
            srcs = ["x_%s0", "x_%s1", "h_%s"] if l < self.L - 1 else ["x_%s0", "h_%s"]
            ws = [torch.cat([self.p["ConvLSTM%d/%s/W" % (l, s % g)] for g in GATES], 0) for s in srcs]
            b = torch.cat([self.p["ConvLSTM%d/h_%s/b" % (l, g)] for g in GATES], 0)
            self.lstm.append((ws, b))
        self.reset(1)

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
            A = F.max_pool2d(F.relu(self.conv(E[l - 1], p["ConvA%d/W" % l], p["ConvA%d/b" % l], padding=1)), 2, 2)
            E[l] = torch.cat((F.relu(A - self.P[l]), F.relu(self.P[l] - A)), 1)
        for l in reversed(range(L)):
            ws, b = self.lstm[l]
            srcs = [E[l]] + ([F.interpolate(self.hs[l + 1], scale_factor=2, mode="nearest")] if l < L - 1 else []) + [self.hs[l]]
            z = sum(self.conv(s, w_, None, padding=1) for s, w_ in zip(srcs, ws)) + b.view(1, -1, 1, 1)
            zi, zf, zc, zo = torch.chunk(z, 4, 1)
            c = self.cs[l]
            i = torch.sigmoid(zi + p["ConvLSTM%d/c_i/W" % l] * c)
            f = torch.sigmoid(zf + p["ConvLSTM%d/c_f/W" % l] * c)
            o = torch.sigmoid(zo + p["ConvLSTM%d/c_o/W" % l] * c)
            cn = torch.tanh(zc) * i + f * c
            self.cs[l] = cn
            self.hs[l] = o * torch.tanh(cn)
            v = self.conv(self.hs[l], p["ConvP%d/W" % l], p["ConvP%d/b" % l], padding=1)
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
