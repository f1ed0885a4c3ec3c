#!/usr/bin/env python3
"""Measurement script (not product, never the headline): what a split-bf16 PredNet would do to the results.

VERDICT r1, item 8: the fp32 MFMA pipe runs the roll-out at 0.88 of its peak, so more speed needs cheaper multiply-adds; the
bf16 pipe is 16x faster per MFMA.  Before anyone writes that kernel this script answers the parity half of the question on
the GPU box: PredNet convolutions with both operands split into bf16 terms (x = hi + mid + lo, every term exactly
representable in bf16, products of two terms exact in fp32, fp32 accumulation -- what v_mfma_f32_*_bf16 computes),
  1 term  : hi*hi                                   (plain bf16 inputs)
  3 terms : hi*hi + hi*mid + mid*hi                 (~2^-16 relative per product)
  6 terms : + mid*mid + hi*lo + lo*hi               (~2^-24: fp32-grade products, different rounding points)
against the engine's fp32 frames and fitness on the same genomes, next to a CONTROL that runs the same torch code with
unsplit fp32 operands (the deviation a mere change of summation order causes).

Everything but the emulated roll-out runs on the product path: renders, reference frames, Lucas-Kanade and scores come from
the HIP engine; the emulation is torch on the same GPU (im2col + matmul, so that no library picks a Winograd kernel).

    python scripts/split_bf16_study.py [--genomes 64] [--w 256 --h 256] [--out profiles/r02_split_bf16_study.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.nn.functional as F

from evolutionary_illusion_generator_amd import genome as genome_mod, grids, synth, weights
from evolutionary_illusion_generator_amd.engine import Engine

GATES = ("i", "f", "c", "o")


def split_terms(x, n):
    """x (fp32) -> list of n fp32 tensors, each exactly representable in bf16, summing to x up to 2^-(8n)."""
    out, r = [], x
    for _ in range(n):
        t = r.to(torch.bfloat16).to(torch.float32)
        out.append(t)
        r = r - t
    return out


PAIRS = {1: [(0, 0)], 3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]}


def conv3x3(x, w_terms, n_terms):
    """3x3 'same' convolution as im2col + matmul.  n_terms = 0: plain fp32 operands (control)."""
    B, C, H, W = x.shape
    cols = F.unfold(x, 3, padding=1)                      # [B, C*9, H*W]
    if n_terms == 0:
        y = torch.matmul(w_terms[0][0], cols)
    else:
        xt = split_terms(cols, max(i for i, _ in PAIRS[n_terms]) + 1)
        y = None
        for i, j in PAIRS[n_terms]:                        # the small terms first would be more accurate; an MFMA chain adds in issue order
            t = torch.matmul(w_terms[n_terms][j], xt[i])
            y = t if y is None else y + t
    return y.view(B, -1, H, W)


class PredNetEmu:
    def __init__(self, wts, channels, w, h, dev):
        self.ch, self.w, self.h, self.L, self.dev = list(channels), w, h, len(channels), dev
        p = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in wts.items()}
        self.p = p

        def prep(wt):  # OIHW -> [O, I*9] in every split
            m = wt.reshape(wt.shape[0], -1).contiguous()
            return {0: [m], 1: split_terms(m, 1), 3: split_terms(m, 2), 6: split_terms(m, 3)}

        self.convA = {l: prep(p["ConvA%d/W" % l]) for l in range(1, self.L)}
        self.convP = {l: prep(p["ConvP%d/W" % l]) for l in range(self.L)}
        self.lstm = []
        for l in range(self.L):
            srcs = ["x_%s0", "x_%s1", "h_%s"] if l < self.L - 1 else ["x_%s0", "h_%s"]
            ws = [prep(torch.cat([p["ConvLSTM%d/%s/W" % (l, s % g)] for g in GATES], 0)) for s in srcs]
            b = torch.cat([p["ConvLSTM%d/h_%s/b" % (l, g)] for g in GATES], 0)
            self.lstm.append((ws, b))

    def reset(self, B):
        z = lambda l: torch.zeros(B, self.ch[l], self.h >> l, self.w >> l, device=self.dev)
        self.hs = [z(l) for l in range(self.L)]
        self.cs = [z(l) for l in range(self.L)]
        self.P = [z(l) for l in range(self.L)]

    @torch.no_grad()
    def step(self, x, nt):
        L, p = self.L, self.p
        E = [None] * L
        E[0] = torch.cat((F.relu(x - self.P[0]), F.relu(self.P[0] - x)), 1)
        for l in range(1, L):
            A = F.max_pool2d(F.relu(conv3x3(E[l - 1], self.convA[l], nt) + p["ConvA%d/b" % l].view(1, -1, 1, 1)), 2, 2)
            E[l] = torch.cat((F.relu(A - self.P[l]), F.relu(self.P[l] - A)), 1)
        for l in reversed(range(L)):
            ws, b = self.lstm[l]
            srcs = [E[l]] + ([F.interpolate(self.hs[l + 1], scale_factor=2, mode="nearest")] if l < L - 1 else []) + [self.hs[l]]
            z = sum(conv3x3(s_, w_, nt) for s_, w_ in zip(srcs, ws)) + b.view(1, -1, 1, 1)
            zi, zf, zc, zo = torch.chunk(z, 4, 1)
            c = self.cs[l]
            i = torch.sigmoid(zi + p["ConvLSTM%d/c_i/W" % l] * c)
            f = torch.sigmoid(zf + p["ConvLSTM%d/c_f/W" % l] * c)
            o = torch.sigmoid(zo + p["ConvLSTM%d/c_o/W" % l] * c)
            cn = torch.tanh(zc) * i + f * c
            self.cs[l] = cn
            self.hs[l] = o * torch.tanh(cn)
            v = conv3x3(self.hs[l], self.convP[l], nt) + p["ConvP%d/b" % l].view(1, -1, 1, 1)
            self.P[l] = v.clamp(0.0, 1.0) if l == 0 else F.relu(v)
        return self.P[0]

    @torch.no_grad()
    def frames_19_20(self, imgs_u8, nt):
        """uint8 [B,C,H,W] (device) -> the two frames Lucas-Kanade reads on the population path, uint8 [B,2,C,H,W]."""
        self.reset(imgs_u8.shape[0])
        x = imgs_u8.float() / 255.0
        out = []
        for t in range(21):
            if t >= 20:
                x = self.P[0]
            p0 = self.step(x, nt)
            if t >= 19:
                out.append((p0 * 255.0).to(torch.uint8))
        return torch.stack(out, 1).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=64)
    ap.add_argument("--w", type=int, default=256)
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=4)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda", 0)
    W, H, CH, ST = a.w, a.h, [3, 48, 96, 192], 1
    cfg = synth.make_config(2, 3)
    pop = synth.make_population(a.genomes, cfg, seed=0)
    genomes = [g for _, g in pop]
    wts = weights.synthetic_prednet_weights(CH, W, H, seed=0)
    grid = grids.create_grid(ST, W, H, 10)
    eng = Engine(W, H, CH, a.genomes)
    eng.set_weights(wts)
    eng.set_grid([grid["x_mat"], grid["y_mat"]])
    gb = genome_mod.GenomeBatch(genomes, cfg, 3)
    fit_ref = eng.eval_population(gb, ST)
    d_img = torch.zeros((a.genomes, 3, H, W), dtype=torch.uint8, device=dev)
    eng.render_cppn(gb, d_img)
    fr_ref = torch.zeros((a.genomes, 2, 3, H, W), dtype=torch.uint8, device=dev)
    eng.prednet_rollout(d_img, a.genomes, 21, 19, fr_ref)
    torch.cuda.synchronize()
    emu = PredNetEmu(wts, CH, W, H, dev)
    vec = torch.zeros((a.genomes, eng.K, 4), dtype=torch.float32, device=dev)
    cnt = torch.zeros(a.genomes, dtype=torch.int32, device=dev)
    fit = torch.zeros(a.genomes, dtype=torch.float64, device=dev)
    report = {"shape": [W, H], "channels": CH, "genomes": a.genomes, "nonzero_reference_fitness": int((fit_ref != 0).sum()), "variants": {}}
    for name, nt in (("fp32 operands, torch order (control)", 0), ("bf16 x1", 1), ("bf16 x3", 3), ("bf16 x6", 6)):
        t0 = time.time()
        fr = torch.cat([emu.frames_19_20(d_img[i:i + a.chunk], nt) for i in range(0, a.genomes, a.chunk)])
        torch.cuda.synchronize()
        flips = int((fr != fr_ref).sum())
        maxdiff = int((fr.int() - fr_ref.int()).abs().max())
        f0, f1 = fr[:, 0].contiguous(), fr[:, 1].contiguous()
        eng.flow(f0, 3 * H * W, f1, 3 * H * W, a.genomes, vec, cnt)
        eng.score(ST, vec, cnt, a.genomes, fit)
        torch.cuda.synchronize()
        got = fit.cpu().numpy()
        both = (fit_ref != 0) & (got != 0)
        rel = np.abs(got[both] - fit_ref[both]) / np.abs(fit_ref[both])
        report["variants"][name] = {
            "frame_flip_rate": flips / float(fr.numel()), "max_abs_byte_difference": maxdiff,
            "genomes_zero_on_one_side_only": int(((fit_ref != 0) != (got != 0)).sum()),
            "fitness_rel_dev_max": float(rel.max()) if len(rel) else None,
            "fitness_rel_dev_median": float(np.median(rel)) if len(rel) else None,
            "genomes_above_1e-4": int((rel > 1e-4).sum()), "genomes_compared": int(both.sum()), "seconds": round(time.time() - t0, 1)}
        print(name, json.dumps(report["variants"][name]), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
