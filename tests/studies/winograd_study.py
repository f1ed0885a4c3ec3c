#!/usr/bin/env python3
"""Study, not product: is there an fp32 route to north_star's >= 200x?  At 0.88 of the fp32 MFMA peak only
FEWER multiply-adds help; Winograd F(2x2, 3x3) needs 2.25x fewer on every 3x3 convolution.  Before anyone writes that kernel this
answers the parity half on the GPU box: the reference's element-wise order (oracle/prednet_torch.py order="chainer") with
  A  im2col + rocBLAS fp32 matmul convolutions                     (the reference-order implementation bench.py classifies against)
  B  the MIOpen library convolution                                 (control: another summation order of the same convolution)
  W  Winograd F(2x2, 3x3), fp32 throughout, fixed transform order  (oracle/prednet_torch.py: _conv_winograd)
and the HIP engine's canonical frames, all on the same genomes of the headline population, classified pair by pair with
oracle/classify.py (byte flip rate of the two frames Lucas-Kanade reads, genomes outside 1e-4, explained or not).

Decision rule (VERDICT): build nothing unless W is indistinguishable from the fp32 re-order control AND the control says re-orders
are benign.

    python tests/studies/winograd_study.py [--genomes 256] [--out profiles/r04_winograd_study.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np
import torch

from evolutionary_illusion_generator_amd import genome as genome_mod, grids, synth, weights
from evolutionary_illusion_generator_amd.engine import Engine
from oracle import classify
from oracle.prednet_torch import PredNetTorch

KEYS = ("genomes", "nonzero_both", "identical_frames", "within_1e-4", "outside_1e-4", "outside_1e-4_unexplained", "within_1e-4_of_nonzero_both",
        "zero_on_one_side_only", "byte_flip_rate", "max_byte_diff", "max_rel", "cliff_genomes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=256)
    ap.add_argument("--w", type=int, default=256)
    ap.add_argument("--h", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--large-tiles", action="store_true", help="round 5: also F(3x3,3x3) and F(4x4,3x3) variants")
    a = ap.parse_args()
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    W, H, CH, ST, n = a.w, a.h, [3, 48, 96, 192], 1, a.genomes
    cfg = synth.make_config(2, 3)
    genomes = [g for _, g in synth.make_population(n, cfg, seed=0)]
    wts = weights.synthetic_prednet_weights(CH, W, H, seed=0)
    grid = grids.create_grid(ST, W, H, 10)
    eng = Engine(W, H, CH, n)
    eng.set_weights(wts)
    eng.set_grid([grid["x_mat"], grid["y_mat"]])
    gb = genome_mod.GenomeBatch(genomes, cfg, 3)
    d_img = torch.zeros((n, 3, H, W), dtype=torch.uint8, device="cuda")
    eng.render_cppn(gb, d_img)
    fit, vecs = eng.eval_images(d_img, n, ST, pairing=0)
    d_fr = torch.zeros((n, 2, 3, H, W), dtype=torch.uint8, device="cuda")
    eng.prednet_rollout(d_img, n, 21, 19, d_fr)
    torch.cuda.synchronize()
    imgs = d_img.cpu().numpy()
    sides = {"HIP canonical (fp32 MFMA, fma chains, 2x2 form)": (d_fr.cpu().numpy(), vecs, fit)}
    # (round 5) the larger tiles: F(4x4,3x3) needs 2.25 multiply-adds per output where F(2x2) needs 4 (1.78x fewer again), F(3x3,3x3) 2.78;
    # "layers >= 1" = the HIP path's own choice of operators (the image layer stays direct), "ConvLSTM only" = ConvA / ConvP stay im2col + matmul
    variants = [("A reference order, im2col + rocBLAS matmul", dict(conv="matmul"), 8), ("B reference order, MIOpen library convolution", dict(conv="library"), 8),
                ("W2 Winograd F(2x2,3x3) fp32, every 3x3 convolution", dict(conv="winograd"), 4)]
    if a.large_tiles:
        variants += [("W2' Winograd F(2x2,3x3) fp32, layers >= 1 (the HIP path's operators)", dict(conv="winograd", wino_min_layer=1), 4),
                     ("W3 Winograd F(3x3,3x3) fp32, layers >= 1", dict(conv="winograd3", wino_min_layer=1), 4),
                     ("W4 Winograd F(4x4,3x3) fp32, layers >= 1", dict(conv="winograd4", wino_min_layer=1), 4),
                     ("W4L Winograd F(4x4,3x3) fp32, ConvLSTM of layers >= 1 only (ConvA / ConvP: F(2x2))", dict(conv="winograd", conv_lstm="winograd4", wino_min_layer=1), 4),
                     ("W4T Winograd F(4x4,3x3) fp32, layers >= 2 only (layers 0, 1: im2col + matmul)", dict(conv="winograd4", wino_min_layer=2), 4)]
    for name, kw, batch in variants:
        t0 = time.time()
        sides[name] = classify.rollout_side(ST, W, H, imgs, PredNetTorch(wts, CH, W, H, device="cuda", order="chainer", **kw), batch=batch)
        print("%s: %.1f s" % (name, time.time() - t0), flush=True)
    names = list(sides)
    report = {"shape": [W, H], "channels": CH, "genomes": n, "pairs": {}}
    anchors = names[:3]   # HIP, A, B: every variant is compared with these (and these with each other), not with every other variant
    for i, x in enumerate(names):
        for y in names[i + 1:]:
            if x not in anchors and y not in anchors:
                continue
            s, _ = classify.compare_sides(ST, W, H, sides[x], sides[y])
            report["pairs"]["%s  vs  %s" % (x, y)] = {k: s[k] for k in KEYS}
            print("%-44s vs %-44s flips %.3g outside %d (unexplained %d) identical %d max_rel %.2g" % (x[:44], y[:44], s["byte_flip_rate"], s["outside_1e-4"], s["outside_1e-4_unexplained"], s["identical_frames"], s["max_rel"]), flush=True)
    if a.out:
        json.dump(report, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
