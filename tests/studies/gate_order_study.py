#!/usr/bin/env python3
"""CPU study: does moving the canonical gate epilogue to the reference's knowable element-wise order
(EIG_GATE_ORDER=1: rounded peephole products, sigmoid = tanh(x/2)/2 + 1/2, un-fused cell update) bring the canonical frames
closer to the reference-order implementations than the round 1-3 epilogue (EIG_GATE_ORDER=0)?

Both canonical variants are the C oracle (= the HIP path bit for bit, tests/test_gpu_parity.py), compiled twice
(oracle/Makefile: libeig_oracle.so, `make gate0` -> libeig_oracle_gate0.so).  Each is compared with
  (a) the C oracle's own statement of the chainer element order (order="chainer": 9-tap unpooled source, separate tensors, libm tanh)
  (b) torch-CPU / oneDNN in the chainer element order (oracle/prednet_torch.py)
on the two frames Lucas-Kanade reads, per genome: flipped bytes, fitness deviation, genomes outside 1e-4.

    python tests/studies/gate_order_study.py [--shape c2|ref160|headline] [--genomes N] [--out profiles/r04_gate_order_cpu.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import pipeline, scores  # noqa: E402
from oracle.prednet_torch import PredNetTorch  # noqa: E402

SHAPES = {"c2": (160, 120, [1, 16, 32, 64], 1, 1), "ref160": (160, 120, [3, 48, 96, 192], 3, 1), "headline": (256, 256, [3, 48, 96, 192], 3, 1)}


def rollout_with(lib, wts, ch, w, h, img, order=0):
    names = oracle.tensor_names(len(ch))
    arrs = [np.ascontiguousarray(wts[n], dtype=np.float32) for n in names]
    tab = (ctypes.POINTER(ctypes.c_float) * len(arrs))(*[a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for a in arrs])
    cha = np.asarray(ch, dtype=np.int32)
    out = np.zeros((21, ch[0], h, w), dtype=np.uint8)
    rc = lib.eig_oracle_prednet_rollout_order(ctypes.c_int(len(ch)), cha.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.c_int(w), ctypes.c_int(h), tab,
                                              img.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_int(20), ctypes.c_int(1), ctypes.c_int(0),
                                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), None, ctypes.c_int(order))
    assert rc == 0
    return out[19:21]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="c2", choices=sorted(SHAPES))
    ap.add_argument("--genomes", type=int, default=40)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from evolutionary_illusion_generator_amd import grids, synth, weights
    w, h, ch, c_dim, st = SHAPES[args.shape]
    cfg = synth.make_config(2, c_dim)
    genomes = [g for _, g in synth.make_population(args.genomes, cfg, seed=0)]
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=0)
    grid = grids.create_grid(st, w, h, 10)
    lib1 = ctypes.CDLL(os.path.join(ROOT, "oracle", "libeig_oracle.so"))
    lib0 = ctypes.CDLL(os.path.join(ROOT, "oracle", "libeig_oracle_gate0.so"))
    assert lib1.eig_oracle_gate_order() == 1 and lib0.eig_oracle_gate_order() == 0
    net = PredNetTorch(wts, ch, w, h, order="chainer")
    fit = lambda fr: scores.fitness_from_vectors(st, oracle.lucas_kanade(fr[0], fr[1]).astype(np.float64), w, h)
    rows = []
    t0 = time.time()
    for i, g in enumerate(genomes):
        img = pipeline.render_chw(g, cfg, grid, c_dim, w, h)
        f1, f0 = rollout_with(lib1, wts, ch, w, h, img), rollout_with(lib0, wts, ch, w, h, img)
        fc = rollout_with(lib1, wts, ch, w, h, img, order=1)
        ft = net.rollout(img[None], n_repeat=20, n_ext=1)[0][0, 19:21]
        fits = {k: fit(v) for k, v in (("gate1", f1), ("gate0", f0), ("chainer_c", fc), ("chainer_torch", ft))}
        r = {"genome": i, "fit": fits}
        for a, fa in (("gate1", f1), ("gate0", f0)):
            for b, fb in (("chainer_c", fc), ("chainer_torch", ft)):
                ref = fits[b]
                r["%s_vs_%s" % (a, b)] = {"flips": int((fa != fb).sum()),
                                          "rel": 0.0 if fits[a] == ref else (abs(fits[a] - ref) / abs(ref) if ref != 0 else float("inf"))}
        r["chainer_c_vs_chainer_torch"] = {"flips": int((fc != ft).sum()), "rel": 0.0 if fits["chainer_c"] == fits["chainer_torch"] else (
            abs(fits["chainer_c"] - fits["chainer_torch"]) / abs(fits["chainer_torch"]) if fits["chainer_torch"] != 0 else float("inf"))}
        r["gate1_vs_gate0"] = {"flips": int((f1 != f0).sum())}
        rows.append(r)
        print(i, {k: (v["flips"], "%.1e" % v.get("rel", 0)) for k, v in r.items() if isinstance(v, dict) and "flips" in v}, flush=True)
    nbytes = 2 * c_dim * h * w
    summ = {"shape": args.shape, "genomes": len(rows), "seconds": time.time() - t0, "bytes_per_genome": nbytes}
    for k in ("gate1_vs_chainer_c", "gate0_vs_chainer_c", "gate1_vs_chainer_torch", "gate0_vs_chainer_torch", "chainer_c_vs_chainer_torch"):
        rel = np.array([r[k]["rel"] for r in rows])
        summ[k] = {"byte_flip_rate": sum(r[k]["flips"] for r in rows) / float(nbytes * len(rows)), "flips": sum(r[k]["flips"] for r in rows),
                   "outside_1e-4": int(((rel > 1e-4) | ~np.isfinite(rel)).sum()), "max_rel": float(rel[np.isfinite(rel)].max()) if np.isfinite(rel).any() else None,
                   "identical_frames": sum(1 for r in rows if r[k]["flips"] == 0)}
    summ["gate1_vs_gate0_flips"] = sum(r["gate1_vs_gate0"]["flips"] for r in rows)
    summ["nonzero_fitness"] = sum(1 for r in rows if r["fit"]["gate1"] != 0)
    print(json.dumps(summ, indent=1))
    if args.out:
        json.dump({"summary": summ, "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
