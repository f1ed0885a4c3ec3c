"""Measurement script (not product): time the conv kernel at PredNet layer shapes, optionally with an ablation build.
usage: python scripts/ablate_conv.py [lib.so ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from evolutionary_illusion_generator_amd import engine

SHAPES = {  # name: (B, H, W, cout, [(cin, up)...])   EPI_RAW with cout = 4*C mimics the gate GEMM
    "L2-lstm-like": (64, 64, 64, 384, [(192, 0), (192, 1), (96, 0)]),
    "L1-lstm-like": (64, 128, 128, 192, [(96, 0), (96, 1), (48, 0)]),
    "L3-lstm-like": (64, 32, 32, 768, [(384, 0), (192, 0)]),
    "L1-lstm-main": (64, 128, 128, 192, [(96, 0), (48, 0)]),   # without the unpooled source (its 2x2-form pass is a separate launch)
    "L2-lstm-main": (64, 64, 64, 384, [(192, 0), (96, 0)]),
    "L0-lstm-like": (64, 256, 256, 16, [(6, 0), (48, 1), (3, 0)]),
    "convA1-like": (64, 256, 256, 48, [(6, 0)]),
    "convA2-like": (64, 128, 128, 96, [(96, 0)]),
    "convP1-like": (64, 128, 128, 48, [(48, 0)]),
}
libs = sys.argv[1:] or [engine.LIB_PATH]
for lib in libs:
    engine._lib = None
    engine.load_library(lib)
    e = engine.Engine(32, 32, [1, 4], 1)
    for name, (B, H, W, cout, srcs) in SHAPES.items():
        rng = np.random.default_rng(0)
        ds = [torch.from_numpy(rng.normal(0, 1, (B, c, H >> u, W >> u)).astype(np.float32)).cuda() for c, u in srcs]
        hw = [rng.normal(0, 0.1, (cout, c, 3, 3)).astype(np.float32) for c, _ in srcs]
        out = torch.empty((B, cout, H, W), device="cuda")
        ms = e.time_conv(ds, [c for c, _ in srcs], [u for _, u in srcs], hw, cout, H, W, B, out, iters=5)
        fl = 2.0 * B * H * W * cout * sum(c * (4 if u else 9) for c, u in srcs)  # executed multiply-adds (2x2 form: 4 per channel)
        print("%-40s %-14s %.3f ms  %.1f TFLOP/s" % (os.path.basename(lib), name, ms, fl / ms * 1e-9), flush=True)
    e.close()
