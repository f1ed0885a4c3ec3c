"""Measurement script (not product): where the host-side time of one generation goes (flatten, engine call, rest)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from evolutionary_illusion_generator_amd import fitness, synth, weights, genome

W = H = 256; CH = [3, 48, 96, 192]; POP = 256
cfg = synth.make_config(2, 3)
pop = synth.make_population(POP, cfg, seed=0, num_hidden=20)
gs = [g for _, g in pop]
wts = weights.synthetic_prednet_weights(CH, W, H, seed=0)
eng = fitness.get_engine(wts, W, H, CH, max_batch=POP)
fitness.evaluate_population(1, gs, wts, cfg, W, H, CH, c_dim=3, gradient=1, max_batch=POP)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gb = genome.GenomeBatch(gs, cfg, 3, n_leaves=2)
    t1 = time.perf_counter()
    out = eng.eval_population(gb, 1, bg=1, gradient=1, pairing=0)
    t2 = time.perf_counter()
    r = fitness.sharded_map(len(gs), lambda lo, hi: fitness.evaluate_population(1, gs[lo:hi], wts, cfg, W, H, CH, c_dim=3, gradient=1, max_batch=POP))
    t3 = time.perf_counter()
    st = eng.timings()
    print("flatten %.2f ms | engine call %.2f ms (stages: %s) | whole step via sharded_map %.2f ms" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, {k: round(v, 2) for k, v in st.items()}, (t3 - t2) * 1e3), flush=True)
