"""Measurement script (not product): where the time between two K loops of the Winograd F(4x4) kernel goes, from an
-DEIG_TIMING=1 build (csrc/conv_wino4.h: timeline; written in round 4 for the sixteen-wave F(2x2) kernel, which round 6 removed -- EIG_TL_WAVES=12 is the F(4x4) block).
    python __graft_entry__.py --lib scripts/_timing/libeigen_timing.so -DEIG_TIMING=1
    EIGEN_TIMELINE=gpurun_out/tl EIG_TL_WAVES=12 EIGEN_TIMELINE_ALL=1 python scripts/timeline_wino4.py [pop]          (--analyze-only: read the .bin files again)
Every wave of one steady-state launch of each ConvLSTM operator records the cycle counter at entry / set-up done (first DMA about to be
issued) / K loop start / K loop end / exchange barrier passed / row transform done (gates start) / exit, and HW_ID.  Blocks are then
ordered per CU: the matrix pipe of a CU is idle from the K-loop end of one block to the K-loop start of the next (ONE block per CU),
and that window is split into its pieces."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

GHZ = float(os.environ.get("EIG_GFX_GHZ", "2.4"))
NW = int(os.environ.get("EIG_TL_WAVES", "12"))   # waves per block of conv_wino4.h (6: half blocks); one record per block and N-block of its walk
out_dir = os.environ.setdefault("EIGEN_TIMELINE", "gpurun_out/tl")
os.makedirs(out_dir, exist_ok=True)
if "--analyze-only" not in sys.argv:
    import torch
    from evolutionary_illusion_generator_amd import engine, fitness, synth, weights
    engine.load_library(os.environ.get("EIGEN_TIMING_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_timing", "libeigen_timing.so")))
    pop = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
    W = H = 256; ch = [3, 48, 96, 192]
    cfg = synth.make_config(2, 3)
    genomes = [g for _, g in synth.make_population(pop, cfg, seed=0)]
    wts = weights.synthetic_prednet_weights(ch, W, H, seed=0)
    fitness.evaluate_population(1, genomes, wts, cfg, W, H, ch, c_dim=3, max_batch=pop)   # (EIGEN_WINOGRAD selects the kernel: set EIG_TL_WAVES = 16 / 12 to match)
    torch.cuda.synchronize()


def us(c):
    return c / (GHZ * 1e3)


for path in sorted(glob.glob(os.path.join(out_dir, "timeline_H*.bin"))):
    if path.endswith("_up4.bin"):
        continue
    raw = np.fromfile(path, dtype=np.uint64)
    if raw.size % (NW * 8):
        continue
    r = raw.reshape(-1, NW, 8)
    r = r[(r[:, :, 0] != 0).all(axis=1)]
    if not len(r):
        continue
    t = r[:, :, :7].astype(np.int64)
    hw = r[:, 0, 7]
    hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
    cu_key = (xcc << 8) | (((hwid >> 13) & 7) << 5) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf)
    names = ["entry", "setup", "k0", "k1", "x", "y", "end"]
    E, S, K0, K1, X, Y, END = (t[:, :, i] for i in range(7))
    print("== %s: %d blocks on %d CUs" % (os.path.basename(path), len(r), len(np.unique(cu_key))))
    print("   per wave, mean cycles (us at %.1f GHz): set-up %.0f (%.2f)  prologue DMA + first transform %.0f (%.2f)  K loop %.0f (%.2f)  publish + barrier %.0f (%.2f)  row transform %.0f (%.2f)  gates + stores %.0f (%.2f)"
          % (GHZ, (S - E).mean(), us((S - E).mean()), (K0 - S).mean(), us((K0 - S).mean()), (K1 - K0).mean(), us((K1 - K0).mean()),
             (X - K1).mean(), us((X - K1).mean()), (Y - X).mean(), us((Y - X).mean()), (END - Y).mean(), us((END - Y).mean())))
    for role, sl in (("transforming waves (first half)", slice(0, NW // 2)), ("fetching waves (second half)", slice(NW // 2, NW))):
        print("   %s: set-up %.0f  prologue %.0f  K loop %.0f  publish+barrier %.0f  row transform %.0f  gates+stores %.0f" % (
            role, (S - E)[:, sl].mean(), (K0 - S)[:, sl].mean(), (K1 - K0)[:, sl].mean(), (X - K1)[:, sl].mean(), (Y - X)[:, sl].mean(), (END - Y)[:, sl].mean()))
    print("   inside a block: first wave in -> last wave in %.0f cycles; first wave out -> last wave out %.0f; K-loop end spread %.0f" % (
        (E.max(axis=1) - E.min(axis=1)).mean(), (END.max(axis=1) - END.min(axis=1)).mean(), (K1.max(axis=1) - K1.min(axis=1)).mean()))
    # co-resident blocks (kernels with two blocks per CU): share of a CU's busy span during which NO block of it is inside its K loop
    cov, spans, conc = 0.0, 0.0, 0.0
    for key in np.unique(cu_key):
        idx = np.nonzero(cu_key == key)[0]
        ev = sorted([(K0[i].mean(), 1) for i in idx] + [(K1[i].mean(), -1) for i in idx])
        depth, last, c1, c2 = 0, ev[0][0], 0.0, 0.0
        for tt, dlt in ev:
            if depth >= 1: c1 += tt - last
            if depth >= 2: c2 += tt - last
            depth += dlt; last = tt
        cov += c1; conc += c2; spans += ev[-1][0] - ev[0][0]
    print("   per CU: some block inside its K loop %.1f %% of the span (two or more: %.1f %%) -> matrix pipe without any K loop %.1f %%" % (100 * cov / spans, 100 * conc / spans, 100 * (1 - cov / spans)))
    # successive blocks of a CU
    idle, parts = [], []
    for key in np.unique(cu_key):
        idx = np.nonzero(cu_key == key)[0]
        idx = idx[np.argsort(E[idx].min(axis=1))]
        for a_, b_ in zip(idx[:-1], idx[1:]):
            k1a, k0b = K1[a_].mean(), K0[b_].mean()
            idle.append(k0b - k1a)
            parts.append((X[a_].mean() - k1a, Y[a_].mean() - X[a_].mean(), END[a_].mean() - Y[a_].mean(), END[a_].max() - END[a_].mean(),
                          E[b_].min() - END[a_].max(), E[b_].mean() - E[b_].min(), S[b_].mean() - E[b_].mean(), k0b - S[b_].mean(),
                          K1[a_].mean() - K0[a_].mean()))
    idle, parts = np.asarray(idle, dtype=np.float64), np.asarray(parts, dtype=np.float64)
    if len(idle):
        m = parts.mean(axis=0)
        print("   MATRIX PIPE IDLE between the K loops of successive blocks of a CU: mean %.0f cycles = %.2f us (median %.0f, p90 %.0f); K loop %.0f cycles = %.2f us -> idle share %.1f %%"
              % (idle.mean(), us(idle.mean()), np.median(idle), np.percentile(idle, 90), m[8], us(m[8]), 100 * idle.mean() / (idle.mean() + m[8])))
        lab = ["publish + exchange barrier", "row transform (LDS reads)", "gates + stores (mean wave)", "  ... until the LAST wave is out", "last wave out -> first wave of the next block in (dispatcher)",
               "first wave in -> mean wave in (launch of the block's waves)", "set-up (addresses, descriptors)", "prologue: DMA round trip, first transform, 2 barriers"]
        for l_, v in zip(lab, m[:8]):
            print("      %-64s %7.0f cycles  %5.2f us  %4.1f %%" % (l_, v, us(v), 100 * v / idle.mean()))
