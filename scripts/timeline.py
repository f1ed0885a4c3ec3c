"""Measurement script (not product): per-SIMD timeline of the ConvLSTM kernel from an -DEIG_TIMING=1 build.
    python __graft_entry__.py --lib scripts/_timing/libeigen_timing.so -DEIG_TIMING=1
    EIGEN_TIMELINE=gpurun_out python scripts/timeline.py [pop]
Every wave of one steady-state launch of each ConvLSTM op records s_memtime at kernel entry / K-loop start / K-loop end /
exit plus HW_ID; this script rebuilds, per SIMD, how much of the time two, one or no wave was inside the MFMA loop."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

out_dir = os.environ.setdefault("EIGEN_TIMELINE", "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
if "--analyze-only" not in sys.argv:
    import torch
    from evolutionary_illusion_generator_amd import engine, fitness, synth, weights
    engine.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_timing", "libeigen_timing.so"))
    pop = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
    W = H = 256; ch = [3, 48, 96, 192]
    cfg = synth.make_config(2, 3)
    genomes = [g for _, g in synth.make_population(pop, cfg, seed=0)]
    wts = weights.synthetic_prednet_weights(ch, W, H, seed=0)
    fitness.evaluate_population(1, genomes, wts, cfg, W, H, ch, c_dim=3, max_batch=pop)
    torch.cuda.synchronize()

for path in sorted(glob.glob(os.path.join(out_dir, "timeline_H*.bin"))):   # one file per ConvLSTM op and per 2x2-form pass (*_up4)
    r = np.fromfile(path, dtype=np.uint64).reshape(-1, 4, 8)          # [block][wave][field]
    live = r[:, 0, 0] != 0
    r = r[live]
    nb = len(r)
    t_entry, t_l0, t_l1, t_end = (r[:, :, i].astype(np.int64) for i in range(4))
    hw = r[:, :, 4]
    hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
    wave_id, simd, cu, sh, se = hwid & 0xf, (hwid >> 4) & 3, (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd                # one SIMD
    t0 = t_entry.min()
    print("==", os.path.basename(path), "blocks", nb, "span %.3f ms @2.4GHz-equivalent cycles %d" % ((t_end.max() - t0) / 100e6 * 1e3, t_end.max() - t0))
    print("   per wave: prologue %.0f  loop %.0f  epilogue %.0f  total %.0f  (s_memtime ticks)" % (
        (t_l0 - t_entry).mean(), (t_l1 - t_l0).mean(), (t_end - t_l1).mean(), (t_end - t_entry).mean()))
    t_setup, t_prewait = r[:, :, 6].astype(np.int64), r[:, :, 7].astype(np.int64)   # conv_mfma.h: t_setup, t_prewait
    print("   prologue split: entry -> slots/descriptors ready %.0f, -> first K-block issued, gather addresses ready %.0f, -> landed + barrier %.0f;"
          " MFMA sections of the K loop %.0f" % (np.median(t_setup - t_entry), np.median(t_prewait - t_setup), np.median(t_l0 - t_prewait), r[:, :, 5].mean()))
    print("   distinct SIMDs %d, CUs %d, xcc ids %s; blockIdx%%8 == xcc for %.1f%% of blocks" % (
        len(np.unique(key)), len(np.unique(key // 4)), np.unique(xcc).tolist(),
        100.0 * (xcc[:, 0] == (np.nonzero(live)[0] % 8)).mean()))
    # per SIMD: sweep over events
    frac = np.zeros(4)
    gaps = []
    tot = 0
    for k in np.unique(key):
        m = key == k
        e0, l0, l1, e1, wid = t_entry[m], t_l0[m], t_l1[m], t_end[m], wave_id[m]
        lo, hi = e0.min(), e1.max()
        ev = np.concatenate([np.stack([l0, np.ones_like(l0)], 1), np.stack([l1, -np.ones_like(l1)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        cur, last = 0, lo
        for t, d in ev:
            frac[min(cur, 3)] += t - last
            last, cur = t, cur + d
        frac[0] += hi - last
        tot += hi - lo
        for w in np.unique(wid):                                            # dispatch gap per wave slot
            mm = wid == w
            o = np.argsort(e0[mm])
            g = e0[mm][o][1:] - e1[mm][o][:-1]
            gaps.append(g)
    gaps = np.concatenate(gaps) if gaps else np.zeros(0)
    print("   SIMD time with 0/1/2/3+ waves in the K loop: %s" % " ".join("%.1f%%" % (100 * f / tot) for f in frac))
    print("   slot turnaround (exit -> next wave's entry in the same wave slot): median %.0f  mean %.0f  p90 %.0f ticks; busy-span coverage %.1f%%" % (
        np.median(gaps), gaps.mean(), np.percentile(gaps, 90), 100.0 * tot / (len(np.unique(key)) * (t_end.max() - t0))))
