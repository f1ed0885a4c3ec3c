"""Measurement script (not product): where a block of the Winograd ConvLSTM kernel spends its time, from an -DEIG_TIMING=1 build.
    hipcc ... -DEIG_TIMING=1 -o scripts/_timing/libeigen_timing.so evolutionary_illusion_generator_amd/csrc/eigen_engine.hip
    EIGEN_TIMELINE=gpurun_out/tl python scripts/timeline_wino.py [pop]
Every wave of one steady-state launch of each ConvLSTM operator records the cycle counter at kernel entry / K-loop start / K-loop end /
exit, the cycles it spent waiting (s_waitcnt + barrier) and working inside the K loop, and HW_ID (conv_wino.h: timeline)."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

out_dir = os.environ.setdefault("EIGEN_TIMELINE", "gpurun_out/tl")
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

for path in sorted(glob.glob(os.path.join(out_dir, "timeline_H*.bin"))):
    if path.endswith("_up4.bin"):
        continue
    r = np.fromfile(path, dtype=np.uint64).reshape(-1, 8, 8)          # [block][wave][field]
    r = r[r[:, 0, 0] != 0]
    t_entry, t_l0, t_l1, t_end = (r[:, :, i].astype(np.int64) for i in range(4))
    wait, work, nkb = r[:, :, 5].astype(np.int64), r[:, :, 6].astype(np.int64), int(r[0, 0, 7])
    hw = r[:, :, 4]
    hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
    simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 7
    tot = (t_end - t_entry).mean()
    print("== %s: %d blocks, %d K-blocks per block" % (os.path.basename(path), len(r), nkb))
    print("   per wave (cycles): prologue %.0f (%.1f %%)  K loop %.0f (%.1f %%)  epilogue %.0f (%.1f %%)  total %.0f" % (
        (t_l0 - t_entry).mean(), 100 * (t_l0 - t_entry).mean() / tot, (t_l1 - t_l0).mean(), 100 * (t_l1 - t_l0).mean() / tot,
        (t_end - t_l1).mean(), 100 * (t_end - t_l1).mean() / tot, tot))
    print("   inside the K loop, per K-block: working %.0f  waiting at s_waitcnt + barrier %.0f (%.1f %% of the loop); 64 MFMAs of a wave = 2048 pipe cycles, 4096 when the SIMD's two waves share the pipe"
          % (work.mean() / nkb, wait.mean() / nkb, 100.0 * wait.mean() / (wait + work).mean()))
    for h in (0, 1):
        print("   waves %d-%d: working %.0f waiting %.0f per K-block" % (4 * h, 4 * h + 3, work[:, 4 * h:4 * h + 4].mean() / nkb, wait[:, 4 * h:4 * h + 4].mean() / nkb))
    # which waves of a block share a SIMD?
    pairs = {}
    for b in range(min(len(r), 2000)):
        for w in range(8):
            pairs.setdefault(int(simd[b, w]), set())
        key = tuple(int(simd[b, w]) for w in range(8))
        pairs[key] = pairs.get(key, 0) + 1 if isinstance(pairs.get(key, 0), int) else 1
    top = sorted(((v, k) for k, v in pairs.items() if isinstance(k, tuple)), reverse=True)[:3]
    print("   SIMD of waves 0..7 (most frequent patterns):", [(k, v) for v, k in top])
