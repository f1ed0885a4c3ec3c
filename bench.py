#!/usr/bin/env python3
"""Headline benchmark: genome fitness evaluations per second at 256x256, pop = 256 per GPU (BASELINE.json).

One "step" = one generation's fitness evaluation of the population through the drop-in path
(evolutionary_illusion_generator_amd.fitness: flatten genomes -> CPPN render -> PredNet 21-step roll-out ->
Lucas-Kanade -> score -> all-gather of the fitness scalars).  Workload = BASELINE.json configs[2]:
neat_configs/circles.txt (num_hidden 20, 3 outputs), colour, channels 3,48,96,192, Circles structure, 256x256,
pop 256 per rank (weak scaling: every rank evaluates its own 256 genomes; the population shards with no
data-path collective, the only exchange is the all-gather of 256*N float64).  Data: seeded synthetic genomes and
seeded synthetic PredNet weights (the trained weights are external downloads and fix the size to 160x120).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline] [--no-roofline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel: the fused
ConvLSTM 3x3 convolution on the fp32 MFMA pipe, timed live with HIP events on its launch stream) and `cpu_baseline`
(the CPU oracle's path -- torch-CPU fp32 PredNet + C Lucas-Kanade + numpy CPPN/scores -- on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

W, H = 256, 256
CHANNELS = [3, 48, 96, 192]
C_DIM = 3
POP_PER_GPU = 256
STRUCTURE = 1  # Circles
N_STEPS_PREDNET = 21  # steps 1-20 + first extension (the 22nd step is never read on the population path)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense


def cpu_baseline(cfg, pop, wts, grid, seconds_budget=25.0):
    """The oracle's CPU path on a bounded sample of the same workload (rank 0, N = 1 only)."""
    from oracle import cppn, pipeline, scores
    from oracle.prednet_torch import PredNetTorch
    import oracle
    net = PredNetTorch(wts, CHANNELS, W, H)
    threads = torch.get_num_threads()
    done, t0 = 0, time.time()
    fits = []
    while done < len(pop) and (done < 2 or time.time() - t0 < seconds_budget):
        g = pop[done][1]
        img = pipeline.render_chw(g, cfg, grid, C_DIM, W, H)
        frames, _ = net.rollout(img[None], n_repeat=20, n_ext=1)
        v = oracle.lucas_kanade(frames[0, 19], frames[0, 20])
        fits.append(scores.fitness_from_vectors(STRUCTURE, v.astype(np.float64), W, H))
        done += 1
    dt = time.time() - t0
    return {"value": done / dt, "unit": "genome evals/s", "cores": threads, "kind": "port",
            "sample": "%d genomes of the same 256x256 colour population, full path (numpy CPPN, torch-CPU fp32 PredNet "
                      "21 steps, C Lucas-Kanade, numpy scores), %.1f s" % (done, dt)}, fits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pop", type=int, default=POP_PER_GPU, help="genomes per GPU")
    ap.add_argument("--shape", default="headline", choices=["headline", "ref160", "c2", "c4", "c5"],
                    help="headline: 256x256 colour pop 256 (BASELINE.json metric, configs[2]); ref160: the reference's own default "
                         "160x120 colour, pop 50 (the only published datum: 0.80 evals/s on a Colab GPU, BASELINE.md); "
                         "c2: configs[1] circles_bw 160x120 gray (channels 1,16,32,64) pop 50; c4: configs[3] per-GPU share, "
                         "bands.txt (8 hidden, 6 outputs) 256x256 colour Bands pop 64; "
                         "c5: configs[4] per-GPU share, 512x512 colour Free structure, pop 128")
    ap.add_argument("--flow", default="lk", choices=["lk", "farneback"],
                    help="supplementary: 'farneback' swaps the reference's Lucas-Kanade call for the dense Farneback option (no CPU leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    global W, H
    if args.shape == "ref160":
        W, H = 160, 120
        if args.pop == POP_PER_GPU:
            args.pop = 50
        args.no_cpu_baseline = True  # supplementary number: no CPU leg, no PMC traffic
    global STRUCTURE, CHANNELS, C_DIM
    n_hidden, n_outputs = 20, 3
    if args.shape == "c2":
        W, H, CHANNELS, C_DIM, n_outputs = 160, 120, [1, 16, 32, 64], 1, 1
        if args.pop == POP_PER_GPU:
            args.pop = 50
        args.no_cpu_baseline = True
    if args.shape == "c4":
        STRUCTURE, n_hidden, n_outputs = 0, 8, 6
        if args.pop == POP_PER_GPU:
            args.pop = 64
        args.no_cpu_baseline = True
    if args.shape == "c5":
        W, H, STRUCTURE = 512, 512, 2
        if args.pop == POP_PER_GPU:
            args.pop = 128
        args.no_cpu_baseline = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from evolutionary_illusion_generator_amd import fitness, grids, synth, weights
    if args.flow != "lk":
        fitness.FLOW_METHOD = args.flow
        args.no_cpu_baseline = True
    cfg = synth.make_config(2, n_outputs)
    global_pop = args.pop * world
    population = synth.make_population(global_pop, cfg, seed=0, num_hidden=n_hidden)  # identical on every rank (seeded)
    genomes = [g for _, g in population]
    wts = weights.synthetic_prednet_weights(CHANNELS, W, H, seed=0)
    eng = fitness.get_engine(wts, W, H, CHANNELS, max_batch=args.pop, **({} if args.flow == "lk" else {"flow": args.flow}))

    def step():
        def evaluate(lo, hi):
            return fitness.evaluate_population(STRUCTURE, genomes[lo:hi], wts, cfg, W, H, CHANNELS, c_dim=C_DIM,
                                               gradient=1, max_batch=args.pop)
        return fitness.sharded_map(len(genomes), evaluate)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        scores_w = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fit = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stage = eng.timings()

    out = {
        "metric": "genome fitness evals/sec at %dx%d, pop=%d per GPU" % (W, H, args.pop),
        "value": global_pop * args.steps / dt,
        "unit": "genome evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s pop=%d/GPU, %dx%d, PredNet %s, 21 steps, %s + %s score" % (
                       {"headline": "neat_configs/circles.txt colour", "ref160": "neat_configs/circles.txt colour", "c2": "neat_configs/circles_bw.txt gray",
                        "c4": "neat_configs/bands.txt colour (first 3 of 6 outputs)", "c5": "neat_configs/free.txt colour"}[args.shape], args.pop, W, H,
                       ",".join(map(str, CHANNELS)), {"lk": "LK", "farneback": "Farneback dense flow"}[args.flow], ["horizontal-symmetry", "rotation-symmetry", "swarm", "rotation-symmetry"][STRUCTURE]),
                   "global_pop": global_pop, "image": [W, H], "channels": CHANNELS, "structure": ["Bands", "Circles", "Free", "CirclesFree"][STRUCTURE],
                   "parallelism": "pop-shard x%d + all-gather(fitness f64)" % world},
        "stage_ms_last_step": {k: round(v, 3) for k, v in stage.items() if k.endswith("_ms") and k != "conv_ms"},
        "nonzero_fitness": int((fit != 0).sum()),
    }

    if rank == 0 and not args.no_roofline:
        # per-launch HIP-event timing of every conv kernel of one more roll-out pass (events on the launch stream)
        eng.conv_profile(True, reset=True)
        step_local = fitness.evaluate_population(STRUCTURE, genomes[:args.pop], wts, cfg, W, H, CHANNELS, c_dim=C_DIM, gradient=1, max_batch=args.pop)
        rows = eng.conv_profile(False, reset=True)
        lstm = [r for r in rows if r["epi"] == "lstm" and r["NI"] == 4]
        fl = sum(r["flops_per_image"] * args.pop * r["launches"] for r in lstm)
        ms = sum(r["ms"] for r in lstm)
        n_l = sum(r["launches"] for r in lstm)
        all_fl = sum(r["flops_per_image"] * args.pop * r["launches"] for r in rows)
        all_ms = sum(r["ms"] for r in rows)
        ach = fl / (ms * 1e-3) / 1e12
        # HBM traffic per launch of the same kernel: from the committed rocprofv3 --pmc passes (separate runs of this
        # command, scripts/pmc_passes.sh + scripts/summarize_pmc.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
        # for 16 B/lane reads).  PMC cannot be collected from inside this process, hence read from profiles/.
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary_latest.json")))
            for kname, kv in pm["kernels"].items():
                if "conv3x3_mfma<4, 16, 1" in kname and pm.get("pop") == args.pop:
                    traffic = kv.get("hbm_read_bytes_per_launch", 0.0) + kv.get("hbm_write_bytes_per_launch", 0.0)
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": "conv3x3_mfma<4,16,EPI_LSTM> (fused ConvLSTM gates, v_mfma_f32_16x16x4_f32)",
                           "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                           "traffic": traffic, "traffic_source": "profiles/pmc_summary_latest.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" if traffic else None, "launches": n_l, "avg_launch_ms": ms / max(n_l, 1),
                           "algorithmic_flops_per_launch": fl / max(n_l, 1),
                           "all_conv_kernels": {"achieved": all_fl / (all_ms * 1e-3) / 1e12, "frac": all_fl / (all_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                                "total_ms": all_ms, "launches": sum(r["launches"] for r in rows)},
                           "per_op": [{"layer": r["layer"], "op": r["epi"] + ("(step 0: zero sources skipped)" if r.get("step0") else ""), "ms": round(r["ms"], 3), "launches": r["launches"],
                                       "tflops": (r["flops_per_image"] * args.pop * r["launches"] / (r["ms"] * 1e-3) / 1e12) if r["ms"] > 0 else 0.0}
                                      for r in rows if r["launches"] > 0]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        grid = grids.create_grid(STRUCTURE, W, H, 10)
        cb, cpu_fit = cpu_baseline(cfg, population, wts, grid)
        out["cpu_baseline"] = cb
        out["gpu_over_cpu"] = out["value"] / cb["value"]
        # parity spot check at the FULL size against the bit-exact C oracle (one genome, ~20 s of CPU)
        from oracle import pipeline
        t1 = time.time()
        ref0 = pipeline.genome_fitness(genomes[0], cfg, grid, wts, CHANNELS, W, H, STRUCTURE)
        out["parity_check"] = {"genome": 0, "gpu": float(fit[0]), "oracle_c": float(ref0),
                               "rel_err": float(abs(fit[0] - ref0) / max(abs(ref0), 1e-300)) if ref0 != 0 else float(abs(fit[0])),
                               "torch_cpu_sample": [float(x) for x in cpu_fit], "gpu_sample": [float(x) for x in fit[:len(cpu_fit)]],
                               "oracle_seconds": time.time() - t1}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # rank 0 is still in its (untimed) roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
