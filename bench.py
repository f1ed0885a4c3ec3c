#!/usr/bin/env python3
"""Headline benchmark: genome fitness evaluations per second at 256x256, pop = 256, on 1/2/4/8 MI355X (BASELINE.json).

One "step" = one generation's fitness evaluation of the population through the drop-in path
(evolutionary_illusion_generator_amd.fitness.population_fitness: flatten genomes -> [rank 0 broadcasts the wire arrays] ->
CPPN render -> PredNet 21-step roll-out -> Lucas-Kanade -> score -> all-gather of the fitness scalars).  Workload =
BASELINE.json configs[2]: neat_configs/circles.txt (num_hidden 20, 3 outputs), colour, channels 3,48,96,192, Circles
structure, 256x256, ONE population of 256 genomes.  Data: seeded synthetic genomes and seeded synthetic PredNet weights
(the trained weights are external downloads and fix the size to 160x120).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--weak] [--no-cpu-baseline] [--no-roofline]

--gpus N > 1 without a torch.distributed environment re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU, backend
nccl = RCCL); launched that way by the driver it just joins.  Either way it asserts WORLD_SIZE == N.
Scaling: the metric names ONE population of 256, so for N > 1 that population is sharded N ways ("scaling": "strong",
32 genomes per GPU at N = 8); --weak keeps --pop genomes PER GPU instead (supplementary).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel: the fused
ConvLSTM 3x3 convolution on the fp32 MFMA pipe, timed live with HIP events on its launch stream), `roofline_hbm` (the two
HBM-side stages: CPPN render and the flow stencils) and `cpu_baseline` (the CPU oracle's path -- numpy CPPN + torch-CPU fp32
PredNet + C Lucas-Kanade + numpy scores -- on a bounded sample of >= 8 genomes).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_TBS = 8.0            # MI355X_MICROARCH.md: HBM3E
N_STEPS_PREDNET = 21          # steps 1-20 + first extension (the 22nd step is never read on the population path)

SHAPES = {
    # name: (W, H, channels, c_dim, structure, n_hidden, n_outputs, default global pop, label)
    "headline": (256, 256, [3, 48, 96, 192], 3, 1, 20, 3, 256, "neat_configs/circles.txt colour"),
    "ref160": (160, 120, [3, 48, 96, 192], 3, 1, 20, 3, 50, "neat_configs/circles.txt colour"),
    "c2": (160, 120, [1, 16, 32, 64], 1, 1, 20, 1, 50, "neat_configs/circles_bw.txt gray"),
    "c4": (256, 256, [3, 48, 96, 192], 3, 0, 8, 6, 512, "neat_configs/bands.txt colour (first 3 of 6 outputs)"),
    "c5": (512, 512, [3, 48, 96, 192], 3, 2, 20, 6, 1024, "neat_configs/free.txt colour (first 3 of 6 outputs)"),
}
STRUCT_NAMES = ["Bands", "Circles", "Free", "CirclesFree"]
SCORE_NAMES = ["horizontal-symmetry", "rotation-symmetry", "swarm", "rotation-symmetry"]


def kernel_sources_sha():
    """Fingerprint of everything that is compiled into libeigen_hip.so: a PMC summary is only valid for the build it was
    taken on (profiles/pmc_summary_latest.json carries the same hash, written by scripts/summarize_pmc.py)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "evolutionary_illusion_generator_amd", "csrc")
    for f in sorted(x for x in os.listdir(csrc) if x.endswith((".h", ".hip"))) + [os.path.join(ROOT, "include", "eigen_engine.h")]:
        p = f if os.path.isabs(f) else os.path.join(csrc, f)
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def git_head():
    """Commit the running build was made at: git here, BUILD_INFO.json (written by __graft_entry__.build()) on the GPU box."""
    try:
        return subprocess.check_output(["git", "-C", ROOT, "describe", "--always", "--dirty"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    try:
        return json.load(open(os.path.join(ROOT, "evolutionary_illusion_generator_amd", "BUILD_INFO.json")))["commit"]
    except Exception:
        return None


def cpu_baseline(cfg, pop, wts, grid, shape, min_genomes=8, seconds_budget=30.0):
    """The oracle's CPU path on a bounded sample of the same workload (rank 0, N = 1 only), with a per-stage split."""
    import numpy as np
    import torch
    from oracle import pipeline, scores
    from oracle.prednet_torch import PredNetTorch
    import oracle
    W, H, CHANNELS, C_DIM, STRUCTURE = shape[:5]
    net = PredNetTorch(wts, CHANNELS, W, H)
    threads = torch.get_num_threads()
    done, t0 = 0, time.time()
    fits, split = [], {"render_s": 0.0, "prednet_s": 0.0, "flow_s": 0.0, "score_s": 0.0}
    while done < len(pop) and (done < min_genomes or time.time() - t0 < seconds_budget):
        g = pop[done][1]
        ta = time.time()
        img = pipeline.render_chw(g, cfg, grid, C_DIM, W, H)
        tb = time.time()
        frames, _ = net.rollout(img[None], n_repeat=20, n_ext=1)
        tc = time.time()
        v = oracle.lucas_kanade(frames[0, 19], frames[0, 20])
        td = time.time()
        fits.append(scores.fitness_from_vectors(STRUCTURE, v.astype(np.float64), W, H))
        te = time.time()
        for k, d in zip(split, (tb - ta, tc - tb, td - tc, te - td)):
            split[k] += d
        done += 1
    dt = time.time() - t0
    return {"value": done / dt, "unit": "genome evals/s", "cores": threads, "kind": "port",
            "host_cpus": os.cpu_count(), "genomes": done, "seconds": dt,
            "per_stage_s_per_genome": {k: v / done for k, v in split.items()},
            "sample": "%d genomes of the same %dx%d population, full path (numpy float64 CPPN 1 thread, torch-CPU/oneDNN fp32 "
                      "PredNet 21 steps on %d threads, C Lucas-Kanade 1 thread, numpy scores), %.1f s" % (done, W, H, threads, dt)}, fits


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` from a plain shell: become N ranks (one per GPU) under torch.distributed.run."""
    import torch
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (n, have))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pop", type=int, default=None, help="population size (global; per GPU with --weak). Default: the shape's (256)")
    ap.add_argument("--weak", action="store_true", help="supplementary: keep --pop genomes PER GPU (weak scaling) instead of sharding ONE population")
    ap.add_argument("--shape", default="headline", choices=sorted(SHAPES),
                    help="headline: 256x256 colour pop 256 (BASELINE.json metric, configs[2]); ref160: the reference's own default "
                         "160x120 colour, pop 50 (the only published datum: 0.80 evals/s on a Colab GPU, BASELINE.md); "
                         "c2: configs[1] circles_bw 160x120 gray (channels 1,16,32,64) pop 50; c4: configs[3] bands.txt (8 hidden, "
                         "6 outputs) 256x256 colour Bands pop 512; c5: configs[4] free.txt 512x512 colour Free structure pop 1024")
    ap.add_argument("--flow", default="lk", choices=["lk", "farneback"],
                    help="supplementary: 'farneback' swaps the reference's Lucas-Kanade call for the dense Farneback option (no CPU leg)")
    ap.add_argument("--source", default=None, choices=["rank0", "replicated"], help="multi-rank genome source (fitness.GENOME_SOURCE)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_backend() == "nccl" and dist.get_world_size() == args.gpus

    shape = SHAPES[args.shape]
    W, H, CHANNELS, C_DIM, STRUCTURE, n_hidden, n_outputs, default_pop, label = shape
    supplementary = args.shape != "headline" or args.flow != "lk" or args.weak
    if supplementary:
        args.no_cpu_baseline = True  # supplementary numbers: no CPU leg
    pop_arg = args.pop or default_pop
    global_pop = pop_arg * world if args.weak else pop_arg
    per_rank = -(-global_pop // world)
    max_batch = min(per_rank, 256)

    from evolutionary_illusion_generator_amd import fitness, grids, synth, weights
    if args.flow != "lk":
        fitness.FLOW_METHOD = args.flow
    cfg = synth.make_config(2, n_outputs)
    population = synth.make_population(global_pop, cfg, seed=0, num_hidden=n_hidden)  # identical on every rank (seeded)
    genomes = [g for _, g in population]
    wts = weights.synthetic_prednet_weights(CHANNELS, W, H, seed=0)
    eng = fitness.get_engine(wts, W, H, CHANNELS, max_batch=max_batch, **({} if args.flow == "lk" else {"flow": args.flow}))

    def step():
        return fitness.population_fitness(STRUCTURE, genomes, wts, cfg, W, H, CHANNELS, c_dim=C_DIM, gradient=1,
                                          max_batch=max_batch, source=args.source)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
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
        "metric": "genome fitness evals/sec at %dx%d, pop=%d" % (W, H, pop_arg) + (" per GPU" if args.weak else ""),
        "value": global_pop * args.steps / dt,
        "unit": "genome evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s pop=%d%s, %dx%d, PredNet %s, 21 steps, %s + %s score" % (
                       label, pop_arg, "/GPU" if args.weak else "", W, H, ",".join(map(str, CHANNELS)),
                       {"lk": "LK", "farneback": "Farneback dense flow"}[args.flow], SCORE_NAMES[STRUCTURE]),
                   "global_pop": global_pop, "genomes_per_gpu": per_rank, "device_batch": max_batch, "image": [W, H],
                   "channels": CHANNELS, "structure": STRUCT_NAMES[STRUCTURE],
                   "parallelism": "pop-shard x%d (%s) + all-gather(fitness f64, %s)" % (
                       world, "genome wire arrays broadcast from rank 0" if (args.source or fitness.GENOME_SOURCE) == "rank0" else "replicated seeded populations",
                       "RCCL" if world > 1 else "single process")},
        "stage_ms_last_step": {k: round(v, 3) for k, v in stage.items() if k.endswith("_ms") and k != "conv_ms"},
        "nonzero_fitness": int((fit != 0).sum()),
        "commit": git_head(), "kernel_sources_sha": kernel_sources_sha(),
    }

    if rank == 0 and not args.no_roofline:
        nb = min(max_batch, len(genomes))
        # per-launch HIP-event timing of every conv kernel of one more roll-out pass (events on the launch stream)
        eng.conv_profile(True, reset=True)
        fitness.evaluate_population(STRUCTURE, genomes[:nb], wts, cfg, W, H, CHANNELS, c_dim=C_DIM, gradient=1, max_batch=max_batch)
        rows = eng.conv_profile(False, reset=True)
        lstm = [r for r in rows if r["epi"] == "lstm" and r["NI"] == 4]
        fl = sum(r["flops_per_image"] * nb * r["launches"] for r in lstm)
        ms = sum(r["ms"] for r in lstm)
        n_l = sum(r["launches"] for r in lstm)
        all_fl = sum(r["flops_per_image"] * nb * r["launches"] for r in rows)
        all_ms = sum(r["ms"] for r in rows)
        ach = fl / (ms * 1e-3) / 1e12
        # HBM traffic per launch of the same kernel: rocprofv3 --pmc passes cannot be collected from inside this process
        # (separate runs of this command: scripts/pmc_passes.sh + scripts/summarize_pmc.py; FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for 16 B/lane reads).  The summary is stamped with the hash of the kernel sources
        # it was taken on; a summary of another build is NOT reported (traffic = null).
        traffic, traffic_commit, traffic_note = None, None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary_latest.json")))
            traffic_commit = pm.get("commit")
            if pm.get("kernel_sources_sha") != out["kernel_sources_sha"]:
                traffic_note = "profiles/pmc_summary_latest.json was taken on kernel sources %s, this build is %s: not reported" % (
                    pm.get("kernel_sources_sha"), out["kernel_sources_sha"])
            elif pm.get("pop") != nb:
                traffic_note = "PMC summary is for a device batch of %s genomes, this run uses %d: not reported" % (pm.get("pop"), nb)
            else:
                for kname, kv in pm["kernels"].items():
                    if "conv3x3_mfma<4, 16, 1" in kname:
                        traffic = kv.get("hbm_read_bytes_per_launch", 0.0) + kv.get("hbm_write_bytes_per_launch", 0.0)
        except Exception as e:  # noqa: BLE001
            traffic_note = "no PMC summary: %s" % e
        flops_step = eng.flops_per_step()
        out["roofline"] = {"bound": "mfma", "kernel": "conv3x3_mfma<4,16,EPI_LSTM> (fused ConvLSTM gates, v_mfma_f32_16x16x4_f32)",
                           "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                           "traffic": traffic, "traffic_commit": traffic_commit, "traffic_note": traffic_note,
                           "traffic_source": "profiles/pmc_summary_latest.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" if traffic else None,
                           "launches": n_l, "avg_launch_ms": ms / max(n_l, 1),
                           "algorithmic_flops_per_launch": fl / max(n_l, 1),
                           # the reference's formulation (9 taps on every source) vs what this build executes
                           "algorithmic_flops_reference_per_genome": flops_step * N_STEPS_PREDNET,
                           "executed_flops_per_genome": all_fl / nb,
                           "effective_tflops_reference_formulation": flops_step * N_STEPS_PREDNET * out["value"] / 1e12,
                           "all_conv_kernels": {"achieved": all_fl / (all_ms * 1e-3) / 1e12, "frac": all_fl / (all_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                                "total_ms": all_ms, "launches": sum(r["launches"] for r in rows)},
                           "per_op": [{"layer": r["layer"], "op": r["epi"] + ("(step 0: zero sources skipped)" if r.get("step0") else ""), "ms": round(r["ms"], 3), "launches": r["launches"],
                                       "tflops": (r["flops_per_image"] * nb * r["launches"] / (r["ms"] * 1e-3) / 1e12) if r["ms"] > 0 else 0.0}
                                      for r in rows if r["launches"] > 0]}
        # The two HBM-side stages (SURVEY 8(d) "report both"): algorithmic bytes / HIP-event time of the stage / 8 TB/s.
        N = W * H
        n_in = 2
        render_bytes = nb * (C_DIM * N + n_in * N * 8)        # uint8 planes out + the float64 coordinate planes read per genome block
        flow_bytes = nb * int(2 * N * (1 + 0.25 + 1.0 / 16) * (1 + 2 * 2))  # 2 gray frames x 3 pyramid levels x (u8 + 2 x i16 derivatives)
        st = eng.timings()
        hb = []
        for name, b, msk in (("cppn_render_kernel (a2+a3)", render_bytes, "render_ms"), ("flow stage: gray, pyrDown, Scharr, min-eig, corner select, LK track (a6)", flow_bytes, "flow_ms")):
            t_ms = st[msk]
            a_ = b / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
            hb.append({"bound": "hbm", "kernel": name, "algorithmic_bytes": b, "ms": t_ms, "achieved": a_, "peak": PEAK_HBM_TBS,
                       "unit": "TB/s", "frac": a_ / PEAK_HBM_TBS})
        out["roofline_hbm"] = hb
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        grid = grids.create_grid(STRUCTURE, W, H, 10)
        cb, cpu_fit = cpu_baseline(cfg, population, wts, grid, shape)
        out["cpu_baseline"] = cb
        out["gpu_over_cpu"] = out["value"] / cb["value"]
        # parity spot check at the FULL size against the bit-exact C oracle (one genome, ~20 s of CPU) and against the
        # independently ordered torch-CPU PredNet of the CPU leg (north_star: 1e-4 relative)
        from oracle import pipeline
        t1 = time.time()
        ref0 = pipeline.genome_fitness(genomes[0], cfg, grid, wts, CHANNELS, W, H, STRUCTURE)
        cpu_fit = np.asarray(cpu_fit)
        gpu_s = fit[:len(cpu_fit)]
        rel = np.abs(gpu_s - cpu_fit) / np.maximum(np.abs(cpu_fit), 1e-300)
        rel[(cpu_fit == 0) & (gpu_s == 0)] = 0.0
        out["parity_check"] = {"genome": 0, "gpu": float(fit[0]), "oracle_c": float(ref0),
                               "rel_err": float(abs(fit[0] - ref0) / max(abs(ref0), 1e-300)) if ref0 != 0 else float(abs(fit[0])),
                               "torch_cpu_sample": [float(x) for x in cpu_fit], "gpu_sample": [float(x) for x in gpu_s],
                               "max_rel_err_vs_independent_order": float(rel.max()) if len(rel) else None,
                               "oracle_seconds": time.time() - t1}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # rank 0 is still in its (untimed) roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
