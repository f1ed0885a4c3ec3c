"""GPU (round 2): the north-star parity claim on the box, BASELINE.json configs[3] / configs[4] end to end, a12, RCCL.

* HIP fitness against an INDEPENDENTLY ORDERED fp32 PredNet (torch-CPU / oneDNN, oracle/prednet_torch.py: library
  convolutions, library sigmoid/tanh, plain unpool -> 9-tap conv) within north_star's 1e-4 relative, with the uint8
  frame flip rate printed.  The bit-exact tests elsewhere prove "kernel == the build's canonical arithmetic"; this one
  bounds what the choice of that arithmetic is worth.
* configs[3]: neat_configs/bands.txt (num_hidden 8, num_outputs 6 -> first 3; bands.txt:48-51), Bands grid at 256x256
  colour (build-defined generalisation), horizontal_symmetry_score.
* configs[4]: neat_configs/free.txt (num_hidden 20, num_outputs 6; free.txt:48-50), 512x512 colour, Free grid
  (generate_illusion.py:308-315), full 21-step roll-out + LK + swarm_score.
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from evolutionary_illusion_generator_amd import fitness, genome as genome_mod, synth, weights
from evolutionary_illusion_generator_amd.engine import Engine


def _hip_population(cuda, w, h, ch, structure, genomes, cfg, wts):
    """-> (stimuli, the two frames the HIP path hands to Lucas-Kanade, its vectors, its fitness, the fitness of the fused
    eval_population entry point), everything through the C ABI."""
    import torch
    from oracle import grids as ogrids
    n, c_dim = len(genomes), ch[0]
    grid = ogrids.create_grid(structure, w, h, 10)
    e = Engine(w, h, ch, n)
    e.set_weights(wts)
    e.set_grid([grid["x_mat"], grid["y_mat"]])
    gb = genome_mod.GenomeBatch(genomes, cfg, c_dim)
    hip = e.eval_population(gb, structure)
    d_img = torch.zeros((n, c_dim, h, w), dtype=torch.uint8, device=cuda)
    e.render_cppn(gb, d_img)
    fit, vecs = e.eval_images(d_img, n, structure, pairing=0)
    assert np.array_equal(fit, hip)
    d_fr = torch.zeros((n, 2, c_dim, h, w), dtype=torch.uint8, device=cuda)
    e.prednet_rollout(d_img, n, 21, 19, d_fr)
    torch.cuda.synchronize()
    imgs, frames = d_img.cpu().numpy(), d_fr.cpu().numpy()
    e.close()
    return imgs, frames, vecs, hip, grid


def _assert_explained(s, min_nonzero, floor_all, floor_nonzero):
    """The north-star tolerance as a property that holds for EVERY genome (oracle/classify.py), with the floors at what was
    measured: `floor_all` over all genomes, `floor_nonzero` over the genomes that score non-zero on
    both sides (a genome that scores 0 on both sides is trivially 'within 1e-4' and must not pad the count)."""
    from oracle import classify
    assert s["nonzero_both"] >= min_nonzero, "only %d non-zero genomes: vacuous" % s["nonzero_both"]
    assert s["max_byte_diff"] <= 1 and s["byte_flip_rate"] < 5e-5  # (measured 1.2e-5 at 256^2 colour, 2e-6 at 160x120 gray)
    assert s["max_rel_identical"] <= 1e-9            # same frames -> same vectors -> same fitness (float64 sum order only)
    assert s["outside_1e-4_unexplained"] == 0, s["outside_1e-4_detail"]
    for d in s["outside_1e-4_detail"]:               # a genome outside 1e-4: a handful of +-1 bytes that reproduce it on OUR frames
        assert 1 <= d["flips"] <= classify.MAX_FLIPS and d["explained"], d
    assert s["max_rel"] <= 2e-2, s["max_rel"]        # absolute cap on any finite deviation (measured max 4.5e-3)
    assert s["within_1e-4"] >= floor_all * s["genomes"], (s["within_1e-4"], s["genomes"])
    assert s["within_1e-4_of_nonzero_both"] >= floor_nonzero * s["nonzero_both"], (s["within_1e-4_of_nonzero_both"], s["nonzero_both"])


def _few_cpu_threads():
    """torch-CPU convolutions at batch 1 are fastest on ~16 threads; the GPU boxes default to 128 of their 256 CPUs (5x slower)."""
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def test_hip_fitness_vs_reference_element_order_c2(cuda, oracle_lib):
    """BASELINE.json configs[1]: circles_bw, 160x120 gray, channels 1,16,32,64, 40 genomes.  HIP path against the reference's
    element-wise order twice: on torch-CPU (oneDNN convolutions) every genome is classified with the measured floors; against the
    torch-free C statement of that order (oracle.PredNetC, host-independent) the north-star bar holds as a HARD assert: every
    genome within 1e-4 (measured max 3.7e-5)."""
    from oracle import classify, pipeline
    from oracle.prednet_torch import PredNetTorch
    w, h, ch, structure = 160, 120, [1, 16, 32, 64], 1
    cfg = synth.make_config(2, 1)
    genomes = [g for _, g in synth.make_population(40, cfg, seed=0)]
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=0)
    imgs, frames, vecs, hip, grid = _hip_population(cuda, w, h, ch, structure, genomes, cfg, wts)
    for i in (0, 7, 23):
        assert np.array_equal(imgs[i], pipeline.render_chw(genomes[i], cfg, grid, 1, w, h))  # same stimulus on both sides
    _few_cpu_threads()
    s, _ = classify.population_report(structure, w, h, imgs, frames, vecs, hip, PredNetTorch(wts, ch, w, h, order="chainer"))
    print("\nC2 160x120 gray vs chainer element order (torch-CPU): %s" % s)
    _assert_explained(s, 8, 0.97, 0.95)
    sc, _ = classify.population_report(structure, w, h, imgs, frames, vecs, hip, oracle_lib.PredNetC(wts, ch, w, h, order="chainer"))
    print("C2 vs chainer element order (C oracle): %s" % sc)
    assert sc["outside_1e-4"] == 0 and sc["max_rel"] <= 1e-4 and sc["zero_on_one_side_only"] == 0, sc   # north_star, hard
    assert sc["nonzero_both"] >= 8 and sc["max_byte_diff"] <= 1


def test_hip_fitness_vs_reference_element_order_c3(cuda, oracle_lib):
    """BASELINE.json configs[2], the headline shape (256x256 colour, 3,48,96,192): 24 genomes against torch-CPU / oneDNN in the
    reference's element-wise order (population floors), and -- VERDICT r5 item 4 -- the first four genomes that score non-zero
    against the torch-free C statement of that order (oracle.PredNetC(order="chainer"): one fp32 chain per output, plain loops,
    host-independent) with HARD per-genome asserts: every differing byte is +-1, at most 2e-5 of the bytes differ, and a genome
    outside north_star's 1e-4 must be reproduced by single-byte flips of the HIP path's own frames (classify.explained)."""
    from oracle import classify
    from oracle.prednet_torch import PredNetTorch
    w, h, ch, structure = 256, 256, [3, 48, 96, 192], 1
    cfg = synth.make_config(2, 3)
    genomes = [g for _, g in synth.make_population(24, cfg, seed=0)]   # (24 genomes with the population-level floors; rounds 3-4: 8 genomes at 0.75 / 0.6)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=0)
    imgs, frames, vecs, hip, _ = _hip_population(cuda, w, h, ch, structure, genomes, cfg, wts)
    _few_cpu_threads()
    nz = [i for i in range(len(genomes)) if hip[i] != 0][:4]
    assert len(nz) == 4, "fewer than four non-zero genomes among 24: vacuous"
    s, _ = classify.population_report(structure, w, h, imgs, frames, vecs, hip, PredNetTorch(wts, ch, w, h, order="chainer"), batch=1)
    print("\nC3 256x256 colour vs chainer element order (torch-CPU): %s" % s)
    _assert_explained(s, 8, 0.88, 0.80)
    sc, rows = classify.population_report(structure, w, h, imgs[nz], frames[nz], [vecs[i] for i in nz], np.asarray(hip)[nz],
                                          oracle_lib.PredNetC(wts, ch, w, h, order="chainer"), batch=1)
    print("C3 genomes %s vs chainer element order (C oracle): %s" % (nz, sc))
    assert sc["max_byte_diff"] <= 1 and sc["byte_flip_rate"] <= 2e-5, sc
    assert sc["zero_on_one_side_only"] == 0 and sc["outside_1e-4_unexplained"] == 0, sc["outside_1e-4_detail"]
    assert sc["max_rel_identical"] <= 1e-9 and sc["max_rel"] <= 2e-2, sc


def test_every_genome_outside_1e4_is_a_single_lsb_case_128_genomes(cuda, oracle_lib):
    """The population-level view at the headline shape.  The reference separates its stages by uint8 PNGs,
    picks corners by a RELATIVE quality threshold and drops vectors by hard thresholds, so ONE flipped byte (+-1 at a
    quantisation boundary) moves a genome's fitness by 1e-5 .. 1e-2 -- whatever implementation flips it (the reference's own
    cuDNN vs CPU paths included).  128 genomes against the reference's element-wise order with im2col + rocBLAS matmul
    convolutions on the GPU (oracle/prednet_torch.py order="chainer", conv="matmul"; oracle C Lucas-Kanade and numpy scores on
    its frames).  Checked for EVERY genome: byte differences are +-1; identical frames give identical fitness; a genome
    outside 1e-4 has a handful of flipped bytes which, applied to the HIP path's own frames, reproduce the deviation
    (classify.explained) -- the deviation is the conditioning of the reference's fitness function.
    CONTROL: the same classification between two NON-HIP implementations of the reference's order (matmul vs
    the MIOpen library convolution): they must disagree with each other the way HIP disagrees with either of them.
    bench.py's parity_check leg repeats both on all 256 genomes of the headline population."""
    import torch
    from oracle import classify
    from oracle.prednet_torch import PredNetTorch
    w, h, ch, structure, n = 256, 256, [3, 48, 96, 192], 1, 128
    cfg = synth.make_config(2, 3)
    genomes = [g for _, g in synth.make_population(n, cfg, seed=0)]
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=0)
    imgs, frames, vecs, hip, _ = _hip_population(cuda, w, h, ch, structure, genomes, cfg, wts)
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    side_a = classify.rollout_side(structure, w, h, imgs, PredNetTorch(wts, ch, w, h, device="cuda", conv="matmul", order="chainer"), batch=8)
    s, rows = classify.population_report(structure, w, h, imgs, frames, vecs, hip, None, other=side_a)
    print("\n128 genomes 256x256 colour vs chainer element order (torch-GPU matmul): %s" % s)
    _assert_explained(s, 48, 0.88, 0.80)
    assert s["zero_on_one_side_only"] <= 2  # (len(good) > 24 is one more cliff; such a genome is in outside_1e-4_detail, explained)
    side_b = classify.rollout_side(structure, w, h, imgs, PredNetTorch(wts, ch, w, h, device="cuda", conv="library", order="chainer"), batch=8)
    ctl = classify.control_report(structure, w, h, side_a, side_b, "chainer order, matmul (GPU)", "chainer order, MIOpen (GPU)")
    print("CONTROL reference-order A vs reference-order B: %s" % ctl)
    assert ctl["control_max_byte_diff"] <= 1 and ctl["control_byte_flip_rate"] < 5e-5
    # two-sided: two implementations of the reference's OWN order are no closer to each other than HIP is to one of them, AND HIP is no
    # further from the reference order than they are from each other (measured on 256 genomes: 15 vs 18-25 outside, 0.86e-5 vs 1.7e-5 flips)
    assert ctl["control_outside_1e-4"] >= 0.5 * s["outside_1e-4"] - 2, (ctl, s["outside_1e-4"])
    assert s["outside_1e-4"] <= 1.5 * ctl["control_outside_1e-4"] + 3, (s["outside_1e-4"], ctl)
    assert s["byte_flip_rate"] <= 1.5 * ctl["control_byte_flip_rate"], (s["byte_flip_rate"], ctl)


def test_config3_bands_256_colour_end_to_end(cuda, oracle_lib):
    """configs[3]: bands.txt genomes (8 hidden, 6 outputs -> the first 3 are rendered) on the Bands grid at 256x256
    (the reference raises there: build-defined generalisation, restated independently in oracle/grids.py), PredNet
    3,48,96,192, horizontal_symmetry_score.  Two genomes against the bit-exact C oracle (~20 s each); 64 through the
    size-independent properties."""
    import torch
    from oracle import grids as ogrids, pipeline
    w, h, ch, structure = 256, 256, [3, 48, 96, 192], 0
    cfg = synth.make_config(2, 6)
    pop = synth.make_population(64, cfg, seed=3, num_hidden=8)
    genomes = [g for _, g in pop]
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=0)
    grid = ogrids.create_grid(structure, w, h, 10, generalised=True)
    got = fitness.evaluate_population(structure, genomes, wts, cfg, w, h, ch, c_dim=3, gradient=1, max_batch=64)
    assert np.isfinite(got).all() and (got != 0).sum() >= 8, got
    # batch-size / batch-position invariance and determinism: reversed order in smaller device batches, bit for bit
    rev = fitness.evaluate_population(structure, genomes[::-1], wts, cfg, w, h, ch, c_dim=3, gradient=1, max_batch=24)
    assert np.array_equal(rev[::-1], got)
    # renders of all 64 byte-exact vs the oracle (6-output genomes: first three outputs)
    imgs = fitness.render_images(structure, genomes, wts, cfg, w, h, ch, c_dim=3, gradient=1, max_batch=64)
    ref_imgs = np.stack([pipeline.render_chw(g, cfg, grid, 3, w, h) for g in genomes])
    assert np.array_equal(imgs, ref_imgs)
    nz = [i for i in range(64) if got[i] != 0]
    for i in nz[:2]:
        ref = pipeline.image_fitness(ref_imgs[i], wts, ch, w, h, structure)
        assert ref != 0 and got[i] == pytest.approx(ref, rel=1e-9, abs=1e-12), (i, got[i], ref)


def test_config4_free_512_colour_end_to_end(cuda, oracle_lib):
    """configs[4]: free.txt genomes (20 hidden, 6 outputs -> first 3) on the Free grid at 512x512 colour: HIP render
    byte-exact, FULL 21-step roll-out + Lucas-Kanade + swarm_score of one genome against the C oracle (~1.5 min of CPU),
    properties for the rest."""
    import torch
    from oracle import grids as ogrids, pipeline
    w, h, ch, structure = 512, 512, [3, 48, 96, 192], 2
    cfg = synth.make_config(2, 6)
    pop = synth.make_population(6, cfg, seed=4, num_hidden=20)
    genomes = [g for _, g in pop]
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=1)
    grid = ogrids.create_grid(structure, w, h, 10)
    batch = genomes + [genomes[1], genomes[0]]
    got = fitness.evaluate_population(structure, batch, wts, cfg, w, h, ch, c_dim=3, gradient=1, max_batch=8)
    assert got[6] == got[1] and got[7] == got[0] and np.isfinite(got).all()
    again = fitness.evaluate_population(structure, genomes[:3], wts, cfg, w, h, ch, c_dim=3, gradient=1, max_batch=8)
    assert np.array_equal(again, got[:3])
    imgs = fitness.render_images(structure, genomes, wts, cfg, w, h, ch, c_dim=3, gradient=1, max_batch=8)
    ref_imgs = np.stack([pipeline.render_chw(g, cfg, grid, 3, w, h) for g in genomes])
    assert np.array_equal(imgs, ref_imgs)
    nz = [i for i in range(6) if got[i] != 0]
    assert nz, "all six genomes scored 0: vacuous"
    i = nz[0]
    ref = pipeline.image_fitness(ref_imgs[i], wts, ch, w, h, structure)
    assert ref != 0 and got[i] == pytest.approx(ref, rel=1e-9, abs=1e-12), (i, got[i], ref)


def test_inside_outside_score_on_the_device(cuda):
    """a12: fitness_calculator.inside_outside_score (:219-304) -- device scorer against the reference-generated fixture
    (vectors rounded to the float32 the flow stage produces, re-scored by the oracle's restatement) and raw against it."""
    import json
    from oracle import scores
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "round2.json")))["inside_outside"]
    nz = 0
    for c in cases:
        v = np.asarray(c["vectors"], dtype=np.float64).reshape(-1, 4)
        v32 = v.astype(np.float32).astype(np.float64)
        got = fitness.inside_outside_score(v32, c["w"], c["h"])
        ref = float(scores.inside_outside_score(v32, c["w"], c["h"]))
        assert got == pytest.approx(ref, rel=1e-9, abs=1e-12), (c["w"], c["h"], len(v), got, ref)
        assert got == pytest.approx(c["score"], rel=1e-5, abs=1e-7)  # float32 rounding of dx, dy only
        nz += got != 0
    assert nz >= 7
    # positions outside the image: the reference indexes numpy arrays with int(x / step) -- past the end raises
    # IndexError, a negative index wraps Python-style; the oracle (numpy indexing, like the reference) is the witness
    v = np.asarray(cases[1]["vectors"], dtype=np.float64).reshape(-1, 4).astype(np.float32).astype(np.float64)
    w_, h_ = cases[1]["w"], cases[1]["h"]
    wrap = v.copy(); wrap[0, 0] = -1.25 * (w_ / 5); wrap[1, 1] = -2.5 * (w_ / 5); wrap[2, 0] = -0.5   # cells -1, -2 wrap; int(-0.x) = 0 does not
    assert fitness.inside_outside_score(wrap, w_, h_) == pytest.approx(float(scores.inside_outside_score(wrap, w_, h_)), rel=1e-9, abs=1e-12)
    for bad in ((0, 6.5 * (w_ / 5)), (1, (int(h_ / (w_ / 5)) + 1.5) * (w_ / 5)), (0, -7.0 * (w_ / 5))):
        out = v.copy(); out[0, bad[0]] = bad[1]
        with pytest.raises(IndexError):
            scores.inside_outside_score(out, w_, h_)
        with pytest.raises(IndexError):
            fitness.inside_outside_score(out, w_, h_)
    with pytest.raises(NameError):   # the reference's own behaviour for an unknown structure is kept
        fitness.calculate_fitness(4, np.zeros((3, 4)), "x.png", 160, 120)


def test_best_flow_vectors_are_the_oracles(cuda, oracle_lib, tmp_path):
    """best_flow.png is an overlay of the flow the fitness used (generate_illusion.py:655-657 copies the LK visualisation):
    the vectors behind it (saved next to it) are the oracle's, bit for bit, and the overlay marks every one of them."""
    from PIL import Image
    from oracle import grids as ogrids, pipeline
    w, h, ch, structure = 96, 64, [3, 12, 24, 48], 1
    cfg = synth.make_config(2, 3)
    pop = synth.make_population(9, cfg, seed=31)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=2)
    fitness.get_fitnesses_neat(structure, pop, wts, cfg, w, h, ch, c_dim=3, best_dir=str(tmp_path), gradient=1)
    got = np.array([g.fitness for _, g in pop])
    best = max(range(len(pop)), key=lambda i: (got[i], i))
    grid = ogrids.create_grid(structure, w, h, 10)
    img = pipeline.render_chw(pop[best][1], cfg, grid, 3, w, h)
    ref_v = pipeline.image_vectors(img, wts, ch, w, h)
    v = np.load(tmp_path / "best_flow_vectors.npy")
    assert v.shape == ref_v.shape and np.array_equal(v, ref_v) and len(v) > 0
    overlay = np.asarray(Image.open(tmp_path / "best_flow.png").convert("RGB")).astype(int)
    base = img.transpose(1, 2, 0).astype(int)
    for x, y, dx, dy in v:   # a yellow dot at every tracked corner
        px = overlay[int(round(y)), int(round(x))]
        assert tuple(px) == (255, 255, 0), (x, y, px)
    assert (overlay != base).any(axis=2).sum() >= len(v)
    # enhanced.png is written for every structure (generate_illusion.py:665-671)
    for st in (0, 2):
        sub = tmp_path / ("s%d" % st)
        fitness.get_fitnesses_neat(st, pop, wts, cfg, w, h, ch, c_dim=3, best_dir=str(sub), gradient=1)
        assert Image.open(sub / "enhanced.png").size == (800, 800)


def test_engine_cache_does_not_reload_weights(cuda, tmp_path, monkeypatch):
    """get_engine resolved (re-read / re-generated) the weight set on every call, even on a cache hit."""
    calls = []
    real = fitness.synthetic_prednet_weights
    monkeypatch.setattr(fitness, "synthetic_prednet_weights", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    fitness.clear_engines()
    cfg = synth.make_config(2, 1)
    pop = synth.make_population(3, cfg, seed=1)
    for _ in range(3):
        fitness.get_fitnesses_neat(2, pop, "synthetic:5", cfg, 64, 64, [1, 4, 8], c_dim=1, best_dir=str(tmp_path))
    assert len(calls) == 1
    with pytest.raises(ValueError):
        fitness.get_vectors(np.zeros((20, 20), np.uint8), "synthetic:5", [1, 4, 8], 64, 64)   # smaller than the engine: refused on the host


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _nccl_worker(rank, world, port, q, source):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from evolutionary_illusion_generator_amd import fitness as F, synth as S, weights as W
    F.GENOME_SOURCE = source
    w, h, ch = 64, 64, [1, 8, 16]
    cfg = S.make_config(2, 1)
    pop = S.make_population(7, cfg, seed=12)
    wts = W.synthetic_prednet_weights(ch, w, h, seed=6)
    F.get_fitnesses_neat(2, pop, wts, cfg, w, h, ch, c_dim=1, best_dir=None)
    q.put((rank, [g.fitness for _, g in pop], dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("source", ["rank0", "replicated"])
def test_two_ranks_over_rccl(cuda, oracle_lib, source):
    """One rank per GPU on the `nccl` (= RCCL) backend: broadcast of the genome wire arrays + all-gather of the fitness
    scalars on device tensors.  Needs two GPUs; the single-GPU boxes of the test tier skip it (the gloo twin runs there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (have %d)" % torch.cuda.device_count())
    import torch.multiprocessing as mp
    from oracle import grids as ogrids, pipeline
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q, source)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (f, b) for r, f, b in (q.get(timeout=600) for _ in procs)}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    w, h, ch = 64, 64, [1, 8, 16]
    cfg = synth.make_config(2, 1)
    pop = synth.make_population(7, cfg, seed=12)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=6)
    grid = ogrids.create_grid(2, w, h, 10)
    ref = np.array([pipeline.genome_fitness(g, cfg, grid, wts, ch, w, h, 2) for _, g in pop])
    assert res[0][1] == res[1][1] == "nccl"
    assert res[0][0] == res[1][0]
    assert np.allclose(res[0][0], ref, rtol=1e-9, atol=1e-12)


_RCCL1_SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, RANK="0", WORLD_SIZE="1", EIGEN_DIST_SINGLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from evolutionary_illusion_generator_amd import fitness as F, synth as S, weights as W
assert F._dist() is not None and dist.get_backend() == "nccl"
w, h, ch = 64, 64, [1, 8, 16]
cfg = S.make_config(2, 1)
wts = W.synthetic_prednet_weights(ch, w, h, seed=6)
out = {}
for source in ("rank0", "replicated"):
    F.GENOME_SOURCE = source
    pop = S.make_population(7, cfg, seed=12)
    F.get_fitnesses_neat(2, pop, wts, cfg, w, h, ch, c_dim=1, best_dir=None)
    out[source] = [g.fitness for _, g in pop]
blob = bytes(range(256)) * 40 + b"tail"
assert F._broadcast_bytes(blob) == blob
v, ex = F.sharded_map(5, lambda lo, hi: np.arange(lo, hi) * 1.5, extra=7.0)
assert v.tolist() == [0.0, 1.5, 3.0, 4.5, 6.0] and ex.tolist() == [7.0]
dist.barrier(); dist.destroy_process_group()
import json
print("RCCL1", json.dumps(out))
"""


def test_collective_path_on_rccl_with_one_rank(cuda, oracle_lib):
    """The multi-rank code path (broadcast of the genome wire arrays, all-gather of the fitness scalars, barrier) on the `nccl` =
    RCCL backend with DEVICE tensors, in a process group of ONE rank -- what a single-GPU box can exercise of it."""
    import subprocess
    from oracle import grids as ogrids, pipeline
    r = subprocess.run([sys.executable, "-c", _RCCL1_SCRIPT % {"root": ROOT, "port": str(_free_port())}], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL1")][-1]
    import json
    out = json.loads(line[len("RCCL1 "):])
    w, h, ch = 64, 64, [1, 8, 16]
    cfg = synth.make_config(2, 1)
    pop = synth.make_population(7, cfg, seed=12)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=6)
    grid = ogrids.create_grid(2, w, h, 10)
    ref = np.array([pipeline.genome_fitness(g, cfg, grid, wts, ch, w, h, 2) for _, g in pop])
    assert out["rank0"] == out["replicated"]
    assert np.allclose(out["rank0"], ref, rtol=1e-9, atol=1e-12) and (ref != 0).any()


def test_bench_gpus_flag_launches_ranks(cuda):
    """`python bench.py --gpus N` must BE N ranks (the flag was parsed and dropped).  On a 1-GPU box: N = 1 prints
    n_gpus 1 and N = 2 refuses loudly instead of silently running one rank."""
    import json
    import subprocess
    import torch
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    small = ["--shape", "c2", "--pop", "8", "--steps", "1", "--warmup", "1", "--no-roofline", "--no-cpu-baseline"]
    bench = os.path.join(ROOT, "bench.py")

    def torchrun(n):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), bench, "--gpus", str(n)]
    env_gloo = dict(env, EIGEN_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    # the six launches are independent processes (tiny workloads sharing the one GPU): started three at a time, checked in order
    jobs = {
        "one": ([sys.executable, bench, "--gpus", "1"] + small[:-2] + ["--no-cpu-baseline"], env),   # (WITH the roofline block: the plain single-rank path the driver runs)
        # the multi-rank reporting path (RCCL group, per-rank device times out of the all-gather, leaving the group before rank 0's
        # untimed legs) in a group of ONE rank: what a single-GPU box can run of `--gpus N`
        "single_rank_group": ([sys.executable, bench, "--gpus", "1"] + small[:-2] + ["--no-cpu-baseline"], dict(env, EIGEN_DIST_SINGLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")),
        # TWO ranks for real (sharding, broadcast, all-gather with per-rank times, rank 1 leaving while rank 0 profiles) -- on a 1-GPU
        # box they share the device and gloo carries the collectives (RCCL refuses two ranks on one device): control flow, not a number
        "two_ranks": (torchrun(2) + small[:-2] + ["--no-cpu-baseline"], env_gloo),
        # EIGHT ranks at the HEADLINE shape: pop 256 -> 32 genomes per rank, engines at max_batch 32, the broadcast of
        # the 256-genome wire arrays, the all-gather, all ranks leaving the group together -- what the driver's `--gpus 8` run does,
        # executed once on a GPU box (the eight ranks share its one device over gloo: control flow, not a measurement)
        "eight_ranks": (torchrun(8) + ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-parity", "--no-supplementary"], env_gloo),
        "two_without_launcher": ([sys.executable, bench, "--gpus", "2"] + small, env),
        # joined under a launcher with the wrong world size: refuse
        "wrong_world_size": ([sys.executable, bench, "--gpus", "2"] + small, dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")),
    }
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=3) as pool:
        res = dict(zip(jobs, pool.map(lambda j: subprocess.run(j[0], env=j[1], capture_output=True, text=True, timeout=900), jobs.values())))

    def json_line(r):
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, "exactly ONE JSON line (rank 0's)"
        return json.loads(lines[0])
    line = json_line(res["one"])
    assert line["n_gpus"] == 1 and line["config"]["global_pop"] == 8 and line["scaling"] == "strong" and line["value"] > 0
    assert line["roofline"]["all_conv_kernels"]["launches"] > 0 and line["roofline"]["winograd_mask"] == "0x0FFFFFFE"
    line = json_line(res["single_rank_group"])
    assert "RCCL" in line["config"]["parallelism"] and len(line["multi_gpu"]["per_rank_device_ms"]) == 1 and line["multi_gpu"]["device_ms_max"] > 0
    assert line["multi_gpu"]["ranks_seen"] == [0] and line["multi_gpu"]["backend"] == "nccl" and line["multi_gpu"]["rccl_version"]
    assert line["multi_gpu"]["hsa_ipc_mode_legacy"] == "0"
    assert line["roofline"]["all_conv_kernels"]["launches"] > 0   # rank 0's roofline pass ran after the group was left
    line = json_line(res["two_ranks"])
    assert line["n_gpus"] == 2 and line["config"]["genomes_per_gpu"] == 4 and len(line["multi_gpu"]["per_rank_device_ms"]) == 2
    assert min(line["multi_gpu"]["per_rank_device_ms"]) > 0 and line["roofline"]["all_conv_kernels"]["launches"] > 0 and line["nonzero_fitness"] >= 0
    assert line["multi_gpu"]["ranks_seen"] == [0, 1] and line["multi_gpu"]["world_size"] == 2 and line["multi_gpu"]["backend"] == "gloo"
    line = json_line(res["eight_ranks"])
    assert line["n_gpus"] == 8 and line["config"]["global_pop"] == 256 and line["config"]["genomes_per_gpu"] == 32 and line["config"]["device_batch"] == 32
    assert line["multi_gpu"]["ranks_seen"] == list(range(8)) and len(line["multi_gpu"]["per_rank_device_ms"]) == 8
    assert line["scaling"] == "strong" and line["nonzero_fitness"] >= 100 and line["roofline"]["all_conv_kernels"]["launches"] > 0
    n = 2
    r = res["two_without_launcher"]
    if torch.cuda.device_count() >= n:
        line = json_line(r)
        assert line["n_gpus"] == n and line["config"]["genomes_per_gpu"] == 4 and "RCCL" in line["config"]["parallelism"]
        assert len(line["multi_gpu"]["per_rank_device_ms"]) == n
    else:
        assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    r = res["wrong_world_size"]
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
