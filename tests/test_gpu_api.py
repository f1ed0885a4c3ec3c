"""GPU: the drop-in Python API (fitness.py) end to end against the CPU oracle, including chunking and sharding."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from evolutionary_illusion_generator_amd import fitness, grids, synth, weights


def _oracle_fitness(pop, cfg, wts, ch, w, h, structure, pairing=0):
    from oracle import grids as ogrids, pipeline
    grid = ogrids.create_grid(structure, w, h, 10)  # the ORACLE's grid: the two sides share no code
    return np.array([pipeline.genome_fitness(g, cfg, grid, wts, ch, w, h, structure, pairing=pairing) for _, g in pop])


@pytest.mark.parametrize("structure,w,h,ch", [(2, 64, 64, [1, 16, 32, 64]), (0, 160, 120, [1, 8, 16, 32]), (1, 96, 64, [3, 12, 24, 48])])
def test_get_fitnesses_neat_drop_in(cuda, oracle_lib, tmp_path, structure, w, h, ch):
    c_dim = ch[0]
    cfg = synth.make_config(2, 3 if c_dim == 3 else 1)
    pop = synth.make_population(9, cfg, seed=31)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=2)
    ret = fitness.get_fitnesses_neat(structure, pop, wts, cfg, w, h, ch, c_dim=c_dim, best_dir=str(tmp_path), gradient=1)
    ref = _oracle_fitness(pop, cfg, wts, ch, w, h, structure)
    got = np.array([g.fitness for _, g in pop])
    assert all(isinstance(g.fitness, float) for _, g in pop)
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-12, equal_nan=True), (got, ref)
    assert np.array_equal(got, ret)
    assert (ref != 0).any()
    # best artefacts: the LAST maximal genome, white and black background renders
    from PIL import Image
    from oracle import pipeline
    best = max(range(len(pop)), key=lambda i: (got[i], i))
    from oracle import grids as ogrids
    grid = ogrids.create_grid(structure, w, h, 10)
    for name, bg in (("best.png", 1), ("best_black_bg.png", 0)):
        img = np.asarray(Image.open(tmp_path / name))
        exp = pipeline.render_chw(pop[best][1], cfg, grid, c_dim, w, h, bg=bg)
        exp = exp.transpose(1, 2, 0) if c_dim == 3 else exp[0]
        assert np.array_equal(img, exp), name
    assert Image.open(tmp_path / "best_flow.png").size == (w, h)
    if structure == 1:  # enhanced.png: the 800x800 3x3 + 2x2 circle grid (generate_illusion.py:665-671), byte-exact
        from oracle import cppn
        eg = grids.enhanced_image_grid(800, 800, structure)
        exp = cppn.render(eg, pop[best][1], cfg, c_dim, 800, 800)
        assert np.array_equal(np.asarray(Image.open(tmp_path / "enhanced.png")), exp)


def test_population_larger_than_device_batch_is_chunked(cuda, oracle_lib):
    w, h, ch = 64, 64, [1, 8, 16]
    cfg = synth.make_config(2, 1)
    pop = synth.make_population(11, cfg, seed=5)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=3)
    genomes = [g for _, g in pop]
    one = fitness.evaluate_population(2, genomes, wts, cfg, w, h, ch, c_dim=1, max_batch=16)
    chunked = fitness.evaluate_population(2, genomes, wts, cfg, w, h, ch, c_dim=1, max_batch=4)
    assert np.array_equal(one, chunked)
    assert np.allclose(one, _oracle_fitness(pop, cfg, wts, ch, w, h, 2), rtol=1e-9, atol=1e-12)


def test_get_vectors_and_calculate_fitness_single_image_api(cuda, oracle_lib, tmp_path):
    from PIL import Image
    from oracle import pipeline, scores
    w, h, ch, structure = 160, 120, [3, 8, 16, 32], 2
    cfg = synth.make_config(2, 3)
    pop = synth.make_population(3, cfg, seed=8)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=4)
    from oracle import grids as ogrids
    grid = ogrids.create_grid(structure, w, h, 10)
    for i, (_, g) in enumerate(pop):
        img = pipeline.render_chw(g, cfg, grid, 3, w, h)
        path = str(tmp_path / ("img%d.png" % i))
        Image.fromarray(img.transpose(1, 2, 0), "RGB").save(path, "PNG")
        v = fitness.get_vectors(path, wts, ch, w, h)
        ref_v = pipeline.image_vectors(img, wts, ch, w, h, pairing=pipeline.PAIR_SINGLE)  # original -> 2nd extension
        if len(ref_v) == 0:
            assert v == [None]
            continue
        assert isinstance(v, np.ndarray) and v.shape == ref_v.shape
        assert np.array_equal(v.astype(np.float32), ref_v)
        for st in (0, 1, 2, 3):
            got = fitness.calculate_fitness(st, v, path, w, h)
            assert got == pytest.approx(scores.fitness_from_vectors(st, ref_v.astype(np.float64), w, h), rel=1e-9, abs=1e-12)
    assert fitness.calculate_fitness(1, [None], "x.png", w, h) == 0.0
    with pytest.raises(NameError):
        fitness.calculate_fitness(7, np.zeros((3, 4)), "x.png", w, h)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)  # the test box has ONE GPU: both ranks share it, gloo carries the all-gather
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from evolutionary_illusion_generator_amd import fitness as F, synth as S, weights as W
    w, h, ch = 64, 64, [1, 8, 16]
    cfg = S.make_config(2, 1)
    pop = S.make_population(7, cfg, seed=12)
    wts = W.synthetic_prednet_weights(ch, w, h, seed=6)
    F.get_fitnesses_neat(2, pop, wts, cfg, w, h, ch, c_dim=1, best_dir=None)
    q.put((rank, [g.fitness for _, g in pop]))
    dist.destroy_process_group()


def test_two_ranks_shard_the_population(cuda, oracle_lib):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    w, h, ch = 64, 64, [1, 8, 16]
    cfg = synth.make_config(2, 1)
    pop = synth.make_population(7, cfg, seed=12)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=6)
    ref = _oracle_fitness(pop, cfg, wts, ch, w, h, 2)
    assert res[0] == res[1]
    assert np.allclose(res[0], ref, rtol=1e-9, atol=1e-12)


def test_abi_error_conventions(cuda):
    """Negative status + eigen_last_error text; nothing is computed on bad input."""
    import torch
    from evolutionary_illusion_generator_amd import genome as gm
    from evolutionary_illusion_generator_amd.engine import Engine, EngineError
    with pytest.raises(EngineError, match="divisible"):
        Engine(66, 64, [1, 4, 8], 2)                       # 2x2 pooling per layer needs W % 4 == 0 here
    with pytest.raises(EngineError, match="c_dim"):
        Engine(64, 64, [2, 4], 2)
    e = Engine(32, 32, [1, 4, 8], 2)
    img = torch.zeros((2, 1, 32, 32), dtype=torch.uint8, device=cuda)
    fr = torch.zeros((2, 2, 1, 32, 32), dtype=torch.uint8, device=cuda)
    with pytest.raises(EngineError, match="eigen_set_prednet_weights"):
        e.prednet_rollout(img, 2, 21, 19, fr)
    cfg = synth.make_config(2, 1)
    gb = gm.GenomeBatch([g for _, g in synth.make_population(2, cfg)], cfg, 1)
    with pytest.raises(EngineError, match="eigen_set_grid"):
        e.render_cppn(gb, img)
    e.set_weights(weights.synthetic_prednet_weights([1, 4, 8], 32, 32))
    with pytest.raises(EngineError, match="max_batch"):
        e.prednet_rollout(img, 3, 21, 19, fr)
    with pytest.raises(EngineError, match="n_steps"):
        e.prednet_rollout(img, 2, 23, 19, fr)
    g = grids.create_grid(2, 32, 32, 10)
    e.set_grid([g["x_mat"], g["y_mat"]])
    gb.edge_src[0] = 10 ** 6                                # corrupt program: must be rejected on the host
    with pytest.raises(EngineError, match="topologically"):
        e.render_cppn(gb, img)
    d = torch.zeros(2, dtype=torch.float64, device=cuda)
    with pytest.raises(EngineError, match="structure"):
        e.score(9, torch.zeros((2, e.K, 4), device=cuda), torch.zeros(2, dtype=torch.int32, device=cuda), 2, d)


def test_fast_path_reproduces_the_fitness_assigned_by_the_references_glue(cuda):
    """The fixture was produced by /root/reference's unmodified get_fitnesses_neat (PNG files, frame index arithmetic,
    scoring) with its absent dependencies backed by the CPU oracle (tests/golden/make_golden.py --e2e).  The HIP path
    -- one batched device pass, no files -- must assign the same fitness to the same genomes."""
    import json
    from test_oracle_golden import GOLD, _genomes_from_fixture
    runs = json.load(open(os.path.join(GOLD, "e2e_reference_glue.json")))["runs"]
    for run in runs:
        cfg, pop = _genomes_from_fixture(run)
        w, h, ch, st = run["w"], run["h"], run["channels"], run["structure"]
        wts = weights.synthetic_prednet_weights(ch, w, h, seed=run["weights_seed"])
        fitness.get_fitnesses_neat(st, pop, wts, cfg, w, h, ch, c_dim=run["c_dim"], best_dir=None, gradient=1)
        got = np.array([g.fitness for _, g in pop])
        assert np.allclose(got, run["fitness"], rtol=1e-9, atol=1e-12), (st, got, run["fitness"])


def test_single_image_api_reproduces_the_references_glue(cuda):
    """fitness_calculator.get_vectors + calculate_fitness of the reference, run unmodified over oracle-backed
    dependencies (fixture 'single'), against the drop-in single-image API on the device."""
    import json
    from test_oracle_golden import GOLD
    for case in json.load(open(os.path.join(GOLD, "e2e_reference_glue.json")))["single"]:
        img = np.asarray(case["image"], dtype=np.uint8)
        w, h, ch = case["w"], case["h"], case["channels"]
        wts = weights.synthetic_prednet_weights(ch, w, h, seed=case["weights_seed"])
        v = fitness.get_vectors(img, wts, ch, w, h)
        assert np.array_equal(np.asarray(v, dtype=np.float64), np.asarray(case["vectors"]))
        if case["fitness"] != "UnboundLocalError":
            got = fitness.calculate_fitness(case["structure"], v, None, w, h)
            assert abs(got - case["fitness"]) <= 1e-9 * abs(case["fitness"])


_FRAMES_SCRIPT = r"""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from evolutionary_illusion_generator_amd import weights
from evolutionary_illusion_generator_amd.engine import Engine
out = []
# (80 x 64: 20 x 16 maps, 4-wide strips; 160 x 120 gray: Winograd ConvLSTMs on 80 x 60 / 40 x 30 / 20 x 15 maps -- tall blocks, packed tiles)
for (w, h, ch) in [(64, 64, [3, 12, 24, 48]), (96, 64, [1, 8, 16]), (80, 64, [1, 4, 8]), (160, 120, [1, 16, 32, 48])]:
    rng = np.random.default_rng(7)
    B = 3
    img = rng.integers(0, 256, (B, ch[0], h, w), dtype=np.uint8)
    img[:, :, ::7] //= 3
    e = Engine(w, h, ch, B, n_repeat=4, n_ext=2)
    e.set_weights(weights.synthetic_prednet_weights(ch, w, h, seed=3))
    d = torch.from_numpy(img).cuda()
    fr = torch.zeros((B, 6, ch[0], h, w), dtype=torch.uint8, device="cuda")
    e.prednet_rollout(d, B, 6, 0, fr)
    torch.cuda.synchronize()
    out.append(hashlib.sha256(fr.cpu().numpy().tobytes()).hexdigest())
    for rep in range(2):   # the same buffers again: a roll-out leaves nothing behind that changes the next one (reset_state)
        fr.zero_()
        e.prednet_rollout(d, B, 6, 0, fr)
        torch.cuda.synchronize()
        assert hashlib.sha256(fr.cpu().numpy().tobytes()).hexdigest() == out[-1], "roll-out %%d of the same buffers differs" %% (rep + 2)
print("FRAMES", *out)
"""


def test_specialised_operators_equal_the_general_mfma_path(cuda):
    """Every A/B switch of the engine that must NOT change a bit (step-0 operators, single-K-block ConvA, one-block 2x2 pass, the two direct image-layer
    kernels, the eight-wave direct ConvA, the tile order, how many N-blocks of a tile a block of the F(4x4) kernel walks, its block shape) set one at a time in a fresh process: all
    six PredNet frames of three small roll-outs are byte-identical."""
    import subprocess
    script = _FRAMES_SCRIPT % {"root": ROOT}

    def run(extra_env):
        env = dict(os.environ)
        env.update(extra_env)
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("FRAMES")]
        assert line, r.stdout[-2000:]
        return line[0]

    base = run({})
    # (round 5: the side stream, the two-streams-per-population pipeline and the schedule variants of the eight-wave Winograd kernel lost their A/Bs; round 6: the
    #  half-block / 4-column-strip / in-kernel-2x2-chain instantiations of the direct kernel and the eight-wave Winograd kernel went the same way.)
    envs = [{sw: "1"} for sw in ("EIGEN_NO_T0", "EIGEN_NO_ONEKB", "EIGEN_NO_UP4C", "EIGEN_LSTM0_MFMA", "EIGEN_CONVP0_MFMA")]
    envs += [{"EIGEN_W8": "0"}, {"EIGEN_W8": "1"}, {"EIGEN_TILE_MAP": "0"}, {"EIGEN_TILE_MAP": "1"},
             {"EIGEN_W4_PARTS": "1"}, {"EIGEN_W4_PARTS": "2"}, {"EIGEN_W4_PARTS": "99"}, {"EIGEN_W4_TALL": "0"}, {"EIGEN_W4_TALL": "1"}, {"EIGEN_W4_TALL": "0", "EIGEN_W4_HALF": "1"}, {"EIGEN_W4_HALF": "0"}, {"EIGEN_W4_PACK": "0"}, {"EIGEN_W4_PACK": "1"}, {"EIGEN_SIDE_STREAM": "1"}, {"EIGEN_SIDE_STREAM": "0"}]   # (ConvP_l of layers > 0 on a side stream: the default at these sizes / off)   # the walk of the F(4x4) kernel: all N-blocks of a tile in one block ... one block per N-block
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as pool:   # (fresh processes sharing the one GPU: the runs are tiny)
        for env, got in zip(envs, pool.map(run, envs)):
            assert got == base, env
