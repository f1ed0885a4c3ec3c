"""Import shims under the reference's dotted module names: evolutionary_illusion_generator_amd/compat.

CPU tests: the names resolve, the unchanged reference file imports against them (only where /root/reference exists: this
container), and nothing falls back to the CPU.  GPU tests: every shim entry point against the oracle, and the reference's
file-based call sequence replayed through the shims against the fixture its unmodified glue produced."""
import json
import os
import sys

import numpy as np
import pytest

from evolutionary_illusion_generator_amd import compat, weights

REF = "/root/reference"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def shims():
    saved_path, saved_mods = list(sys.path), set(sys.modules)
    compat.install("all")
    yield
    sys.path[:] = saved_path
    for m in set(sys.modules) - saved_mods:
        if m.split(".")[0] in ("chainer_prednet", "optical_flow", "pytorch_neat", "neat", "cv2", "google", "generate_illusion", "fitness_calculator"):
            del sys.modules[m]


def test_every_dotted_name_the_reference_imports_resolves(shims):
    """generate_illusion.py:1-21 / fitness_calculator.py:1-5, name by name."""
    from chainer_prednet.PredNet.call_prednet import test_prednet
    from chainer_prednet.utilities.mirror_images import mirror, mirror_multiple, TransformationType
    from optical_flow.optical_flow import lucas_kanade, draw_tracks, save_data
    from pytorch_neat.pytorch_neat.cppn import create_cppn
    from pytorch_neat.pytorch_neat.multi_env_eval import MultiEnvEvaluator
    from pytorch_neat.pytorch_neat.neat_reporter import LogReporter
    from pytorch_neat.pytorch_neat.recurrent_net import RecurrentNet
    from google.colab.patches import cv2_imshow
    import cv2
    import neat
    for f in (test_prednet, lucas_kanade, create_cppn, mirror, mirror_multiple, draw_tracks, save_data, cv2_imshow):
        assert callable(f)
    assert sys.modules[test_prednet.__module__].__file__.startswith(compat.SHIM_DIR)
    assert {"Config", "DefaultGenome", "DefaultReproduction", "DefaultSpeciesSet", "DefaultStagnation", "Population",
            "StdOutReporter", "StatisticsReporter", "Checkpointer"} <= set(dir(neat))
    assert int(TransformationType.MirrorH) == 1
    rgb = np.arange(24, dtype=np.uint8).reshape(2, 4, 3)
    assert np.array_equal(cv2.cvtColor(rgb, cv2.COLOR_RGB2BGR), rgb[:, :, ::-1])
    assert cv2.cvtColor(rgb[:, :, 0], cv2.COLOR_GRAY2BGR).shape == (2, 4, 3)
    with pytest.raises(NotImplementedError):
        MultiEnvEvaluator()


def test_install_prefers_real_packages_and_adds_only_what_is_missing():
    saved = list(sys.path)
    try:
        added = compat.install("missing")
        assert sys.path[0] == compat.SHIM_DIR
        assert "neat" in added  # neat-python is not installed in this image
        for name in added:  # stand-ins go to the END of sys.path: a real installation always wins
            assert sys.path.index(os.path.join(compat.OPTIONAL_DIR, compat.OPTIONAL[name])) > sys.path.index(compat.SHIM_DIR)
        import numpy
        assert compat._missing("numpy") is False
    finally:
        sys.path[:] = saved


def test_mirror_and_flow_drawing_helpers(shims, tmp_path):
    from PIL import Image
    from chainer_prednet.utilities.mirror_images import mirror, TransformationType
    from optical_flow.optical_flow import draw_tracks, save_data
    a = np.arange(12, dtype=np.uint8).reshape(3, 4)
    Image.fromarray(a).save(tmp_path / "a.png")
    out = mirror(str(tmp_path / "a.png"), str(tmp_path / "m"), TransformationType.MirrorH)
    assert np.array_equal(np.asarray(Image.open(out)), a[:, ::-1])
    img = draw_tracks(np.zeros((1, 16, 16), np.uint8), [[4.0, 4.0, 0.2, 0.0]])
    assert img.mode == "RGB" and np.asarray(img).any()
    save_data([[1, 2, 3, 4]], str(tmp_path / "v" / "v.csv"))
    assert open(tmp_path / "v" / "v.csv").read().splitlines() == ["x,y,dx,dy", "1.0,2.0,3.0,4.0"]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "generate_illusion.py")), reason="reference checkout only exists in the build container")
def test_unchanged_reference_module_imports_against_the_shims_and_has_no_cpu_fallback(shims):
    import torch
    sys.path.append(REF)
    import generate_illusion as gi
    assert gi.__file__.startswith(REF)
    assert gi.test_prednet.__module__ == "chainer_prednet.PredNet.call_prednet"
    assert gi.lucas_kanade.__module__ == "optical_flow.optical_flow" and gi.create_cppn.__module__ == "pytorch_neat.pytorch_neat.cppn"
    from evolutionary_illusion_generator_amd import fitness, synth
    from evolutionary_illusion_generator_amd.engine import EngineError
    if not torch.cuda.is_available():
        cfg = synth.make_config(2, 1)
        g = synth.make_population(1, cfg, seed=0)[0][1]
        grid = gi.create_grid(gi.StructureType.Free, 16, 8, 10)
        with pytest.raises(EngineError):  # the reference's own renderer reaches the device call and fails loudly
            gi.get_image_from_cppn(grid, g, 1, 16, 8, cfg)
    compat.use_fast_path(gi)
    assert gi.get_fitnesses_neat is fitness.get_fitnesses_neat and gi.get_vectors is fitness.get_vectors


# ------------------------------------------------------------------------------------------------ GPU
def _fixture_run(i):
    from test_oracle_golden import _genomes_from_fixture
    run = json.load(open(os.path.join(GOLD, "e2e_reference_glue.json")))["runs"][i]
    cfg, pop = _genomes_from_fixture(run)
    return run, cfg, pop


@pytest.mark.gpu
def test_create_cppn_shim_node_calls_equal_the_oracle(cuda, shims, oracle_lib):
    import torch
    from pytorch_neat.pytorch_neat.cppn import create_cppn
    from evolutionary_illusion_generator_amd import grids
    from oracle import cppn as ocppn
    run, cfg, pop = _fixture_run(1)  # colour, Circles, 160x120
    grid = grids.create_grid(run["structure"], run["w"], run["h"], 10)
    x = torch.tensor(grid["x_mat"].flatten())
    y = torch.tensor(grid["y_mat"].flatten())
    for _, g in pop:
        nodes = create_cppn(g, cfg, ["x", "y"], [])
        assert len(nodes) == 3
        ref = ocppn.render_planes(g, cfg, [grid["x_mat"].reshape(-1), grid["y_mat"].reshape(-1)])
        for c, node in enumerate(nodes):
            got = node(x=x, y=y)
            assert got.dtype == torch.float64 and got.shape == x.shape
            assert np.array_equal(got.numpy(), np.asarray(ref[c]), equal_nan=True)
    with pytest.raises(AssertionError):
        create_cppn(pop[0][1], cfg, ["x", "y", "r"], [])


@pytest.mark.gpu
def test_file_based_shims_reproduce_the_reference_glue_fixture(cuda, shims, oracle_lib, tmp_path):
    """The reference's call sequence (generate_illusion.py:501-556) replayed through the shims with its file names:
    images/%010d.png -> test_prednet(sequence of 20x repeated paths) -> prediction/%010d.png, %010d_extended.png ->
    lucas_kanade(prediction_0, prediction_1).  Frames and vectors must equal the oracle's, and the fitness computed
    from the vectors must equal what the reference's unmodified glue assigned (tests/golden/e2e_reference_glue.json)."""
    import oracle
    from PIL import Image
    from chainer_prednet.PredNet.call_prednet import test_prednet
    from optical_flow.optical_flow import lucas_kanade
    from evolutionary_illusion_generator_amd import fitness
    from oracle import scores
    for ri in (0, 1):
        run, cfg, pop = _fixture_run(ri)
        w, h, ch, st, c_dim = run["w"], run["h"], run["channels"], run["structure"], run["c_dim"]
        wts = weights.synthetic_prednet_weights(ch, w, h, seed=run["weights_seed"])
        out = str(tmp_path / ("run%d" % ri)) + "/"
        os.makedirs(out + "images")
        imgs = fitness.render_images(st, [g for _, g in pop], wts, cfg, w, h, ch, c_dim=c_dim)
        repeat, ext = 20, 2
        names, seq = [], [None] * (len(pop) + repeat)  # over-allocated like generate_illusion.py:499
        seq = [None] * (len(pop) * repeat + repeat)
        for i, im in enumerate(imgs):
            name = out + "images/" + str(i).zfill(10) + ".png"
            Image.fromarray(im[0] if c_dim == 1 else im.transpose(1, 2, 0)).save(name, "PNG")
            names.append(name)
            seq[i * repeat:(i + 1) * repeat] = [name] * repeat
        pred = out + "/prediction/"
        test_prednet(initmodel=wts, sequence_list=[seq], size=[w, h], channels=ch, gpu=0, output_dir=pred, skip_save_frames=1,
                     extension_start=repeat, extension_duration=ext, reset_at=repeat + ext, verbose=0, c_dim=c_dim)
        assert len(os.listdir(pred)) == len(pop) * (repeat + ext)
        got_fit = []
        for i in range(len(pop)):
            i0 = i * repeat + repeat - 1
            p0, p1 = pred + str(i0).zfill(10) + ".png", pred + str(i0 + ext - 1).zfill(10) + "_extended.png"
            if i == 0:
                fr = oracle.prednet_rollout(wts, ch, w, h, imgs[0], n_repeat=repeat, n_ext=ext)
                for t, path in ((0, pred + "0".zfill(10) + ".png"), (19, p0), (20, p1), (21, pred + str(i0 + 2).zfill(10) + "_extended.png")):
                    a = np.asarray(Image.open(path))
                    assert np.array_equal(a[None] if a.ndim == 2 else a.transpose(2, 0, 1), fr[t]), path
            res = lucas_kanade(p0, p1, out + "/flow/", save=True, verbose=0, save_name=out + "/images/" + str(i).zfill(10) + "_f.png")
            assert os.path.exists(out + "/images/" + str(i).zfill(10) + "_f.png")
            v = np.asarray(res["vectors"], dtype=np.float64) if res["vectors"] else np.zeros((0, 4))
            got_fit.append(scores.fitness_from_vectors(st, v, w, h))
        assert got_fit == run["fitness"], (got_fit, run["fitness"])


@pytest.mark.gpu
def test_test_prednet_shim_rejects_what_it_cannot_honour(cuda, shims, tmp_path):
    from PIL import Image
    from chainer_prednet.PredNet.call_prednet import test_prednet
    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    Image.fromarray(np.zeros((16, 16), np.uint8)).save(a)
    Image.fromarray(np.ones((16, 16), np.uint8)).save(b)
    wts = weights.synthetic_prednet_weights([1, 2], 16, 16, seed=0)
    kw = dict(initmodel=wts, size=[16, 16], channels=[1, 2], gpu=0, output_dir=str(tmp_path / "o"), skip_save_frames=1,
              extension_start=2, extension_duration=1, verbose=0, c_dim=1)
    with pytest.raises(NotImplementedError):
        test_prednet(sequence_list=[[a, b]], reset_at=3, **kw)       # a run of different frames
    with pytest.raises(NotImplementedError):
        test_prednet(sequence_list=[[a, a]], reset_at=7, **kw)       # state carried across stimuli
    test_prednet(sequence_list=[[a, a, b, b]], reset_at=3, **kw)
    assert sorted(os.listdir(tmp_path / "o")) == ["0000000000.png", "0000000001.png", "0000000002.png", "0000000002_extended.png",
                                                  "0000000003.png", "0000000004_extended.png"]


# ------------------------------------------------------------ the unedited reference driver on the shims (CPU, build container)
def _oracle_stage_doubles(monkeypatch, wts_holder):
    """TEST-ONLY: the three stage-level device calls the shims funnel into, answered by the CPU oracle, so that the
    reference's unedited Python (and the shims' file handling) can be executed where there is no GPU."""
    import oracle
    from evolutionary_illusion_generator_amd import fitness
    from oracle import cppn as ocppn

    def cppn_node_planes(genome, config, planes, n_outputs=None):
        out = ocppn.render_planes(genome, config, [np.asarray(p, dtype=np.float64).reshape(-1) for p in planes])
        return np.stack([np.asarray(o, dtype=np.float64) for o in out])

    def prednet_predictions(images, model_name, channels, w, h, n_repeat=20, n_ext=2):
        wts = wts_holder["weights"]
        return np.stack([oracle.prednet_rollout(wts, channels, w, h, im, n_repeat=n_repeat, n_ext=n_ext) for im in images])

    monkeypatch.setattr(fitness, "cppn_node_planes", cppn_node_planes)
    monkeypatch.setattr(fitness, "prednet_predictions", prednet_predictions)
    monkeypatch.setattr(fitness, "flow_vectors", lambda a, b: oracle.lucas_kanade(a, b))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "generate_illusion.py")), reason="reference checkout only exists in the build container")
def test_unedited_reference_get_fitnesses_neat_on_the_shims_assigns_the_fixture_fitness(shims, oracle_lib, monkeypatch, tmp_path):
    """generate_illusion.get_fitnesses_neat, unedited, importing the shims as its submodules.  With the device calls
    answered by the oracle (no GPU here) the shims' file handling + the reference's glue must assign the same fitness
    as the fixture (made with throw-away fakes instead of the shims) -- and as the HIP fast path does on the GPU
    (tests/test_gpu_api.py)."""
    sys.path.append(REF)
    import generate_illusion as gi
    holder = {}
    _oracle_stage_doubles(monkeypatch, holder)
    run, cfg, pop = _fixture_run(0)
    w, h, ch = run["w"], run["h"], run["channels"]
    holder["weights"] = weights.synthetic_prednet_weights(ch, w, h, seed=run["weights_seed"])
    monkeypatch.chdir(tmp_path)
    os.makedirs(tmp_path / "best")  # neat_illusion creates best_dir before the first generation (generate_illusion.py:683-685)
    gi.get_fitnesses_neat(gi.StructureType(run["structure"]), pop, "model.npz", cfg, w, h, ch, c_dim=run["c_dim"],
                          best_dir=str(tmp_path / "best"), gradient=1)
    assert [g.fitness for _, g in pop] == run["fitness"]
    for name in ("best.png", "best_flow.png", "best_black_bg.png"):
        assert (tmp_path / "best" / name).exists()


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "generate_illusion.py")), reason="reference checkout only exists in the build container")
def test_unedited_neat_illusion_driver_evolves_on_the_shims(shims, oracle_lib, monkeypatch, tmp_path):
    """The whole unedited driver: neat.Config on the reference's own neat_configs/circles_bw.txt, neat.Population,
    reporters, checkpointer (`neat` = neat_lite stand-in), eval_genomes -> get_fitnesses_neat for two generations."""
    sys.path.append(REF)
    import generate_illusion as gi
    holder = {"weights": weights.synthetic_prednet_weights([1, 4, 8], 64, 48, seed=3)}
    _oracle_stage_doubles(monkeypatch, holder)
    calls = []
    inner = gi.get_fitnesses_neat

    class _Stop(Exception):
        pass

    def counted(structure, genomes, *a, **kw):
        inner(structure, genomes, *a, **kw)
        calls.append([g.fitness for _, g in genomes])
        if len(calls) == 2:
            raise _Stop()

    monkeypatch.setattr(gi, "get_fitnesses_neat", counted)
    monkeypatch.chdir(tmp_path)
    with pytest.raises(_Stop):
        gi.neat_illusion(str(tmp_path / "out"), "model.npz", os.path.join(REF, "neat_configs", "circles_bw.txt"),
                         gi.StructureType.Free, 64, 48, [1, 4, 8], c_dim=1, gradient=1)
    assert len(calls) == 2 and all(len(c) >= 5 and all(isinstance(f, float) for f in c) for c in calls)  # generation 1 is refilled to min_species_size
    assert (tmp_path / "out" / "enhanced.png").exists()
