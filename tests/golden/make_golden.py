#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference (only possible in the build container).

Run:  python tests/golden/make_golden.py        (needs /root/reference; writes tests/golden/*.npz|json)

The reference modules import packages that are absent here (empty submodules chainer_prednet /
optical_flow / pytorch_neat, plus cv2, neat, google.colab).  They are replaced by inert stubs in
sys.modules so that the reference's OWN pure-python/numpy functions can be executed:
  * fitness_calculator.py:18-215  scorers (plausibility_ratio, strength_number,
    horizontal_symmetry_score, swarm_score, rotation_symmetry_score) and :505-548 calculate_fitness
  * generate_illusion.py:38-117,196-317 fill_circle / create_grid
  * generate_illusion.py:372-460 get_image_from_cppn (with an injected fake create_cppn whose
    output planes are seeded arrays stored in the fixture)
  * generate_illusion.py:478-673 get_fitnesses_neat score combination, with test_prednet stubbed
    out and lucas_kanade returning seeded vectors stored in the fixture.
Only inputs and expected outputs are stored -- no reference source text.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs(state):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Enum:  # TransformationType placeholder
        pass

    mod("chainer_prednet"); mod("chainer_prednet.PredNet"); mod("chainer_prednet.utilities")
    mod("chainer_prednet.PredNet.call_prednet", test_prednet=lambda **kw: state.setdefault("prednet_calls", []).append(kw))
    mod("chainer_prednet.utilities.mirror_images", mirror=None, mirror_multiple=None, TransformationType=_Enum)
    mod("cv2", imread=lambda p: None, cvtColor=None, COLOR_RGB2BGR=0, COLOR_GRAY2BGR=1)
    mod("google"); mod("google.colab"); mod("google.colab.patches", cv2_imshow=lambda img: None)
    mod("neat")

    def lucas_kanade(f0, f1, out_dir, save=True, verbose=0, save_name=""):
        state.setdefault("lk_calls", []).append((f0, f1, save_name))
        i = len(state["lk_calls"]) - 1
        if save_name:  # the reference later copies this file (generate_illusion.py:655-657)
            os.makedirs(os.path.dirname(save_name) or ".", exist_ok=True)
            open(save_name, "wb").close()
        v = state["lk_vectors"][i]
        return {"vectors": [list(map(float, r)) for r in v]}

    mod("optical_flow"); mod("optical_flow.optical_flow", lucas_kanade=lucas_kanade, draw_tracks=None, save_data=None)
    mod("pytorch_neat"); mod("pytorch_neat.pytorch_neat")

    def create_cppn(genome, config, leaf_names, node_names):
        planes = state["planes"][genome.key]

        def mk(c):
            def f(x=None, y=None):
                import torch
                if x is not None and len(x) != planes.shape[1]:  # 800x800 'enhanced' render: content irrelevant
                    return torch.zeros(len(x), dtype=torch.float64)
                return torch.tensor(planes[c])
            return f
        return [mk(c) for c in range(len(planes))]

    mod("pytorch_neat.pytorch_neat.cppn", create_cppn=create_cppn)
    mod("pytorch_neat.pytorch_neat.multi_env_eval", MultiEnvEvaluator=None)
    mod("pytorch_neat.pytorch_neat.neat_reporter", LogReporter=None)
    mod("pytorch_neat.pytorch_neat.recurrent_net", RecurrentNet=None)


class FakeGenome:
    def __init__(self, key):
        self.key = key
        self.fitness = None


def vec_set(rng, n, w, h, mag):
    v = np.zeros((n, 4))
    v[:, 0] = rng.integers(1, w - 1, n)
    v[:, 1] = rng.integers(1, h - 1, n)
    v[:, 2:] = rng.normal(0, mag, (n, 2))
    return v


def main():
    state = {}
    install_stubs(state)
    sys.path.insert(0, REF)
    import fitness_calculator as fc
    import generate_illusion as gi

    rng = np.random.default_rng(20260928)

    # ---------------- grids (a1) ----------------
    grids = {}
    for name, st, w, h in [("circles_64x64", gi.StructureType.Circles, 64, 64),
                           ("circles_160x120", gi.StructureType.Circles, 160, 120),
                           ("circles_96x64", gi.StructureType.Circles, 96, 64),
                           ("circlesfree_64x64", gi.StructureType.CirclesFree, 64, 64),
                           ("circlesfree_80x60", gi.StructureType.CirclesFree, 80, 60),
                           ("free_64x64", gi.StructureType.Free, 64, 64),
                           ("free_160x120", gi.StructureType.Free, 160, 120),
                           ("bands_160x120", gi.StructureType.Bands, 160, 120),
                           ("bands_80x40", gi.StructureType.Bands, 80, 40)]:
        g = gi.create_grid(st, w, h, 10)
        grids[name + "_x"] = np.asarray(g["x_mat"], dtype=np.float64)
        grids[name + "_y"] = np.asarray(g["y_mat"], dtype=np.float64)
    eg = gi.enhanced_image_grid(120, 120, gi.StructureType.Circles)
    grids["enhanced_circles_120x120_x"] = eg["x_mat"]
    grids["enhanced_circles_120x120_y"] = eg["y_mat"]
    np.savez_compressed(os.path.join(OUT, "grids.npz"), **grids)

    # ---------------- image post-processing (a2) ----------------
    post = {}
    w, h = 48, 32
    grid = gi.create_grid(gi.StructureType.Circles, w, h, 10)
    post["grid_x"] = grid["x_mat"]; post["grid_y"] = grid["y_mat"]
    cases = []
    for ci, (c_dim, gradient, bg) in enumerate([(3, 1, 1), (3, 1, 0), (3, 0, 1), (1, 1, 1), (1, 0, 1), (1, 0, 0)]):
        nplanes = 3 if c_dim == 3 else 1
        planes = rng.normal(0.5, 0.9, (nplanes, w * h))
        # sprinkle exact / edge values: wrap-around, negatives, >1, halves, nan, inf
        edge = np.array([0.0, 1.0, 1.004, -0.5, 2.0, 0.5, 0.25, 0.75, 1.0 / 255, 254.5 / 255, np.nan, np.inf, -np.inf, 1e9, -1e9, 3.999 / 4,
                         3000000123.0 / 255, -3000000123.0 / 255, 5000000321.0 / 255, 2147483000.0 / 255, -2147483900.0 / 255,
                         1e19, -1e19, 300.7 / 255, -300.7 / 255, 65535.9 / 255, 1.5, 2.5, 0.5000001, 1e-320])
        fg = np.flatnonzero(np.asarray(grid["x_mat"]).reshape(-1) != -1)  # probes must sit on non-background pixels
        planes[:, fg[:edge.size]] = edge
        g = FakeGenome(1000 + ci)
        state.setdefault("planes", {})[g.key] = planes
        img = gi.get_image_from_cppn(grid, g, c_dim, w, h, None, bg=bg, gradient=gradient)
        arr = np.asarray(img)
        post["case%d_planes" % ci] = planes
        post["case%d_img" % ci] = arr
        cases.append({"c_dim": c_dim, "gradient": gradient, "bg": bg})
    np.savez_compressed(os.path.join(OUT, "postprocess.npz"), **post)
    with open(os.path.join(OUT, "postprocess.json"), "w") as f:
        json.dump({"w": w, "h": h, "cases": cases}, f, indent=1)

    # ---------------- scorers (a7-a13) ----------------
    sc = {"cases": []}
    w, h = 160, 120
    sets = []
    for n, mag in [(1, 0.1), (2, 0.1), (5, 0.05), (24, 0.1), (25, 0.1), (26, 0.12), (40, 0.08), (75, 0.15), (100, 0.2), (60, 0.01)]:
        sets.append(vec_set(rng, n, w, h, mag))
    # a structured rotating field (high rotation symmetry), a centred vector (distance 0), duplicates
    ang = np.linspace(0, 2 * np.pi, 36, endpoint=False)
    rot = np.stack([80 + 40 * np.cos(ang), 60 + 40 * np.sin(ang), -0.1 * np.sin(ang), 0.1 * np.cos(ang)], 1)
    sets.append(rot)
    sets.append(np.concatenate([rot, [[80.0, 60.0, 0.05, 0.02]], rot[:3]], 0))
    sets.append(np.array([[0.0, 0.0, -1000.0, 0.0]]))  # the "no vectors" sentinel, generate_illusion.py:554
    for v in sets:
        case = {"vectors": v.tolist()}
        for lim in (0.15, 0.3, 0.4):
            ratio, good = fc.plausibility_ratio(v, lim)
            case["plaus_%g" % lim] = [ratio, len(good)]
            if len(good) > 0:
                gv = np.asarray(good)
                case["strength_%g" % lim] = float(fc.strength_number(good, lim))
                case["hsym_%g" % lim] = float(fc.horizontal_symmetry_score(good, [0, h / 4 * 2]))
                case["rot_%g" % lim] = float(fc.rotation_symmetry_score(good, w, h, [0, h / 2]))
                case["swarm_%g" % lim] = float(fc.swarm_score(good))
        for st in (fc.StructureType.Bands, fc.StructureType.Circles, fc.StructureType.CirclesFree, fc.StructureType.Free):
            try:
                case["fitness_%d" % int(st)] = float(fc.calculate_fitness(st, v, "x.png", w, h))
            except UnboundLocalError:
                case["fitness_%d" % int(st)] = "UnboundLocalError"
        sc["cases"].append(case)
    sc["w"] = w; sc["h"] = h
    with open(os.path.join(OUT, "scores.json"), "w") as f:
        json.dump(sc, f)

    # ---------------- get_fitnesses_neat orchestration (a13, B1) ----------------
    orch = {"runs": []}
    cwd = os.getcwd()
    for st, c_dim in [(gi.StructureType.Circles, 3), (gi.StructureType.Free, 1), (gi.StructureType.CirclesFree, 1)]:
        w, h = 40, 32
        pop = [(100 + i, FakeGenome(100 + i)) for i in range(6)]
        planes = {}
        for _, g in pop:
            planes[g.key] = rng.uniform(0, 1, (c_dim, w * h))
        state["planes"] = planes
        nvec = [30, 0, 26, 5, 40, 25]
        lkv = [vec_set(rng, n, w, h, 0.1) if n else np.zeros((0, 4)) for n in nvec]
        state["lk_vectors"] = lkv
        state["lk_calls"] = []; state["prednet_calls"] = []
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                gi.get_fitnesses_neat(st, pop, "model.npz", None, w, h, [c_dim, 4, 8, 16], c_dim=c_dim, best_dir=td, gradient=1)
            finally:
                os.chdir(cwd)
        kw = state["prednet_calls"][0]
        orch["runs"].append({
            "structure": int(st), "c_dim": c_dim, "w": w, "h": h,
            "lk_vectors": [v.tolist() for v in lkv],
            "fitness": [float(g.fitness) for _, g in pop],
            "lk_files": [[os.path.basename(a), os.path.basename(b)] for a, b, _ in state["lk_calls"]],
            "prednet_kwargs": {k: kw[k] for k in ("size", "channels", "skip_save_frames", "extension_start", "extension_duration", "reset_at", "c_dim")},
            "sequence_len": len(kw["sequence_list"][0]),
        })
    with open(os.path.join(OUT, "orchestration.json"), "w") as f:
        json.dump(orch, f)
    print("golden fixtures written to", OUT)


if __name__ == "__main__" and "--e2e" not in sys.argv and "--round2" not in sys.argv:
    main()


# ----------------------------------------------------------------------------------------------------------------
# End-to-end fixture: the reference's OWN get_fitnesses_neat (glue code, PNG round trips, frame index arithmetic,
# scoring) executed unmodified, with its three absent dependencies substituted by the oracle's restatements:
#   pytorch_neat.cppn.create_cppn  -> oracle.cppn.render_planes      (torch float64 tensors in / out)
#   chainer_prednet test_prednet   -> oracle.prednet_rollout          (reads / writes the PNG files the caller names)
#   optical_flow.lucas_kanade      -> oracle.lucas_kanade             (reads the two PNG files)
# The fixture stores the genomes, the run parameters and the fitness values the reference assigned.
def make_e2e():
    import json as _json
    import tempfile as _tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))  # repo root: oracle/, evolutionary_illusion_generator_amd/
    import torch
    from PIL import Image
    import oracle
    from oracle import cppn as ocppn
    from evolutionary_illusion_generator_amd import synth, weights

    state = {}
    install_stubs(state)
    holder = {}

    def create_cppn(genome, config, leaf_names, node_names):
        def mk(c):
            def f(x=None, y=None):
                planes = ocppn.render_planes(genome, config, [x.numpy(), y.numpy()])
                return torch.as_tensor(np.broadcast_to(np.asarray(planes[c]), x.shape).copy())
            return f
        return [mk(c) for c in range(len(config.genome_config.output_keys))]

    def read_chw(path, c_dim):
        a = np.asarray(Image.open(path).convert("L" if c_dim == 1 else "RGB"))
        return np.ascontiguousarray(a[None] if a.ndim == 2 else a.transpose(2, 0, 1))

    def write_chw(img, path):
        Image.fromarray(img[0] if img.shape[0] == 1 else img.transpose(1, 2, 0)).save(path)

    def test_prednet(initmodel, sequence_list, size, channels, gpu, output_dir, skip_save_frames, extension_start,
                     extension_duration, reset_at, verbose, c_dim):
        seq = sequence_list[0]
        w, h = size
        for g0 in range(0, len(seq) - len(seq) % extension_start, extension_start):
            frames_in = seq[g0:g0 + extension_start]
            assert len(set(frames_in)) == 1 and reset_at == extension_start + extension_duration
            img = read_chw(frames_in[0], c_dim)
            fr = oracle.prednet_rollout(holder["weights"], channels, w, h, img, n_repeat=extension_start, n_ext=extension_duration)
            for t in range(extension_start):
                write_chw(fr[t], output_dir + str(g0 + t).zfill(10) + ".png")
            for j in range(extension_duration):
                write_chw(fr[extension_start + j], output_dir + str(g0 + extension_start - 1 + j + 1).zfill(10) + "_extended.png")

    def lucas_kanade(f0, f1, out_dir, save=True, verbose=0, save_name=""):
        c = holder["c_dim"]
        v = oracle.lucas_kanade(read_chw(f0, c), read_chw(f1, c))
        if save_name:
            os.makedirs(os.path.dirname(save_name) or ".", exist_ok=True)
            open(save_name, "wb").close()
        return {"vectors": [[float(x) for x in row] for row in v]}

    sys.modules["pytorch_neat.pytorch_neat.cppn"].create_cppn = create_cppn
    sys.modules["chainer_prednet.PredNet.call_prednet"].test_prednet = test_prednet
    sys.modules["optical_flow.optical_flow"].lucas_kanade = lucas_kanade
    for m in ("generate_illusion", "fitness_calculator"):
        sys.modules.pop(m, None)
    sys.path.insert(0, REF)
    import generate_illusion as gi

    runs = []
    cwd = os.getcwd()
    for structure, c_dim, w, h, channels, n_pop, seed in [(2, 1, 64, 48, [1, 4, 8], 5, 3), (1, 3, 160, 120, [3, 6, 8], 4, 9), (3, 1, 160, 120, [1, 4, 8], 4, 21)]:
        cfg = synth.make_config(2, 3 if c_dim == 3 else 1)
        pop = synth.make_population(n_pop, cfg, seed=seed)
        holder["weights"] = weights.synthetic_prednet_weights(channels, w, h, seed=seed)
        holder["c_dim"] = c_dim
        with _tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                gi.get_fitnesses_neat(gi.StructureType(structure), pop, "model.npz", cfg, w, h, channels, c_dim=c_dim, best_dir=td, gradient=1)
            finally:
                os.chdir(cwd)
        runs.append({"structure": structure, "c_dim": c_dim, "w": w, "h": h, "channels": channels, "weights_seed": seed,
                     "genomes": [{"key": g.key,
                                  "nodes": {str(k): [n.bias, n.response, n.activation, n.aggregation] for k, n in g.nodes.items()},
                                  "connections": [[list(c.key), c.weight, c.enabled] for c in g.connections.values()]} for _, g in pop],
                     "fitness": [float(g.fitness) for _, g in pop]})
        print("e2e", structure, [round(g.fitness, 6) for _, g in pop])
    # single-image API: fitness_calculator.get_vectors (original -> 2nd extended frame) + calculate_fitness, unmodified
    import fitness_calculator as fcm
    fcm.test_prednet = test_prednet
    fcm.lucas_kanade = lucas_kanade
    single = []
    for structure, c_dim, w, h, channels, seed in [(2, 3, 96, 64, [3, 6, 8], 5), (1, 1, 160, 120, [1, 4, 8], 6)]:
        cfg = synth.make_config(2, 3 if c_dim == 3 else 1)
        g = synth.make_population(1, cfg, seed=seed)[0][1]
        from oracle import grids as ogrids
        grid = ogrids.create_grid(structure, w, h, 10)
        img = ocppn.render(grid, g, cfg, c_dim, w, h)
        holder["weights"] = weights.synthetic_prednet_weights(channels, w, h, seed=seed)
        holder["c_dim"] = c_dim
        with _tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                Image.fromarray(img).save("stim.png")
                v = fcm.get_vectors("stim.png", "model.npz", channels, w, h)
                try:
                    fit = float(fcm.calculate_fitness(fcm.StructureType(structure), v, "stim.png", w, h))
                except UnboundLocalError:
                    fit = "UnboundLocalError"
            finally:
                os.chdir(cwd)
        single.append({"structure": structure, "c_dim": c_dim, "w": w, "h": h, "channels": channels, "weights_seed": seed,
                       "image": img.tolist(), "vectors": np.asarray(v, dtype=np.float64).tolist() if v[0] is not None else None, "fitness": fit})
        print("single", structure, len(v), fit)
    with open(os.path.join(OUT, "e2e_reference_glue.json"), "w") as f:
        _json.dump({"runs": runs, "single": single}, f)


if __name__ == "__main__" and "--e2e" in sys.argv:
    make_e2e()


# ----------------------------------------------------------------------------------------------------------------
# Round-2 additions, in files of their own so that the fixtures above keep their bytes (separate random stream):
#   grids at 256x256 (SURVEY 8(c) lists "64^2, 160x120, 256^2"), enhanced_image_grid for structures without a theta branch,
#   inside_outside_score (fitness_calculator.py:219-304, a12).
def make_round2():
    state = {}
    install_stubs(state)
    sys.path.insert(0, REF)
    import fitness_calculator as fc
    import generate_illusion as gi
    rng = np.random.default_rng(20260929)
    import hashlib
    grids, grid_sha = {}, {}
    # full-size grids: the SHA-256 of the float64 bytes pins every value; every 16th row is kept for diagnosis (the full
    # 256x256 ring planes are ~1 MB of incompressible doubles)
    for name, st, w, h in [("circles_256x256", gi.StructureType.Circles, 256, 256), ("free_256x256", gi.StructureType.Free, 256, 256),
                           ("circlesfree_256x256", gi.StructureType.CirclesFree, 256, 256), ("free_512x512", gi.StructureType.Free, 512, 512)]:
        g = gi.create_grid(st, w, h, 10)
        for ax in ("x", "y"):
            a = np.ascontiguousarray(np.asarray(g[ax + "_mat"], dtype=np.float64).reshape(h, w))
            grid_sha[name + "_" + ax] = hashlib.sha256(a.tobytes()).hexdigest()
            grids[name + "_" + ax + "_rows16"] = a[::16]
    for name, st in [("bands", gi.StructureType.Bands), ("free", gi.StructureType.Free), ("circlesfree", gi.StructureType.CirclesFree)]:
        eg = gi.enhanced_image_grid(120, 120, st)
        grids["enhanced_%s_120x120_x" % name] = eg["x_mat"]
        grids["enhanced_%s_120x120_y" % name] = eg["y_mat"]
    np.savez_compressed(os.path.join(OUT, "grids_round2.npz"), **grids)
    cases = []
    for (w, h), n, mag in [((160, 120), 40, 0.1), ((160, 120), 3, 0.2), ((256, 256), 100, 0.15), ((256, 256), 60, 0.3), ((512, 512), 75, 0.05),
                           ((96, 64), 12, 0.4), ((160, 120), 0, 0.1), ((64, 64), 25, 1.5)]:
        v = vec_set(rng, n, w, h, mag) if n else np.zeros((0, 4))
        if n > 5:
            v[3, 2:] = 0.0                      # a zero-length vector
            v[4, :2] = v[5, :2]                 # two vectors in the same cell, same position
        cases.append({"w": w, "h": h, "vectors": v.tolist(), "score": float(fc.inside_outside_score(v, w, h))})
    ang = np.linspace(0, 2 * np.pi, 48, endpoint=False)
    rot = np.stack([128 + 100 * np.cos(ang), 128 + 100 * np.sin(ang), -0.2 * np.sin(ang), 0.2 * np.cos(ang)], 1)
    cases.append({"w": 256, "h": 256, "vectors": rot.tolist(), "score": float(fc.inside_outside_score(rot, 256, 256))})
    with open(os.path.join(OUT, "round2.json"), "w") as f:
        json.dump({"inside_outside": cases, "grid_sha256": grid_sha}, f)
    print("round-2 fixtures written:", [round(c["score"], 6) for c in cases])


if __name__ == "__main__" and "--round2" in sys.argv:
    make_round2()
