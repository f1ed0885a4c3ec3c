"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle, stage by stage and end to end.

Bars: byte/integer/index outputs bit-exact (rendered images, PredNet frames, corner
lists, vector counts); fp32 conv accumulators and gate math bit-exact against the oracle's canonical fma
chain; flow vectors bit-exact (same integer window sums, same fp32 solve); fitness within 1e-9 relative
(float64 sums in a different order), far inside north_star's 1e-4.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

from evolutionary_illusion_generator_amd import genome as genome_mod
from evolutionary_illusion_generator_amd import synth, weights
from evolutionary_illusion_generator_amd.engine import PAIR_POPULATION, PAIR_SINGLE, Engine


def _eng(w, h, ch, B, **kw):
    return Engine(w, h, ch, B, **kw)


def test_det_math_bit_exact(cuda, oracle_lib):
    import torch
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 3, 200000), rng.uniform(-100, 100, 20000), np.linspace(-1, 1, 20001),
                        [0.0, -0.0, 0.625, -0.625, 80, -80, 88, 1e-30, -1e-30]]).astype(np.float32)
    e = _eng(16, 16, [1, 4], 1)
    dx = torch.from_numpy(x).to(cuda)
    de, ds, dt = (torch.empty_like(dx) for _ in range(3))
    e.test_det_math(dx, x.size, de, ds, dt)
    torch.cuda.synchronize()
    oe, os_, ot = oracle_lib.det_math(x)
    assert np.array_equal(de.cpu().numpy(), oe)
    assert np.array_equal(ds.cpu().numpy(), os_)
    assert np.array_equal(dt.cpu().numpy(), ot)
    # and they are the functions they claim to be
    assert np.max(np.abs(ot - np.tanh(x.astype(np.float64)))) < 3e-7
    assert np.max(np.abs(os_ - 1 / (1 + np.exp(-x.astype(np.float64))))) < 3e-7


CONV_CASES = [
    # (B, H, W, cout, [(cin, up), ...])
    (2, 16, 16, 16, [(8, 0)]),
    (3, 32, 32, 48, [(6, 0)]),            # ConvA1-like, channel padding 6 -> 8, NI = 3
    (2, 16, 32, 3, [(3, 0)]),             # ConvP0-like, cout 3, cin 3 -> 4
    (2, 32, 16, 20, [(16, 0), (12, 1), (8, 0)]),  # three sources, one unpooled
    (5, 8, 8, 64, [(32, 0)]),             # 8x8 maps: four images per block
    (3, 20, 12, 32, [(20, 0), (8, 1)]),   # ragged map -> partial tiles (TW = 8)
    (1, 24, 40, 36, [(40, 0)]),           # ragged map with 16x16 tiles, >1 K-block, cout padding
    (2, 64, 64, 12, [(6, 0), (48, 1), (3, 0)]),  # LSTM0-like source mix
    (2, 12, 10, 16, [(8, 0)]),            # W % 4 != 0: 4-byte DMA path (VEC = false)
    (3, 6, 6, 8, [(5, 0), (4, 1)]),       # tiny odd map with an unpooled 3x3 source
    (1, 48, 80, 64, [(16, 0), (16, 1)]),  # W % 8 == 0: unpooled source staged at its own resolution
    # round 3: strips of 4 columns x 16 rows (conv_mfma.h TW = 4, half blocks of two images) where they cover the map >= 15 % better
    (5, 15, 20, 64, [(24, 0)]),           # 20 x 15, the top-layer map of 160 x 120: five strips, odd image count, ragged last row
    (3, 16, 20, 40, [(12, 0), (8, 1)]),   # 4-wide main chain + the chain of an unpooled 10 x 8 source added after the K loop
    (2, 32, 40, 32, [(8, 0), (16, 1)]),   # the 2x2-form pass itself on 4-wide strips (source 20 x 16), main chain on 8 x 8 tiles
    (4, 16, 20, 100, [(20, 0)]),          # 4-wide, two N-blocks, a partial last K-block
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_chain_bit_exact(cuda, oracle_lib, case):
    import torch
    B, H, W, cout, srcs = case
    rng = np.random.default_rng(hash(case[:4]) & 0xFFFF)
    e = _eng(16, 16, [1, 4], 1)
    hs, hw, ds = [], [], []
    for cin, up in srcs:
        a = rng.normal(0, 1, (B, cin, H >> up, W >> up)).astype(np.float32)
        a[rng.random(a.shape) < 0.3] = 0.0  # relu-like sparsity
        hs.append(a)
        hw.append(rng.normal(0, 0.2, (cout, cin, 3, 3)).astype(np.float32))
        ds.append(torch.from_numpy(a).to(cuda))
    out = torch.full((B, cout, H, W), float("nan"), device=cuda)
    e.test_conv(ds, [c for c, _ in srcs], [u for _, u in srcs], hw, cout, H, W, B, out)
    got = out.cpu().numpy()
    for b in range(B):
        ref = oracle_lib.conv_chain([s[b] for s in hs], [u for _, u in srcs], hw, H, W)
        assert np.array_equal(got[b], ref), "image %d: max abs diff %g" % (b, np.nanmax(np.abs(got[b] - ref)))


def _render_setup(w, h, c_dim, n, seed, structure=1):
    from oracle import grids
    cfg = synth.make_config(2, 3 if c_dim == 3 else 1)
    pop = synth.make_population(n, cfg, seed=seed)
    grid = grids.create_grid(structure, w, h, 10)
    return cfg, pop, grid


@pytest.mark.parametrize("c_dim,gradient,bg", [(3, 1, 1), (1, 1, 1), (3, 0, 1), (1, 0, 0), (3, 1, 0), (3, 2, 1), (3, 2, 0)])
def test_cppn_render_matches_oracle(cuda, oracle_lib, c_dim, gradient, bg):
    """gradient 1 / 0: get_image_from_cppn (generate_illusion.py:372-460); gradient 2: get_equilum_image_from_cppn (:333-367),
    h,s,v nodes through colorsys.hsv_to_rgb per pixel."""
    import torch
    from oracle import cppn
    w, h = 64, 48
    cfg, pop, grid = _render_setup(w, h, c_dim, 24, seed=3)
    # make some genomes hit the constant-node folding
    for gid, g in pop[::4]:
        for key, c in g.connections.items():
            if key[1] in (5, 6):
                c.enabled = False
    e = _eng(w, h, [c_dim, 4, 8], len(pop))
    e.set_grid([grid["x_mat"], grid["y_mat"]])
    gb = genome_mod.GenomeBatch([g for _, g in pop], cfg, c_dim if gradient in (1, 2) else 1)
    img = torch.zeros((len(pop), c_dim, h, w), dtype=torch.uint8, device=cuda)
    e.render_cppn(gb, img, bg=bg, gradient=gradient)
    torch.cuda.synchronize()
    got = img.cpu().numpy()
    n_diff = 0
    for i, (_, g) in enumerate(pop):
        ref = cppn.render(grid, g, cfg, c_dim, w, h, bg=bg, gradient=gradient)
        ref = ref.transpose(2, 0, 1) if ref.ndim == 3 else ref[None]
        n_diff += int((ref != got[i]).sum())
    # device and oracle share the canonical float64 kernels (det_math64.h / detmath64.py): byte-exact
    assert n_diff == 0, "%d bytes differ" % n_diff


@pytest.mark.parametrize("w,h,structure,seed,n", [(160, 120, 2, 5, 32), (256, 256, 1, 0, 16), (64, 64, 3, 9, 40)])
def test_cppn_render_saturating_bands_byte_exact(cuda, oracle_lib, w, h, structure, seed, n):
    """Regression: with libm tanh on one side and ocml tanh on the other, outputs that saturate (1 - 1e-16 vs 1.0)
    quantised to 254 vs 255 along whole contour bands (326 bytes of 1.8 M at 160x120)."""
    import torch
    from oracle import pipeline
    cfg, pop, grid = _render_setup(w, h, 3, n, seed=seed, structure=structure)
    e = _eng(w, h, [3, 4, 8], n)
    e.set_grid([grid["x_mat"], grid["y_mat"]])
    gb = genome_mod.GenomeBatch([g for _, g in pop], cfg, 3)
    img = torch.zeros((n, 3, h, w), dtype=torch.uint8, device=cuda)
    e.render_cppn(gb, img)
    torch.cuda.synchronize()
    got = img.cpu().numpy()
    ref = np.stack([pipeline.render_chw(g, cfg, grid, 3, w, h) for _, g in pop])
    assert np.array_equal(got, ref), "%d bytes differ" % (got != ref).sum()
    assert (ref == 255).mean() > 0.01 and ref.std() > 20


@pytest.mark.parametrize("w,h,ch,requant", [(64, 64, [1, 16, 32, 64], False), (48, 32, [3, 8, 16, 32], False),
                                              (80, 40, [3, 12, 20], True), (160, 120, [1, 16, 32, 64], False),
                                              (20, 12, [1, 4, 8], False),   # widths 20 / 10 / 5: mixed VEC and 4-byte DMA layers
                                              # direct ConvLSTMs with a 12-channel unpooled source (partial last K-block of the 2x2-form pass), ragged tile rows
                                              # (layer 1: 80 x 60 = 3.75 tiles), one-K-block source (8)
                                              (160, 120, [1, 8, 12, 8], False), (64, 96, [1, 8, 12, 8], True),
                                              # five layers, 20 x 16 maps at layer 3 (8 x 8 tiles)
                                              (160, 128, [1, 4, 8, 8, 8], False)])
def test_prednet_rollout_frames_bit_exact(cuda, oracle_lib, w, h, ch, requant, monkeypatch):
    import torch
    from oracle import cppn
    c_dim = ch[0]
    cfg, pop, grid = _render_setup(w, h, c_dim, 3, seed=11)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=4)
    imgs = []
    for _, g in pop:
        r = cppn.render(grid, g, cfg, c_dim, w, h)
        imgs.append(r.transpose(2, 0, 1) if r.ndim == 3 else r[None])
    imgs = np.ascontiguousarray(np.stack(imgs))
    e = _eng(w, h, ch, len(pop), requant_feedback=requant)
    e.set_weights(wts)
    d_img = torch.from_numpy(imgs).to(cuda)
    T = 22
    d_fr = torch.zeros((len(pop), T, c_dim, h, w), dtype=torch.uint8, device=cuda)
    e.prednet_rollout(d_img, len(pop), T, 0, d_fr)
    torch.cuda.synchronize()
    got = d_fr.cpu().numpy()
    for i in range(len(pop)):
        ref = oracle_lib.prednet_rollout(wts, ch, w, h, imgs[i], requant=requant)
        for t in range(T):
            assert np.array_equal(got[i, t], ref[t]), "genome %d step %d: %d bytes differ" % (i, t, (got[i, t] != ref[t]).sum())
    # the population path asks only for steps 19, 20
    d2 = torch.zeros((len(pop), 2, c_dim, h, w), dtype=torch.uint8, device=cuda)
    e.prednet_rollout(d_img, len(pop), 21, 19, d2)
    torch.cuda.synchronize()
    assert np.array_equal(d2.cpu().numpy(), got[:, 19:21])


def _textured_pairs(rng, B, c, h, w):
    """Smooth random textures and a sub-pixel-shifted, slightly perturbed copy (uint8 planar)."""
    from numpy.fft import irfft2, rfft2
    a = rng.normal(0, 1, (B, c, h, w))
    fy, fx = np.meshgrid(np.fft.fftfreq(h), np.fft.rfftfreq(w), indexing="ij")
    filt = np.exp(-(fy ** 2 + fx ** 2) * 60.0)
    base = irfft2(rfft2(a) * filt, s=(h, w))
    sh = irfft2(rfft2(a) * filt * np.exp(-2j * np.pi * (fy * 0.12 + fx * -0.17)), s=(h, w))
    def q(v):
        v = (v - v.min()) / (v.max() - v.min())
        return (v * 255).astype(np.uint8)
    return q(base), q(sh)


@pytest.mark.parametrize("w,h,c", [(64, 64, 1), (160, 120, 3), (96, 40, 3)])
def test_flow_vectors_bit_exact(cuda, oracle_lib, w, h, c):
    import torch
    rng = np.random.default_rng(5)
    B = 6
    i0, i1 = _textured_pairs(rng, B, c, h, w)
    i0[-1] = 128  # a flat image: no corners
    e = _eng(w, h, [c, 4, 8], B)
    d0, d1 = torch.from_numpy(i0).to(cuda), torch.from_numpy(i1).to(cuda)
    dv = torch.zeros((B, e.K, 4), dtype=torch.float32, device=cuda)
    dc = torch.zeros(B, dtype=torch.int32, device=cuda)
    e.flow(d0, c * h * w, d1, c * h * w, B, dv, dc)
    torch.cuda.synchronize()
    corners, ncorn, _, _ = e.debug_corners(B)
    v, n = dv.cpu().numpy(), dc.cpu().numpy()
    total = 0
    for b in range(B):
        g0 = oracle_lib.gray(i0[b])
        ref_c = oracle_lib.good_features(g0)
        assert ncorn[b] == len(ref_c)
        assert np.array_equal(corners[b, :ncorn[b]], ref_c)
        ref = oracle_lib.lucas_kanade(i0[b], i1[b])
        assert n[b] == len(ref), (b, n[b], len(ref))
        assert np.array_equal(v[b, :n[b]], ref), np.abs(v[b, :n[b]] - ref).max()
        total += len(ref)
    assert total > 50 and n[-1] == 0


@pytest.mark.parametrize("w,h,c,B", [(64, 64, 1, 4), (160, 120, 3, 3), (256, 256, 3, 2), (96, 40, 3, 2)])
def test_farneback_dense_flow_bit_exact(cuda, oracle_lib, w, h, c, B):
    """The Farneback option (csrc/farneback_kernels.h) against oracle/farneback.c: dense field and sampled vectors, bit for bit."""
    import torch
    rng = np.random.default_rng(11)
    i0, i1 = _textured_pairs(rng, B, c, h, w)
    i0[-1] = 77  # a flat first frame
    e = _eng(w, h, [c, 4, 8], B, flow="farneback")
    d0, d1 = torch.from_numpy(i0).to(cuda), torch.from_numpy(i1).to(cuda)
    dv = torch.zeros((B, e.K, 4), dtype=torch.float32, device=cuda)
    dc = torch.zeros(B, dtype=torch.int32, device=cuda)
    e.flow(d0, c * h * w, d1, c * h * w, B, dv, dc)
    torch.cuda.synchronize()
    dense = e.debug_dense_flow(B)
    v, n = dv.cpu().numpy(), dc.cpu().numpy()
    params = oracle_lib.FBParams(max_vectors=e.K)
    moved = 0.0
    for b in range(B):
        ref = oracle_lib.farneback_flow(oracle_lib.gray(i0[b]), oracle_lib.gray(i1[b]), params)
        assert np.array_equal(dense[b].transpose(1, 2, 0), ref), np.abs(dense[b].transpose(1, 2, 0) - ref).max()
        rv = oracle_lib.farneback_vectors(ref, params)
        assert n[b] == len(rv) and np.array_equal(v[b, :n[b]], rv)
        moved = max(moved, float(np.abs(ref).max()))
    assert moved > 0.05  # the shifted copies do move


def test_farneback_end_to_end_fitness(cuda, oracle_lib):
    """The whole path with flow="farneback": the same fitness as the oracle pipeline with its Farneback restatement."""
    from evolutionary_illusion_generator_amd import fitness
    from oracle import grids, pipeline
    w, h, ch, structure = 64, 64, [1, 8, 16], 2
    cfg = synth.make_config(2, 1)
    pop = synth.make_population(6, cfg, seed=4)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=2)
    got = fitness.evaluate_population(structure, [g for _, g in pop], wts, cfg, w, h, ch, c_dim=1, flow="farneback")
    grid = grids.create_grid(structure, w, h, 10)
    ref = np.array([pipeline.genome_fitness(g, cfg, grid, wts, ch, w, h, structure, flow="farneback") for _, g in pop])
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-12) and np.isfinite(got).all() and (got != 0).any(), (got, ref)
    lk = fitness.evaluate_population(structure, [g for _, g in pop], wts, cfg, w, h, ch, c_dim=1)
    assert lk.shape == got.shape  # the default path is untouched by the option (separate engine)


@pytest.mark.parametrize("structure", [0, 1, 2, 3])
def test_scores_match_oracle(cuda, structure):
    import torch
    from oracle import scores
    rng = np.random.default_rng(structure)
    w, h, K = 160, 120, 100
    B = 12
    e = _eng(w, h, [1, 4, 8], B)
    vec = np.zeros((B, K, 4), np.float32)
    cnt = np.zeros(B, np.int32)
    for b in range(B):
        n = [0, 1, 2, 24, 25, 26, 40, 75, 100, 60, 30, 99][b]
        cnt[b] = n
        vec[b, :n, 0] = rng.integers(1, w - 1, n); vec[b, :n, 1] = rng.integers(1, h - 1, n)
        vec[b, :n, 2:] = rng.normal(0, [0.05, 0.1, 0.12, 0.2][b % 4], (n, 2))
    dv, dc = torch.from_numpy(vec).to(cuda), torch.from_numpy(cnt).to(cuda)
    df = torch.zeros(B, dtype=torch.float64, device=cuda)
    e.score(structure, dv, dc, B, df)
    torch.cuda.synchronize()
    got = df.cpu().numpy()
    nz = 0
    for b in range(B):
        ref = scores.fitness_from_vectors(structure, vec[b, :cnt[b]].astype(np.float64), w, h)
        assert got[b] == pytest.approx(ref, rel=1e-9, abs=1e-12), (b, got[b], ref)
        nz += ref != 0
    assert nz >= 3


@pytest.mark.parametrize("w,h,ch,structure,pairing", [(64, 64, [1, 16, 32, 64], 2, PAIR_POPULATION),
                                                       (160, 120, [1, 16, 32, 64], 1, PAIR_POPULATION),
                                                       (96, 64, [3, 12, 24, 48], 1, PAIR_POPULATION),
                                                       (160, 120, [1, 8, 16, 32], 3, PAIR_POPULATION),
                                                       (64, 64, [3, 8, 16, 32], 2, PAIR_SINGLE)])
def test_end_to_end_fitness(cuda, oracle_lib, w, h, ch, structure, pairing):
    from oracle import pipeline
    c_dim = ch[0]
    cfg, pop, grid = _render_setup(w, h, c_dim, 6, seed=21, structure=structure)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=7)
    e = _eng(w, h, ch, len(pop))
    e.set_weights(wts)
    e.set_grid([grid["x_mat"], grid["y_mat"]])
    gb = genome_mod.GenomeBatch([g for _, g in pop], cfg, c_dim)
    got = e.eval_population(gb, structure, pairing=pairing)
    ref = np.array([pipeline.genome_fitness(g, cfg, grid, wts, ch, w, h, structure, pairing=pairing) for _, g in pop])
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-12, equal_nan=True), (got, ref)
    assert (ref != 0).sum() >= 1, "vacuous parity: the oracle scored everything 0"


# ------------------------------------------------------------------------------------------ full-size checks
def test_full_size_256_colour_properties_and_oracle_spot_check(cuda, oracle_lib):
    """BASELINE.json's headline shape (256x256, channels 3,48,96,192).  The CPU oracle needs ~10 s per genome here, so
    ONE genome is compared end to end; the rest are size-independent properties: batch-position invariance
    (duplicates in one batch agree bit for bit), batch-size invariance (a genome alone == inside a batch) and
    run-to-run determinism."""
    import torch
    from oracle import pipeline
    w, h, ch, structure = 256, 256, [3, 48, 96, 192], 1
    cfg, pop, grid = _render_setup(w, h, 3, 5, seed=0, structure=structure)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=0)
    genomes = [g for _, g in pop]
    batch = [genomes[0], genomes[1], genomes[0], genomes[2], genomes[3], genomes[1], genomes[4]]
    e = _eng(w, h, ch, len(batch))
    e.set_weights(wts)
    e.set_grid([grid["x_mat"], grid["y_mat"]])
    gb = genome_mod.GenomeBatch(batch, cfg, 3)
    f1 = e.eval_population(gb, structure)
    f2 = e.eval_population(gb, structure)
    assert np.array_equal(f1, f2)                                 # deterministic
    assert f1[0] == f1[2] and f1[1] == f1[5]                      # position in the batch does not matter
    alone = e.eval_population(genome_mod.GenomeBatch([genomes[3]], cfg, 3), structure)
    assert alone[0] == f1[4]                                      # nor does the batch size
    # frames of duplicates are byte-identical
    d_img = torch.zeros((len(batch), 3, h, w), dtype=torch.uint8, device=cuda)
    e.render_cppn(gb, d_img)
    d_fr = torch.zeros((len(batch), 2, 3, h, w), dtype=torch.uint8, device=cuda)
    e.prednet_rollout(d_img, len(batch), 21, 19, d_fr)
    torch.cuda.synchronize()
    fr = d_fr.cpu().numpy()
    assert np.array_equal(fr[0], fr[2]) and np.array_equal(fr[1], fr[5]) and not np.array_equal(fr[0], fr[1])
    # one genome against the bit-exact CPU oracle at full size: frames and fitness
    img = pipeline.render_chw(genomes[0], cfg, grid, 3, w, h)
    assert np.array_equal(img, d_img[0].cpu().numpy())
    ref_fr = oracle_lib.prednet_rollout(wts, ch, w, h, img, n_repeat=20, n_ext=1)
    assert np.array_equal(ref_fr[19], fr[0, 0]) and np.array_equal(ref_fr[20], fr[0, 1])
    ref = pipeline.image_fitness(img, wts, ch, w, h, structure)
    assert f1[0] == pytest.approx(ref, rel=1e-9, abs=1e-12) and ref != 0


def test_largest_config_512_colour_one_step_window(cuda, oracle_lib):
    """BASELINE.json configs[4] shape (512x512 colour): a short roll-out (3 repeats + 1 extension) against the oracle
    keeps the CPU side to a few seconds while still covering every layer shape of that configuration."""
    import torch
    from oracle import pipeline
    w, h, ch = 512, 512, [3, 48, 96, 192]
    cfg, pop, grid = _render_setup(w, h, 3, 2, seed=4, structure=2)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=1)
    imgs = np.stack([pipeline.render_chw(g, cfg, grid, 3, w, h) for _, g in pop])
    e = _eng(w, h, ch, 2, n_repeat=3, n_ext=1)
    e.set_weights(wts)
    d_img = torch.from_numpy(imgs).to(cuda)
    d_fr = torch.zeros((2, 4, 3, h, w), dtype=torch.uint8, device=cuda)
    e.prednet_rollout(d_img, 2, 4, 0, d_fr)
    torch.cuda.synchronize()
    got = d_fr.cpu().numpy()
    ref = oracle_lib.prednet_rollout(wts, ch, w, h, imgs[0], n_repeat=3, n_ext=1)
    assert np.array_equal(got[0], ref)
    v = oracle_lib.lucas_kanade(ref[2], ref[3])
    dv = torch.zeros((2, e.K, 4), device=cuda); dc = torch.zeros(2, dtype=torch.int32, device=cuda)
    e.flow(d_fr[:, 2].contiguous(), 3 * h * w, d_fr[:, 3].contiguous(), 3 * h * w, 2, dv, dc)
    torch.cuda.synchronize()
    assert int(dc[0]) == len(v) and np.array_equal(dv[0, :len(v)].cpu().numpy(), v)


def test_reference_big_size_640x480_colour_window(cuda, oracle_lib):
    """The reference's other real size, `--size big` = 640x480 (generate_illusion.py:742-746), colour 3,48,96,192: maps of
    640x480 / 320x240 / 160x120 / 80x60 -- 80x60 is a tiling case no other configuration has (60 rows: 16-row tiles cover 94 %,
    8-row tiles 100 %).  3 repeats + 1 extension of three genomes against the C oracle (every layer shape, step-0 operators,
    steady-state operators, the extension feedback; one genome through the oracle: 0.8 TFLOP of scalar C) and Lucas-Kanade on
    the last pair; the other genomes through batch-position invariance."""
    import torch
    from oracle import pipeline
    w, h, ch = 640, 480, [3, 48, 96, 192]
    cfg, pop, grid = _render_setup(w, h, 3, 3, seed=6, structure=1)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=2)
    imgs = np.stack([pipeline.render_chw(g, cfg, grid, 3, w, h) for _, g in pop])
    e = _eng(w, h, ch, 3, n_repeat=3, n_ext=1)
    e.set_weights(wts)
    d_img = torch.from_numpy(imgs).to(cuda)
    d_fr = torch.zeros((3, 4, 3, h, w), dtype=torch.uint8, device=cuda)
    e.prednet_rollout(d_img, 3, 4, 0, d_fr)
    torch.cuda.synchronize()
    got = d_fr.cpu().numpy()
    ref = oracle_lib.prednet_rollout(wts, ch, w, h, imgs[2], n_repeat=3, n_ext=1)
    assert np.array_equal(got[2], ref)
    d_rev = torch.zeros((3, 4, 3, h, w), dtype=torch.uint8, device=cuda)   # the same genomes at other batch positions
    e.prednet_rollout(torch.from_numpy(np.ascontiguousarray(imgs[::-1])).to(cuda), 3, 4, 0, d_rev)
    torch.cuda.synchronize()
    assert np.array_equal(d_rev.cpu().numpy()[::-1], got)
    v = oracle_lib.lucas_kanade(ref[2], ref[3])
    dv = torch.zeros((3, e.K, 4), device=cuda); dc = torch.zeros(3, dtype=torch.int32, device=cuda)
    e.flow(d_fr[:, 2].contiguous(), 3 * h * w, d_fr[:, 3].contiguous(), 3 * h * w, 3, dv, dc)
    torch.cuda.synchronize()
    assert int(dc[2]) == len(v) and np.array_equal(dv[2, :len(v)].cpu().numpy(), v)
    e.close()


_WINO_SCRIPT = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
import oracle
from evolutionary_illusion_generator_amd import weights
from evolutionary_illusion_generator_amd.engine import Engine
assert oracle.wino_mask_default() == %(mask)d
ok = True
# (w, h, channels, batch): 16-channel gate groups at layers >= 1; ragged 16 x 16 tiles (40 x 24, 20 x 12 maps), a 4-layer net, a top
# layer without an unpooled source, colour and gray image layers (which keep the direct operators); 48- / 96- / 192-channel ConvA and
# ConvP (N-blocks of 48 and of 64 columns); the reference's own 160 x 120 with its 20 x 15 top layer (odd height)
# and at three images (packed tiles: three main blocks + an edge block with one image missing); 128 x 128 with a 16 x 16 top layer of 48 channels (packed tiles, 4 x 4 per
# image: main blocks only; ConvP in N-blocks of 48 columns); 160 x 104 gray at five images: a 20 x 13 top layer (13 rows of 16), a second edge block holding ONE image
for (w, h, ch, B) in [(64, 64, [3, 16, 32], 3), (80, 48, [1, 16, 32, 48], 2), (96, 64, [3, 48, 96], 2), (160, 120, [3, 48, 96, 192], 3), (128, 128, [3, 16, 32, 48], 3), (160, 104, [1, 16, 32, 48], 5)]:
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (B, ch[0], h, w), dtype=np.uint8)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=5)
    e = Engine(w, h, ch, B, n_repeat=4, n_ext=2)
    e.set_weights(wts)
    fr = torch.zeros((B, 6, ch[0], h, w), dtype=torch.uint8, device="cuda")
    e.conv_profile(True)
    e.prednet_rollout(torch.from_numpy(img).cuda(), B, 6, 0, fr)
    torch.cuda.synchronize()
    got = fr.cpu().numpy()
    took = sorted({(r["epi"], r["layer"]) for r in e.conv_profile(False) if r["wino"] and r["launches"]})
    for b in range(B):
        ref = oracle.prednet_rollout(wts, ch, w, h, img[b], n_repeat=4, n_ext=2)            # the same switch: EIGEN_WINOGRAD
        same = np.array_equal(got[b], ref)
        note = ""
        if b == 0:   # (information only: how far the selected canonical order is from the direct one)
            direct = oracle.prednet_rollout(wts, ch, w, h, img[b], n_repeat=4, n_ext=2, wino_mask=0)
            note = "| differs from the direct order in %%d bytes " %% int((ref != direct).sum())
        print("WINO", (w, h, ch), b, "bit-exact" if same else "MISMATCH %%d bytes" %% int((got[b] != ref).sum()), note + "| Winograd operators:", took)
        ok = ok and same
    e.close()
print("WINO_OK" if ok else "WINO_FAIL")
"""


_WINO_SWITCHES = [None, "parts=1", "tall=1", "half=1", "pack=0", "0x03FFFFFE", "0x0C0E0E00"]
_WINO_RUNS = {}


def _wino_run(switch):
    import subprocess
    env = dict(os.environ)
    for k in ("EIGEN_WINOGRAD", "EIGEN_W4_PARTS", "EIGEN_W4_TALL", "EIGEN_W4_HALF", "EIGEN_W4_PACK"):
        env.pop(k, None)
    mask = 0x0FFFFFFE
    if switch is not None and "=" in switch:
        name, val = switch.split("=")
        env["EIGEN_W4_" + name.upper()] = val
        if name == "half":
            env["EIGEN_W4_TALL"] = "0"   # (the half blocks exist in the wide shape)
    elif switch is not None:
        env["EIGEN_WINOGRAD"] = switch
        mask = int(switch, 0)
    return subprocess.run([sys.executable, "-c", _WINO_SCRIPT % {"root": ROOT, "mask": mask}], env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("switch", _WINO_SWITCHES)
def test_winograd_operators_frames_bit_exact(cuda, oracle_lib, switch):
    """The Winograd forms of the 3x3 convolutions against the oracle's statement of exactly that arithmetic (eig_oracle.c: wino_*, wino4_*; the oracle follows the
    same environment switch): all frames of six small roll-outs, bit for bit -- incl. step-0 operators (one source), ragged tiles, a top layer without an unpooled
    source, the 20 x 15 top layer of 160 x 120 (odd height), N-blocks of 48 and 64 columns.
    None = THE DEFAULT (0x0FFFFFFE): every eligible ConvLSTM / ConvA / ConvP as Winograd F(4x4, 3x3) on the twelve-wave kernel (csrc/conv_wino4.h), the unpooled source
    inside the ConvLSTM's chains -- launches this small do not walk, and the block shape is picked per operator by map size (these roll-outs include 80 x 60 and 40 x 30
    maps, which take the tall shape, and a 20 x 15 one, which does not); "parts=1": the same with a block of that kernel walking ALL N-blocks of its tile (EIGEN_W4_PARTS: the
    launch geometry must not show in a single bit; test_specialised_operators... covers 2 / 99 and the forced shapes on other roll-outs); "tall=1": every F(4x4) operator on
    32 x 16-pixel blocks (EIGEN_W4_TALL); "half=1": every F(4x4) operator on 8 x 32-pixel half blocks (EIGEN_W4_HALF; the shape of launches smaller than one block per compute unit:
    twelve waves with the N-tiles split for 64-column ConvLSTMs / ConvPs, six waves otherwise); "pack=0": no packed tiles (EIGEN_W4_PACK: by default the ConvLSTM and the
    ConvP of a 20 x 15 or 16 x 16 top layer run on main + edge half blocks -- the 160 x 120 roll-out, three images = an edge block with one image missing, and the
    128 x 128 one here); 0x03FFFFFE: only the ConvLSTMs in F(4x4), ConvA / ConvP direct (class bits 26 / 27 clear); 0x0C0E0E00: ConvA and ConvP in
    F(4x4), every ConvLSTM direct.  (The F(2x2, 3x3) sixteen-wave kernel of rounds 4-5 and its masks went in round 6.)"""
    if not _WINO_RUNS:   # the settings are independent fresh processes (small roll-outs on the one GPU, the oracle on the CPU): started together, four at a time
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=4) as pool:
            _WINO_RUNS.update(zip(_WINO_SWITCHES, pool.map(_wino_run, _WINO_SWITCHES)))
    r = _WINO_RUNS[switch]
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    print(r.stdout[-3000:])
    assert "WINO_OK" in r.stdout, r.stdout[-3000:]


def test_default_config_four_inputs_six_outputs(cuda, oracle_lib):
    """neat_configs/default.txt (num_inputs = 4, num_outputs = 6; BASELINE.json configs[0], 64x64 gray): the reference
    asserts here; build-defined: leaves x, y, r = sqrt(x^2 + y^2), bias = 1 and the first c_dim outputs."""
    from oracle import cppn, pipeline, scores
    import oracle
    w, h, ch, structure = 64, 64, [1, 16, 32, 64], 2
    cfg = synth.make_config(4, 6)
    pop = synth.make_population(10, cfg, seed=2, num_hidden=8)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=0)
    from evolutionary_illusion_generator_amd import fitness, grids
    fitness.get_fitnesses_neat(structure, pop, wts, cfg, w, h, ch, c_dim=1, best_dir=None)
    grid = grids.create_grid(structure, w, h, 10)
    x, y = grid["x_mat"].reshape(-1), grid["y_mat"].reshape(-1)
    extra = [np.sqrt(x * x + y * y), np.ones_like(x)]
    for _, g in pop:
        img = cppn.render(grid, g, cfg, 1, w, h, extra_leaves=extra)[None]
        ref = pipeline.image_fitness(np.ascontiguousarray(img), wts, ch, w, h, structure)
        assert g.fitness == pytest.approx(ref, rel=1e-9, abs=1e-12)


def test_flow_rare_branches_bit_exact(cuda, oracle_lib):
    """Lucas-Kanade paths that smooth sub-pixel pairs never take: multi-pixel motion (iteration cap, lost tracks),
    noise (oscillation damping branch), corners hugging the border (window reads through the REFLECT_101 / zero
    borders), a level that the pyramid refuses to build (small image)."""
    import torch
    rng = np.random.default_rng(77)
    lost_total = 0
    for (w, h, c, B) in [(96, 72, 1, 24), (40, 36, 3, 8), (160, 120, 3, 8)]:
        i0 = np.zeros((B, c, h, w), np.uint8)
        i1 = np.zeros((B, c, h, w), np.uint8)
        for b in range(B):
            base = rng.integers(0, 256, (c, h // 4 + 12, w // 4 + 12)).astype(np.float64)
            big = np.kron(base, np.ones((1, 4, 4)))[:, :h + 40, :w + 40]                    # blocky texture: strong corners
            big = (big + np.roll(big, 1, 1) + np.roll(big, 1, 2)) / 3.0
            sy, sx = rng.integers(0, 6, 2) if b % 4 else rng.integers(0, 3, 2) * 13                # 0..5 px, sometimes 13/26 px
            a = big[:, 30:30 + h, 30:30 + w]
            bimg = big[:, 30 - sy:30 - sy + h, 30 - sx:30 - sx + w]
            noise = rng.normal(0, [0, 3, 12][b % 3], a.shape)
            i0[b] = np.clip(a, 0, 255)
            i1[b] = np.clip(bimg + noise, 0, 255)
        e = _eng(w if w % 2 == 0 else w + 1, h if h % 2 == 0 else h + 1, [c, 4], B)
        d0, d1 = torch.from_numpy(i0).to(cuda), torch.from_numpy(i1).to(cuda)
        dv = torch.zeros((B, e.K, 4), dtype=torch.float32, device=cuda)
        dc = torch.zeros(B, dtype=torch.int32, device=cuda)
        e.flow(d0, c * h * w, d1, c * h * w, B, dv, dc)
        torch.cuda.synchronize()
        corners, ncorn, nxt, st = e.debug_corners(B)
        lost = 0
        for b in range(B):
            g0, g1 = oracle_lib.gray(i0[b]), oracle_lib.gray(i1[b])
            pts = oracle_lib.good_features(g0)
            assert ncorn[b] == len(pts) and np.array_equal(corners[b, :len(pts)], pts)
            rn, rs = oracle_lib.pyr_lk(g0, g1, pts)
            assert np.array_equal(st[b, :len(pts)], rs)
            assert np.array_equal(nxt[b, :len(pts)][rs == 1], rn[rs == 1])
            lost += int((rs == 0).sum())
            ref = oracle_lib.lucas_kanade(i0[b], i1[b])
            assert int(dc[b]) == len(ref) and np.array_equal(dv[b, :len(ref)].cpu().numpy(), ref)
        lost_total += lost
    assert lost_total > 0, "no track was ever lost: the status==0 paths were not exercised"


def test_cppn_render_evolved_and_extreme_genomes_byte_exact(cuda, oracle_lib):
    """Genomes after several generations of neat_lite mutation (added/removed nodes and links, mixed activations), with
    weights pushed to the +-30 clamp so that sin / exp / tanh see large arguments and the outputs wrap mod 256."""
    import os
    import random
    import torch
    from evolutionary_illusion_generator_amd import grids, neat_lite as neat
    from oracle import pipeline
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = neat.Config(neat.DefaultGenome, neat.DefaultReproduction, neat.DefaultSpeciesSet, neat.DefaultStagnation,
                      os.path.join(root, "examples", "circles_neat.cfg"))
    p = neat.Population(cfg, seed=5)
    rnd = random.Random(1)

    def fake(genomes, config):
        for _, g in genomes:
            g.fitness = rnd.random()

    p.run(fake, 6)
    genomes = list(p.population.values())[:48]
    for i, g in enumerate(genomes):
        if i % 3 == 0:
            for c in g.connections.values():
                c.weight = max(-30.0, min(30.0, c.weight * 12.0))
        if i % 5 == 0:
            for n in g.nodes.values():
                n.activation = ["identity", "sin", "gauss", "abs"][i % 4]
    w, h = 64, 64
    for structure in (1, 2):
        grid = grids.create_grid(structure, w, h, 10)
        e = _eng(w, h, [3, 4, 8], len(genomes))
        e.set_grid([grid["x_mat"], grid["y_mat"]])
        gb = genome_mod.GenomeBatch(genomes, cfg, 3)
        img = torch.zeros((len(genomes), 3, h, w), dtype=torch.uint8, device=cuda)
        e.render_cppn(gb, img)
        torch.cuda.synchronize()
        got = img.cpu().numpy()
        ref = np.stack([pipeline.render_chw(g, cfg, grid, 3, w, h) for g in genomes])
        assert np.array_equal(got, ref), "%d bytes differ (structure %d)" % ((got != ref).sum(), structure)
    sizes = [g.size()[0] for g in genomes]
    assert max(sizes) > min(sizes)
