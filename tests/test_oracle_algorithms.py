"""CPU: self-consistency of the C oracle (the stages whose parity is UNPINNED -- no reference fixture exists)."""
import os

import numpy as np
import pytest


def _fma32(v, w, acc):  # fused multiply-add: exact product, one rounding
    return np.float32(np.float64(v) * np.float64(w) + np.float64(acc))


def test_conv_chain_is_the_documented_fma_chain(oracle_lib):
    """one 9-term (c, ky, kx) chain over the full-resolution sources plus (one fp32 addition) the chain of
    the unpooled source in its 2x2 form with pre-summed weights (4 terms per channel); and that 2x2 form IS unpool -> conv3x3
    up to summation order."""
    rng = np.random.default_rng(0)
    H, W = 6, 8
    srcs = [rng.normal(0, 1, (5, H, W)).astype(np.float32), rng.normal(0, 1, (3, H // 2, W // 2)).astype(np.float32)]
    ws = [rng.normal(0, 0.3, (4, 5, 3, 3)).astype(np.float32), rng.normal(0, 0.3, (4, 3, 3, 3)).astype(np.float32)]
    got = oracle_lib.conv_chain(srcs, [0, 1], ws, H, W)
    ref = np.zeros((4, H, W), np.float32)
    plain = np.zeros((4, H, W), np.float64)  # unpool x2 -> conv3x3 + conv3x3 in float64
    rows = {0: ([0], [1, 2]), 1: ([0, 1], [2])}  # parity -> 3x3 taps collected by 2x2 tap 0 / 1
    for o in range(4):
        for y in range(H):
            for x in range(W):
                acc_up = np.float32(0)
                py, px, Y, X = y & 1, x & 1, y >> 1, x >> 1
                for c in range(srcs[1].shape[0]):
                    for a in range(2):
                        for b in range(2):
                            wsum = None
                            for ky in rows[py][a]:
                                for kx in rows[px][b]:
                                    wsum = ws[1][o, c, ky, kx] if wsum is None else np.float32(wsum + ws[1][o, c, ky, kx])
                            sy, sx = Y + a - 1 + py, X + b - 1 + px
                            v = srcs[1][c, sy, sx] if (0 <= sy < H // 2 and 0 <= sx < W // 2) else np.float32(0)
                            acc_up = _fma32(v, wsum, acc_up)
                acc = np.float32(0)
                for c in range(srcs[0].shape[0]):
                    for ky in range(3):
                        for kx in range(3):
                            yy, xx = y + ky - 1, x + kx - 1
                            v = srcs[0][c, yy, xx] if (0 <= yy < H and 0 <= xx < W) else np.float32(0)
                            acc = _fma32(v, ws[0][o, c, ky, kx], acc)
                ref[o, y, x] = np.float32(acc + acc_up)
                for src, w, up in zip(srcs, ws, [0, 1]):
                    for c in range(src.shape[0]):
                        for ky in range(3):
                            for kx in range(3):
                                yy, xx = y + ky - 1, x + kx - 1
                                if 0 <= yy < H and 0 <= xx < W:
                                    plain[o, y, x] += np.float64(src[c, yy >> up, xx >> up]) * np.float64(w[o, c, ky, kx])
    assert np.array_equal(got, ref)
    assert np.max(np.abs(got - plain)) < 5e-6  # same function, fp32 summation order aside


def test_det_math32_accuracy(oracle_lib):
    x = np.concatenate([np.linspace(-30, 30, 200001), [0.0, 0.625, -0.625, 80, -80, 100, -100]]).astype(np.float32)
    e, s, t = oracle_lib.det_math(x)
    xd = x.astype(np.float64)
    ok = np.abs(xd) <= 80
    assert np.max(np.abs(e[ok] - np.exp(xd[ok])) / np.exp(xd[ok])) < 3e-7
    assert np.max(np.abs(s - 1 / (1 + np.exp(-xd)))) < 2e-7
    assert np.max(np.abs(t - np.tanh(xd))) < 2e-7
    assert t[np.abs(x) > 10].tolist() == np.sign(x[np.abs(x) > 10]).tolist()


def test_det_math64_accuracy():
    from oracle import detmath64 as d
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(0, 5, 100000), rng.uniform(-300, 300, 50000), np.linspace(-1, 1, 4001)])
    ulp = lambda a, b: np.abs(a - b) / np.spacing(np.abs(b))
    assert ulp(d.det_exp(x), np.exp(x)).max() <= 2
    assert ulp(d.det_tanh(x), np.tanh(x))[x != 0].max() <= 3
    s = np.sin(x)
    assert (np.abs(d.det_sin(x) - s) <= 4 * np.spacing(np.abs(s)) + 1e-16).all()
    assert np.all(d.det_tanh(np.array([20.0, 40.0, 1e300, np.inf])) == 1.0) and np.all(d.det_tanh(-np.array([20.0, np.inf])) == -1.0)
    assert np.isnan(d.det_sin(np.array([np.inf, np.nan]))).all() and np.isnan(d.det_exp(np.array([np.nan]))).all()


@pytest.mark.parametrize("w,h,ch,requant", [(32, 24, [1, 4, 8], False), (32, 32, [3, 6, 8, 12], False), (40, 24, [3, 5, 7], True)])
def test_prednet_c_matches_independent_torch_restatement(oracle_lib, w, h, ch, requant):
    from evolutionary_illusion_generator_amd import weights
    from oracle.prednet_torch import PredNetTorch
    rng = np.random.default_rng(3)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=5)
    img = rng.integers(0, 256, (ch[0], h, w)).astype(np.uint8)
    img[:, : h // 2] = np.clip(np.linspace(0, 255, w)[None, None, :] + rng.normal(0, 8, (ch[0], h // 2, w)), 0, 255).astype(np.uint8)
    fr, p0 = oracle_lib.prednet_rollout(wts, ch, w, h, img, requant=requant, return_float=True)
    f2, p2 = PredNetTorch(wts, ch, w, h).rollout(img[None], requant=requant)
    if not requant:
        assert np.abs(p0 - p2[0]).max() < 5e-6
    assert (fr != f2[0]).mean() < 2e-3          # uint8 flips only at quantisation boundaries
    assert np.abs(fr.astype(int) - f2[0].astype(int)).max() <= (1 if not requant else 2)
    assert fr.shape == (22, ch[0], h, w)
    assert np.abs(fr[19].astype(int) - img.astype(int)).mean() < 40  # the synthetic net does track its input


def test_oracle_threads_do_not_change_a_bit(oracle_lib):
    """The C oracle's convolution loops run one OpenMP work item per (output channel, row); every output pixel is its own fma
    chain, so the thread count cannot change a result (the GPU suite's full-size oracle cases on a many-core box)."""
    from evolutionary_illusion_generator_amd import weights
    ch, w, h = [3, 6, 12], 48, 32
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=4)
    img = (np.random.default_rng(4).random((3, h, w)) * 255).astype(np.uint8)
    n0 = oracle_lib.lib().eig_oracle_get_threads()
    try:
        outs = []
        for th in (1, 3, max(2, n0)):
            oracle_lib.set_threads(th)
            assert oracle_lib.lib().eig_oracle_get_threads() == th
            outs.append([oracle_lib.prednet_rollout(wts, ch, w, h, img, 3, 1, return_float=True, order=o) for o in ("canonical", "chainer")])
    finally:
        oracle_lib.set_threads(n0)
    for o in outs[1:]:
        for (fa, pa), (fb, pb) in zip(outs[0], o):
            assert np.array_equal(fa, fb) and np.array_equal(pa, pb)


def test_winograd_statement_is_the_same_convolution(oracle_lib):
    """oracle/eig_oracle.c: wino_* (the canonical arithmetic of the operators csrc/conv_wino4.h takes; m = 2 kept as a study form) IS the 3x3 'same' convolution: against a
    float64 reference it is as accurate as the direct fma chain (F(2x2, 3x3) in fp32: ~1e-6 relative), on even and on odd heights (a
    20 x 15 top-layer map), with several chained sources; and a roll-out under any switch setting stays within fp32 round-off of the
    direct one -- the switch selects a summation order, never a different function."""
    import torch
    import torch.nn.functional as F
    from evolutionary_illusion_generator_amd import weights
    rng = np.random.default_rng(3)
    for H, W, cins, cout in ((12, 20, (6, 3), 5), (15, 20, (8,), 16), (2, 4, (1, 2, 3), 2)):
        srcs = [rng.standard_normal((c, H, W)).astype(np.float32) for c in cins]
        ws = [rng.standard_normal((cout, c, 3, 3)).astype(np.float32) for c in cins]
        ref = sum(F.conv2d(torch.from_numpy(s_).double()[None], torch.from_numpy(w_).double(), padding=1)[0] for s_, w_ in zip(srcs, ws)).numpy()
        wino = oracle_lib.wino_chain(srcs, ws, H, W)
        direct = oracle_lib.conv_chain(srcs, [0] * len(srcs), ws, H, W)
        scale = np.abs(ref).max()
        assert np.abs(wino - ref).max() <= 3e-6 * scale and np.abs(direct - ref).max() <= 3e-6 * scale
        assert not np.array_equal(wino, direct)   # ... but another order: not the same bits
        # F(4x4, 3x3) (round 5, csrc/conv_wino4.h): the same convolution again, at the round-off of its larger transforms (~1e-5 relative)
        wino4 = oracle_lib.wino_chain(srcs, ws, H, W, m=4)
        assert np.abs(wino4 - ref).max() <= 3e-5 * scale and not np.array_equal(wino4, wino)
    with pytest.raises(ValueError):
        oracle_lib.wino_chain([np.zeros((1, 4, 5), np.float32)], [np.zeros((1, 1, 3, 3), np.float32)], 4, 5)
    ch, w, h = [3, 48, 96], 64, 40   # ConvLSTM 1-2, ConvA 2, ConvP 1-2 eligible; layer 1 has W % 8 == 0: its unpooled source can ride in the chains
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=9)
    img = (rng.random((3, h, w)) * 255).astype(np.uint8)
    _, p_direct = oracle_lib.prednet_rollout(wts, ch, w, h, img, 4, 1, return_float=True, wino_mask=0)
    seen = set()
    # an operator is a Winograd one with its own bit AND its class bit (25 ConvLSTMs / 26 ConvAs / 27 ConvPs): the top ConvLSTM alone, ConvA_2, ConvP_1-2, all of
    # them, the ConvLSTMs only, all without the unpooled source in the chains (bit 24 clear: ConvLSTM_1 is then a direct operator)
    for mask in (0x02000006, 0x04000600, 0x08060000, 0x0FFFFFFE, 0x03FFFFFE, 0x0EFFFFFE):
        _, p = oracle_lib.prednet_rollout(wts, ch, w, h, img, 4, 1, return_float=True, wino_mask=mask)
        assert np.abs(p - p_direct).max() <= 2e-5
        seen.add(p.tobytes())
    assert len(seen) == 6 and p_direct.tobytes() not in seen   # every setting is its own (documented) order
    for mask in (0x00FFFFFE, 0x01FFFFFE):   # no class bit: every operator direct (rounds 4-5 ran F(2x2, 3x3) kernels here; removed in round 6)
        _, p = oracle_lib.prednet_rollout(wts, ch, w, h, img, 4, 1, return_float=True, wino_mask=mask)
        assert p.tobytes() == p_direct.tobytes()
    assert oracle_lib.wino_mask_default() == 0x0FFFFFFE or "EIGEN_WINOGRAD" in os.environ or os.environ.get("EIGEN_WINO_FUSEUP") == "0"


def test_prednet_rejects_sizes_the_pooling_cannot_halve(oracle_lib):
    from evolutionary_illusion_generator_amd import weights
    with pytest.raises(ValueError):
        weights.tensor_shapes([1, 4, 8, 16], 36, 36)
    wts = weights.synthetic_prednet_weights([1, 4, 8], 12, 12)
    with pytest.raises(ValueError):
        oracle_lib.prednet_rollout(wts, [1, 4, 8], 10, 10, np.zeros((1, 10, 10), np.uint8))


def test_gray_and_pyrdown_known_values(oracle_lib):
    img = np.zeros((3, 4, 4), np.uint8)
    img[0], img[1], img[2] = 255, 0, 0
    assert oracle_lib.gray(img)[0, 0] == (255 * 9798 + (1 << 14)) >> 15  # 76
    img[:] = 200
    assert np.all(oracle_lib.gray(img) == 200)
    g = np.full((8, 10), 37, np.uint8)
    assert np.all(oracle_lib.pyr_down(g) == 37) and oracle_lib.pyr_down(g).shape == (4, 5)
    g = np.zeros((9, 9), np.uint8); g[4, 4] = 255
    d = oracle_lib.pyr_down(g)
    assert d.shape == (5, 5) and d[2, 2] == (255 * 36 + 128) >> 8 and d[1, 2] == (255 * 6 + 128) >> 8


def _texture(rng, h, w, shift=(0.0, 0.0)):
    from numpy.fft import irfft2, rfft2
    a = rng.normal(0, 1, (h, w))
    fy, fx = np.meshgrid(np.fft.fftfreq(h), np.fft.rfftfreq(w), indexing="ij")
    filt = np.exp(-(fy ** 2 + fx ** 2) * 80.0)
    ph = np.exp(-2j * np.pi * (fy * shift[1] + fx * shift[0]))
    f = rfft2(a) * filt
    base, sh = irfft2(f, s=(h, w)), irfft2(f * ph, s=(h, w))
    lo, hi = base.min(), base.max()
    q = lambda v: np.clip((v - lo) / (hi - lo) * 255, 0, 255).astype(np.uint8)
    return q(base), q(sh)


def test_lucas_kanade_recovers_a_known_subpixel_shift(oracle_lib):
    rng = np.random.default_rng(4)
    g0, g1 = _texture(rng, 96, 128, shift=(0.30, -0.20))
    v = oracle_lib.lucas_kanade(g0[None], g1[None])
    assert len(v) >= 20
    assert abs(np.median(v[:, 2]) - 0.30) < 0.05 and abs(np.median(v[:, 3]) + 0.20) < 0.05
    # corners: integer coordinates, inside the 1-pixel border, at least minDistance apart, at most maxCorners
    pts = oracle_lib.good_features(g0)
    assert len(pts) <= 100 and np.all(pts == np.round(pts)) and pts[:, 0].min() >= 1 and pts[:, 1].max() <= 94
    d = np.hypot(pts[:, None, 0] - pts[None, :, 0], pts[:, None, 1] - pts[None, :, 1]) + np.eye(len(pts)) * 100
    assert d.min() >= 7
    eig = oracle_lib.min_eig(g0)
    vals = eig[pts[:, 1].astype(int), pts[:, 0].astype(int)]
    assert np.all(np.diff(vals) <= 0) and vals[-1] > 0.3 * eig.max()


def test_lucas_kanade_edge_cases(oracle_lib):
    flat = np.full((1, 64, 64), 90, np.uint8)
    assert len(oracle_lib.lucas_kanade(flat, flat)) == 0            # nothing to track -> caller substitutes the sentinel
    rng = np.random.default_rng(5)
    g0, _ = _texture(rng, 64, 64)
    v = oracle_lib.lucas_kanade(g0[None], g0[None])
    assert len(v) > 0 and np.all(v[:, 2:] == 0)                      # identical frames -> zero flow
    small = rng.integers(0, 255, (1, 20, 24)).astype(np.uint8)       # pyramid cannot go below the 15-px window
    oracle_lib.lucas_kanade(small, small)


def test_flow_building_blocks_against_independent_scipy_math(oracle_lib):
    """pyrDown, the Shi-Tomasi response and the Scharr-based LK normal equations re-derived with scipy / float64."""
    from scipy import ndimage
    rng = np.random.default_rng(7)
    g0, _ = _texture(rng, 40, 56)
    # pyrDown = 5x5 binomial blur (BORDER_REFLECT_101 == scipy 'mirror'), every second pixel, round half up
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    blur = ndimage.correlate1d(ndimage.correlate1d(g0.astype(np.int64), k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
    assert np.array_equal(oracle_lib.pyr_down(g0), ((blur[::2, ::2] + 128) >> 8).astype(np.uint8))
    # cornerMinEigenVal: smaller eigenvalue of the 7x7 box-summed Sobel structure tensor, scaled by (1/(4*7*255))^2
    f = g0.astype(np.float64)
    dx = ndimage.correlate1d(ndimage.correlate1d(f, [-1, 0, 1], axis=1, mode="mirror"), [1, 2, 1], axis=0, mode="mirror")
    dy = ndimage.correlate1d(ndimage.correlate1d(f, [-1, 0, 1], axis=0, mode="mirror"), [1, 2, 1], axis=1, mode="mirror")
    box = lambda a: ndimage.uniform_filter(a, 7, mode="mirror") * 49.0
    sc = (1.0 / (4 * 7 * 255.0)) ** 2
    a, b, c = box(dx * dx) * sc, box(dx * dy) * sc, box(dy * dy) * sc
    lam = 0.5 * (a + c) - np.sqrt((0.5 * (a - c)) ** 2 + b * b)
    got = oracle_lib.min_eig(g0).astype(np.float64)
    assert np.max(np.abs(got - lam)) <= 2e-6 * max(1.0, lam.max())
    # the box-filter border of the tensor is the tensor AT the mirrored pixel (not the tensor of a mirrored image):
    # uniform_filter(mode='mirror') on the product images is exactly that, so the agreement above covers the border.


def _canonical_population(oracle_lib, st, c_dim, w, h, ch, genomes, cfg, wts):
    from evolutionary_illusion_generator_amd import grids
    from oracle import pipeline, scores
    grid = grids.create_grid(st, w, h, 10)
    imgs = np.stack([pipeline.render_chw(g, cfg, grid, c_dim, w, h) for g in genomes])
    frames, vecs, fits = [], [], []
    for im in imgs:
        fr = oracle_lib.prednet_rollout(wts, ch, w, h, im, n_repeat=20, n_ext=1)
        v = oracle_lib.lucas_kanade(fr[19], fr[20])
        frames.append(fr[19:21]); vecs.append(v); fits.append(scores.fitness_from_vectors(st, v.astype(np.float64), w, h))
    return imgs, np.stack(frames), vecs, np.asarray(fits)


def test_fitness_deviation_under_the_reference_element_order_is_explained_genome_by_genome(oracle_lib):
    """BASELINE.json north_star: fitness within 1e-4 relative of the reference CPU path.  The canonical order of the oracle /
    HIP kernels against the element-wise order of the reference's own ConvLSTM (oracle/prednet_torch.py
    order="chainer": separate convolution tensors added left to right, un-fused gate products, sigmoid = tanh(x/2)/2 + 1/2,
    plain unpool -> 9-tap; im2col + matmul convolutions as chainer's CPU path runs them).  Every genome is classified
    (oracle/classify.py): identical frames => identical fitness; a genome outside 1e-4 must have differing frames (all +-1)
    AND its deviation must be reproduced by ONE of those byte flips applied to the canonical frames alone -- i.e. it is the
    conditioning of the reference's fitness function (uint8 stage boundary, relative corner threshold, hard vector
    thresholds), not an implementation error.  configs[1] shape, two populations; population seed 5 contains such a genome
    (2 flipped bytes of 38 400, same tracked corners, 3e-4)."""
    from evolutionary_illusion_generator_amd import synth, weights
    from oracle import classify
    from oracle.prednet_torch import PredNetTorch
    st, c_dim, w, h, ch = 1, 1, 160, 120, [1, 16, 32, 64]
    cfg = synth.make_config(2, 1)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=0)
    net = PredNetTorch(wts, ch, w, h, conv="matmul", order="chainer")
    genomes = [g for _, g in synth.make_population(28, cfg, seed=5)][12:] + [g for _, g in synth.make_population(8, cfg, seed=0)]
    imgs, frames, vecs, fits = _canonical_population(oracle_lib, st, c_dim, w, h, ch, genomes, cfg, wts)
    # the same question against the torch-free C statement of the reference order (host-independent apart from libm's tanh)
    s2, _ = classify.population_report(st, w, h, imgs[:12], frames[:12], vecs[:12], fits[:12], oracle_lib.PredNetC(wts, ch, w, h))
    assert s2["max_byte_diff"] <= 1 and s2["max_rel_identical"] <= 1e-12 and s2["outside_1e-4_unexplained"] == 0, s2
    s, rows = classify.population_report(st, w, h, imgs, frames, vecs, fits, net)
    print("\n%s" % s)
    assert s["nonzero_both"] >= 8 and s["zero_on_one_side_only"] == 0
    assert s["max_byte_diff"] <= 1 and s["byte_flip_rate"] < 1e-4
    assert s["max_rel_identical"] <= 1e-12                      # same frames -> same vectors -> same fitness
    assert s["outside_1e-4_unexplained"] == 0, s["outside_1e-4_detail"]
    if s["outside_1e-4"] == 0:  # (the flips depend on the host's sgemm blocking: seen with 2 flipped bytes on the build container)
        import warnings
        warnings.warn("population seed 5 genome 23 did not deviate on this host: the attribution branch was not exercised")
    for d in s["outside_1e-4_detail"]:
        assert 1 <= d["flips"] <= 16 and d["explained"], d
    assert s["max_rel"] <= 2e-2
    assert s["within_1e-4"] >= 0.9 * s["genomes"] and s["within_1e-4_of_nonzero_both"] >= 0.85 * s["nonzero_both"]  # (measured here: 23 of 24, 13 of 14)


def test_attribution_accepts_a_constructed_single_byte_deviation_and_rejects_unrelated_ones(oracle_lib):
    """the classifier's positive branch on a CONSTRUCTED case, independent of which bytes the host's BLAS happens to flip.  A genome's canonical frame
    pair, and an "other implementation" that is the same pair with ONE byte changed where that moves the fitness: classify.attribute must find that byte's effect and
    classify.explained must accept the deviation; a deviation in the OPPOSITE direction, one three times as large, and a claimed deviation whose only flipped byte
    does not move the fitness must all be rejected."""
    from evolutionary_illusion_generator_amd import synth, weights
    from oracle import classify, scores
    st, c_dim, w, h, ch = 1, 1, 160, 120, [1, 16, 32, 64]
    cfg = synth.make_config(2, 1)
    wts = weights.synthetic_prednet_weights(ch, w, h, seed=0)
    genomes = [g for _, g in synth.make_population(8, cfg, seed=0)]
    imgs, frames, vecs, fits = _canonical_population(oracle_lib, st, c_dim, w, h, ch, genomes, cfg, wts)
    k = int(np.argmax(fits != 0))
    assert fits[k] != 0
    ours = frames[k]
    fit_of = lambda fr: (lambda v: (v, scores.fitness_from_vectors(st, v.astype(np.float64), w, h)))(oracle_lib.lucas_kanade(fr[0], fr[1]))
    # a byte whose +-1 flip moves the fitness: search the 7 x 7 neighbourhood of the first tracked corners in the second frame
    found = None
    for x, y in np.asarray(vecs[k])[:6, :2]:
        for dy in range(-3, 4):
            for dx in range(-3, 4):
                yy, xx = int(round(y)) + dy, int(round(x)) + dx
                if not (0 <= yy < h and 0 <= xx < w):
                    continue
                other = ours.copy()
                other[1, 0, yy, xx] = other[1, 0, yy, xx] + 1 if other[1, 0, yy, xx] < 255 else 254
                v2, f2 = fit_of(other)
                if f2 != 0 and abs(f2 - fits[k]) / abs(fits[k]) > 1.5e-4:
                    found = (other, v2, f2)
                    break
            if found:
                break
        if found:
            break
    assert found, "no single byte near the tracked corners moves the fitness by 1.5e-4: the fixture population changed?"
    other, v2, f2 = found
    r = classify.classify(st, ours, vecs[k], float(fits[k]), other, v2, float(f2))
    assert r["flips"] == 1 and r["max_byte_diff"] == 1 and r["rel"] > 1e-4
    r["fit_ours"], r["fit_other"] = float(fits[k]), float(f2)
    r["single_lsb_effects"] = [float(e) for e in classify.attribute(st, w, h, ours, other, float(fits[k]))]
    assert len(r["single_lsb_effects"]) == 1
    dev = (f2 - fits[k]) / abs(fits[k])
    assert abs(r["single_lsb_effects"][0] - dev) <= 1e-12 * max(1.0, abs(dev))   # the flip, applied alone, IS the deviation
    assert classify.explained(r)
    # the same flipped byte cannot explain a deviation the other way, nor one three times as large
    for claimed in (fits[k] - (f2 - fits[k]), fits[k] + 3.0 * (f2 - fits[k])):
        bad = dict(r, fit_other=float(claimed))
        assert not classify.explained(bad), claimed
    # ... and a byte that does not move the fitness explains nothing
    flat = ours.copy()
    yy, xx = (2, 2) if abs(float(np.asarray(vecs[k])[0, 1]) - 2) > 20 else (h - 3, w - 3)
    flat[0, 0, yy, xx] = flat[0, 0, yy, xx] + 1 if flat[0, 0, yy, xx] < 255 else 254
    eff = [float(e) for e in classify.attribute(st, w, h, ours, flat, float(fits[k]))]
    assert len(eff) == 1
    if abs(eff[0]) < 1e-7:   # (a corner that far from every feature: the usual case)
        assert not classify.explained(dict(r, single_lsb_effects=eff, fit_other=float(fits[k] * (1 + 1e-3))))
    # no attribution at all is never an explanation
    assert not classify.explained(dict(r, single_lsb_effects=[]))


def test_reference_order_is_stated_twice_and_deviates_from_the_canonical_one_by_ulps(oracle_lib):
    """The reference's element-wise ConvLSTM order (separate convolution tensors, plain unpool -> 9-tap, un-fused products,
    sigmoid = tanh(x/2)/2 + 1/2) in C (eig_oracle.c: lstm_reference_order, fma-chain convolutions, libm tanh) and in torch
    (prednet_torch order="chainer", im2col + sgemm, vectorised tanh): two independent statements that agree to fp32 round-off
    on the float predictions of all 22 steps, and both sit within 1e-6 of the canonical arithmetic -- the whole difference
    between the orders is a handful of +-1 bytes at quantisation boundaries (classified genome by genome elsewhere)."""
    from evolutionary_illusion_generator_amd import weights
    from oracle.prednet_torch import PredNetTorch
    for ch, w, h, seed in (([1, 16, 32, 64], 160, 120, 0), ([3, 12, 24, 48], 64, 64, 1)):
        wts = weights.synthetic_prednet_weights(ch, w, h, seed=seed)
        img = (np.random.default_rng(seed).random((ch[0], h, w)) * 255).astype(np.uint8)
        canon, f0 = oracle_lib.prednet_rollout(wts, ch, w, h, img, 20, 2, return_float=True)
        ref_c, f1 = oracle_lib.prednet_rollout(wts, ch, w, h, img, 20, 2, return_float=True, order="chainer")
        ref_t, f2 = PredNetTorch(wts, ch, w, h, conv="matmul", order="chainer").rollout(img[None], 20, 2)
        fr_c, _ = oracle_lib.PredNetC(wts, ch, w, h).rollout(img[None], 20, 2)
        assert np.array_equal(fr_c[0], ref_c)
        assert np.abs(f1 - f2[0]).max() <= 2e-6 and np.abs(f1 - f0).max() <= 2e-6
        for a, b in ((canon, ref_c), (ref_c, ref_t[0]), (canon, ref_t[0])):
            d = np.abs(a.astype(int) - b.astype(int))
            assert d.max() <= 1 and (d != 0).mean() <= 1e-4
    with pytest.raises(KeyError):
        oracle_lib.prednet_rollout(wts, ch, w, h, img, 1, 0, order="cudnn")


def test_classify_tells_identical_smooth_and_cliff_apart():
    """oracle/classify.py on constructed inputs: equal frames; a byte flip with the same tracked features; a dropped corner;
    a vector pushed across the plausibility limit (fitness_calculator.py:18-27)."""
    from oracle import classify
    f = np.zeros((2, 1, 8, 8), np.uint8)
    g = f.copy(); g[0, 0, 3, 3] = 1
    v = np.array([[10.0, 12.0, 0.1, 0.0], [30.0, 40.0, 0.0, 0.29]])
    assert classify.classify(1, f, v, 0.5, f, v, 0.5)["kind"] == "identical"
    v2 = v.copy(); v2[0, 2] = 0.1001
    r = classify.classify(1, f, v, 0.5, g, v2, 0.50001)
    assert r["kind"] == "smooth" and r["flips"] == 1 and abs(r["rel"] - 2e-5) < 1e-6
    assert classify.classify(1, f, v, 0.5, g, v[:1], 0.4)["kind"] == "cliff"                      # a corner dropped
    v3 = v.copy(); v3[1, 3] = 0.31                                                                   # norm crosses 0.3
    r = classify.classify(1, f, v, 0.5, g, v3, 0.45)
    assert r["kind"] == "cliff" and r["same_corners"] and not r["same_kept"]
    assert classify.classify(1, f, v, 0.0, g, v, 0.3)["rel"] == float("inf")
    s = classify.summarize([classify.classify(1, f, v, 0.5, f, v, 0.5), r], f.size)
    assert s["cliff_genomes"] == 1 and s["identical_frames"] == 1 and s["outside_1e-4"] == 1
    # explained(): SIGNED single-byte effects (a sum of absolute effects explains almost anything)
    ex = classify.explained
    base = {"flips": 3, "fit_ours": 0.5, "fit_other": 0.5 * (1 + 4e-4)}
    assert ex(dict(base, single_lsb_effects=[4e-4, 1e-6, -2e-6]))            # one byte IS the deviation
    assert ex(dict(base, single_lsb_effects=[2.1e-4, 1e-6, 0.0]))            # at least half of it, in its direction
    assert not ex(dict(base, single_lsb_effects=[1.9e-4, 1e-6, 0.0]))        # less than half and the sum does not get there either
    assert not ex(dict(base, single_lsb_effects=[-4e-4, 1e-6, 0.0]))         # right size, wrong direction
    assert ex(dict(base, single_lsb_effects=[1.5e-4, 1.4e-4, 1.2e-4]))       # no single byte, but they add up to the deviation (+-20 %)
    assert not ex(dict(base, single_lsb_effects=[1.5e-4, -1.4e-4, 1.2e-4]))  # |.| would add up to it; signed they do not
    assert not ex(dict(base, flips=classify.MAX_FLIPS + 1, single_lsb_effects=[4e-4]))  # too many flips to call it a handful
    assert not ex(dict(base))                                                 # nothing attributed
    assert ex({"flips": 2, "fit_ours": 0.0, "fit_other": 0.3, "single_lsb_effects": [0.0, float("inf")]})    # one flip alone crosses the cliff
    assert not ex({"flips": 2, "fit_ours": 0.0, "fit_other": 0.3, "single_lsb_effects": [0.0, 0.0]})
    assert ex({"flips": 2, "fit_ours": 0.3, "fit_other": 0.0, "single_lsb_effects": [1e-5, -1.0]})
    assert not ex({"flips": 2, "fit_ours": 0.3, "fit_other": 0.0, "single_lsb_effects": [1e-5, -0.5]})


def test_hsv_renderer_is_colorsys_per_pixel():
    """oracle/cppn.py: the h,s,v renderer (generate_illusion.py:333-367) = colorsys.hsv_to_rgb on every pixel, bg in all three
    planes, uint8(rgb * 255); out-of-range and non-finite node values do not raise."""
    import colorsys
    from oracle import cppn
    rng = np.random.default_rng(5)
    w, h = 16, 8
    planes = [rng.uniform(-0.5, 1.5, w * h) for _ in range(3)]
    planes[1][:10] = 0.0                      # s == 0 -> gray
    planes[0][10:14] = [np.nan, np.inf, -np.inf, 1e300]
    x_mat = rng.uniform(0, 1, (h, w)); x_mat[0, :5] = -1
    img = cppn.postprocess(planes, x_mat, 3, w, h, bg=1, gradient=2)
    assert img.shape == (h, w, 3) and img.dtype == np.uint8
    assert (img[0, :5] == np.array([255, 0, 0], np.uint8)).all()     # hsv(1,1,1): the reference fills h, s and v with bg
    for i in range(20, w * h):
        if x_mat.reshape(-1)[i] == -1:
            continue
        r = colorsys.hsv_to_rgb(planes[0][i], planes[1][i], planes[2][i])
        assert (img.reshape(-1, 3)[i] == cppn.to_u8(np.array(r) * 255.0)).all()
    for i in range(5, 10):
        v = cppn.to_u8(np.array([planes[2][i] * 255.0]))[0]
        assert (img.reshape(-1, 3)[i] == v).all()


def test_farneback_oracle_recovers_a_translation(oracle_lib):
    """oracle/farneback.c (the unpinned restatement of cv::calcOpticalFlowFarneback): a smooth texture moved by (+2, -1) px
    gives a dense field of (+2, -1) away from the borders, and the sampled vectors sit on the documented grid."""
    from scipy import ndimage
    rng = np.random.default_rng(0)
    H, W = 128, 160
    base = ndimage.gaussian_filter(rng.normal(0, 1, (H + 40, W + 40)), 3.0)
    base = (base - base.min()) / (base.max() - base.min()) * 255

    def crop(dy, dx):
        return ndimage.shift(base, (dy, dx), order=3)[20:20 + H, 20:20 + W].clip(0, 255).astype(np.uint8)

    fl = oracle_lib.farneback_flow(crop(0, 0), crop(-1.0, 2.0))
    core = fl[30:-30, 30:-30]
    assert abs(np.median(core[..., 0]) - 2.0) < 0.02 and abs(np.median(core[..., 1]) + 1.0) < 0.02
    assert np.percentile(np.abs(core[..., 0] - 2.0), 90) < 0.1 and np.percentile(np.abs(core[..., 1] + 1.0), 90) < 0.1
    v = oracle_lib.farneback_vectors(fl)
    assert v.shape == (80, 4)                       # 160/16 x 128/16
    assert (v[:3, :2] == [[8, 8], [24, 8], [40, 8]]).all() and (v[10, :2] == [8, 24]).all()
    assert np.array_equal(v[:, 2], fl[v[:, 1].astype(int), v[:, 0].astype(int), 0])
    assert len(oracle_lib.farneback_vectors(np.zeros_like(fl))) == 0   # exactly-zero flow carries no vector
    # identical frames: no motion away from the borders (the last row / column take UpdateMatrices' out-of-range branch)
    z = oracle_lib.farneback_flow(crop(0, 0), crop(0, 0))
    assert np.abs(z[30:-30, 30:-30]).max() < 1e-4 and np.abs(z).max() < 0.5
    # level count and grid step rules
    assert oracle_lib.lib().eig_oracle_fb_levels(256, 256, 3) == 3 and oracle_lib.lib().eig_oracle_fb_levels(120, 160, 3) == 1
    assert oracle_lib.lib().eig_oracle_fb_grid_step(256, 256, 16, 100) == 32
