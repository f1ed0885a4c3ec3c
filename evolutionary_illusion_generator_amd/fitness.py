"""Host-side mirror of the reference's fitness API, running on the HIP engine.

Same names, argument order and result conventions as the reference:

* ``get_fitnesses_neat(structure, population, model_name, config, w, h, channels, id=0, c_dim=3, best_dir=".",
  gradient=1)`` -- /root/reference/generate_illusion.py:478-673; the ``eval_genomes`` closure of
  ``neat_illusion`` (:692-694) calls it once per generation and reads nothing back: the result is delivered by
  setting ``genome.fitness`` on every genome of ``population`` (:623-624).
* ``get_vectors(image_path, model_name, channels, w, h)`` and
  ``calculate_fitness(structure, vectors, image_path, w, h)`` -- /root/reference/fitness_calculator.py:468-548.

What changes underneath: no PNG round trips through ./temp (the uint8 quantisation points they imply are kept
on the device), the whole population is one batched device pass, and with torch.distributed initialised the
population is sharded over the ranks (one rank per GPU) with ONE all-gather of float64 fitness scalars
(RCCL over xGMI when the backend is ``nccl``).

``model_name`` is the chainer npz weight file of the reference (``-m`` flag).  ``"synthetic"`` /
``"synthetic:<seed>"`` selects seeded stand-in weights (weights.synthetic_prednet_weights) because the trained
files are external downloads; a dict of tensors is accepted too.
"""
import math
import os

import time

import numpy as np

from . import grids
from .engine import PAIR_POPULATION, PAIR_SINGLE, Engine, EngineError
from .genome import GenomeBatch
from .grids import StructureType
from .weights import load_chainer_npz, synthetic_prednet_weights

SCALING = 10          # generate_illusion.py:501
DEFAULT_MAX_BATCH = 256

_engines = {}
_dict_digests = {}
FULL_CHECK_EVERY = 16   # a weight dict's full content hash is re-checked on every FULL_CHECK_EVERY-th lookup of that dict (see _weights_key)


def _content_digest(d):
    """Hash of every array of a weight dict (names, shapes, bytes): xxh3 where the module is installed, sha1 otherwise."""
    try:
        import xxhash
        hsh = xxhash.xxh3_128()
    except ImportError:
        import hashlib
        hsh = hashlib.sha1()
    for k in sorted(d):
        a = np.ascontiguousarray(d[k])
        hsh.update(k.encode()); hsh.update(str(a.shape).encode()); hsh.update(str(a.dtype).encode())
        if a.size:   # (a memoryview of a zero-size array cannot be cast)
            hsh.update(a.reshape(-1).view(np.uint8))
    return hsh.hexdigest()


def _freeze(d):
    """The contract for weight DICTS is immutability: every numpy array of a dict handed to the engine is made
    read-only, so that an in-place edit through that array raises instead of being evaluated on the engine that holds the OLD weights.
    -> the arrays frozen here (invalidate_weights() thaws exactly those)."""
    frozen = []
    for k in d:
        a = d[k]
        if isinstance(a, np.ndarray) and a.flags.writeable:
            try:
                a.flags.writeable = False
                frozen.append(a)
            except ValueError:
                pass   # (a view of a foreign buffer that refuses: the probe and the periodic full hash still cover it)
    return frozen


def _evict_digest(old_digest):
    """The engines built for a digest no dict object maps to any more are closed (a dict whose content changed must not leave a second
    full-size engine behind: ADVICE r5)."""
    if any(ent[1] == old_digest for ent in _dict_digests.values()):
        return
    for key in [k for k in _engines if len(k) > 4 and k[4] == ("dict", old_digest)]:
        _engines.pop(key).close()


def invalidate_weights(model_dict):
    """Announce an in-place edit of a weight dict: its arrays become writeable again and its cached digest is dropped, so the next
    lookup hashes the full content (and builds a new engine, closing the old one, if it changed)."""
    ent = _dict_digests.get(id(model_dict))
    if ent is not None and ent[0] is model_dict:
        for a in ent[4]:
            try:
                a.flags.writeable = True
            except ValueError:
                pass
        _dict_digests[id(model_dict)] = (model_dict, ent[1], None, 0, [])   # fingerprint None: the next lookup is a full check


def _weights_key(model_name):
    """Cache key of a weight source WITHOUT reading it: get_fitnesses_neat resolves its engine several times per
    generation, and a 33 MB npz (0.2 s) or a synthetic set (0.6 s) must only be materialised on a cache miss."""
    if isinstance(model_name, dict):
        # Content digest, memoised per dict OBJECT (the table keeps the dict alive, so its id cannot be reused): two equal weight dicts
        # resolve to ONE engine instead of two 9.5 GB ones.  A dict that was MUTATED after its first use must not keep resolving to the
        # engine holding the old weights; the rules are deterministic (no clock):
        #   * the arrays are made READ-ONLY at first use (_freeze): an in-place edit through them raises; invalidate_weights(d) thaws them;
        #   * a cheap fingerprint on EVERY lookup -- per array: buffer address, shape, dtype, the writeable flag and a strided sample of 64
        #     elements: a replaced array, a thawed array or an edit that hits the sample triggers a full hash at once;
        #   * a FULL content hash (xxh3: 5 ms for the 46 MB of the headline network) on every FULL_CHECK_EVERY-th lookup of the dict, which
        #     is what an edit through ANOTHER view of the same memory (the one hole the flag leaves) is caught by.
        # When the digest of a dict object changes, the engine of the old digest is closed unless another dict still maps to it.
        def probe(a):
            a = np.asarray(a)
            flat = a.reshape(-1)
            return (a.__array_interface__["data"][0], a.shape, a.dtype.str, bool(a.flags.writeable), flat[::max(1, flat.size // 64)][:64].tobytes())
        ent = _dict_digests.get(id(model_name))
        if ent is not None and ent[0] is not model_name:
            ent = None
        fp = tuple((k, probe(model_name[k])) for k in sorted(model_name))
        if ent is None or ent[2] != fp or ent[3] + 1 >= FULL_CHECK_EVERY:
            frozen = (ent[4] if ent is not None else []) + _freeze(model_name)
            fp = tuple((k, probe(model_name[k])) for k in sorted(model_name))   # (the flags just changed)
            digest = _content_digest(model_name)
            _dict_digests[id(model_name)] = (model_name, digest, fp, 0, frozen)
            if ent is not None and ent[1] != digest:
                _evict_digest(ent[1])
            return ("dict", digest)
        _dict_digests[id(model_name)] = (ent[0], ent[1], ent[2], ent[3] + 1, ent[4])
        return ("dict", ent[1])
    name = str(model_name)
    if name.startswith("synthetic"):
        return ("synthetic", int(name.split(":")[1]) if ":" in name else 0)
    return ("file", os.path.abspath(name), os.path.getmtime(name))


def _resolve_weights(model_name, channels, w, h):
    if isinstance(model_name, dict):
        return model_name
    name = str(model_name)
    if name.startswith("synthetic"):
        return synthetic_prednet_weights(channels, w, h, seed=int(name.split(":")[1]) if ":" in name else 0)
    return load_chainer_npz(name, channels, w, h)


def _local_device():
    import torch
    if not torch.cuda.is_available():
        raise EngineError("no HIP device visible: the fitness path runs on MI355X only (no CPU fallback)")
    return torch.cuda.current_device()


def get_engine(model_name, w, h, channels, max_batch=None, any_batch=False, **kw):
    """Engine handles are cached per (device, size, channels, weights, device batch): workspaces and packed weights stay in
    HBM across generations.  any_batch: a caller that evaluates a single image (best artefacts, get_vectors) takes whatever
    engine of this shape and these weights already exists instead of building a second full-size one."""
    channels = [int(c) for c in channels]
    dev = _local_device()
    wk, kwk = _weights_key(model_name), tuple(sorted(kw.items()))
    if any_batch and max_batch is None:
        for k, e in _engines.items():
            if len(k) == 7 and k[:5] == (dev, w, h, tuple(channels), wk) and k[6] == kwk:
                return e
    mb = int(max_batch or int(os.environ.get("EIGEN_MAX_BATCH", DEFAULT_MAX_BATCH)))
    key = (dev, w, h, tuple(channels), wk, mb, kwk)
    eng = _engines.get(key)
    if eng is None:
        eng = Engine(w, h, channels, mb, device=dev, **kw)
        eng.set_weights(_resolve_weights(model_name, channels, w, h))
        eng._weights_ref = model_name  # a dict is keyed by id(): keep it alive so the id cannot be reused
        eng._grid_key = None
        _engines[key] = eng
    return eng


def clear_engines():
    for e in _engines.values():
        e.close()
    _engines.clear()
    _dict_digests.clear()


def leaf_planes(structure, w, h, n_inputs=2):
    """CPPN input planes.  The reference feeds x_mat and y_mat only (generate_illusion.py:374-378); for configs with
    num_inputs = 4 (neat_configs/default.txt:49), where PyTorch-NEAT would assert, the build-defined extra leaves are
    r = sqrt(x^2 + y^2) and a constant 1 (BASELINE.json north_star: "(x, y, r, bias)")."""
    g = grids.create_grid(structure, w, h, SCALING)
    x, y = g["x_mat"], g["y_mat"]
    if n_inputs == 2:
        return [x, y]
    if n_inputs == 4:
        return [x, y, np.sqrt(x * x + y * y), np.ones_like(x)]
    raise ValueError("CPPNs with %d inputs are not defined (2: x, y; 4: x, y, r, bias)" % n_inputs)


def _set_grid(eng, structure, w, h, n_inputs=2):
    key = (int(structure), w, h, n_inputs)
    if eng._grid_key != key:
        eng.set_grid(leaf_planes(structure, w, h, n_inputs))
        eng._grid_key = key


# ----------------------------------------------------------------------------------------------- sharding
def shard_bounds(n_items, world_size, rank):
    """Contiguous slice of the population list owned by `rank` (list order matters: scores[i] and the
    'last maximal genome wins' tie-break, generate_illusion.py:623-628)."""
    per = int(math.ceil(n_items / float(world_size))) if n_items else 0
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per), per


def _dist():
    try:
        import torch.distributed as dist
    except Exception:
        return None
    # EIGEN_DIST_SINGLE=1: take the collective path even in a group of ONE rank (tests: RCCL on a single-GPU box)
    min_world = 1 if os.environ.get("EIGEN_DIST_SINGLE") == "1" else 2
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() >= min_world) else None


# What the last sharded_map call cost on THIS rank and, from the extras that rode in its all-gather, on every rank:
# {"local_ms": [R] each rank's evaluate() wall time, "collective_ms": this rank's all-gather incl. device round trip}.
# bench.py --gpus N prints it so that a multi-GPU run explains its own efficiency (stragglers vs. collective latency).
LAST_SHARD_STATS = {}


def sharded_map(n_items, evaluate, group=None, extra=None):
    """Evaluate items [lo, hi) of this rank with ``evaluate(lo, hi) -> float64 array`` and all-gather the
    scalars so that every rank returns the full float64 vector.  One collective per call; no data-path exchange.
    ``extra``: float64 value(s) per rank that ride in the same collective (a digest, a timing) -- a scalar or a
    sequence; when given the result is ``(vector, extras[R])`` (``extras[R, k]`` for a sequence of k).  Every rank's
    evaluate() wall time always rides along (LAST_SHARD_STATS)."""
    import time
    dist = _dist()
    scalar_extra = extra is not None and np.ndim(extra) == 0
    ex = np.zeros(0) if extra is None else np.atleast_1d(np.asarray(extra, dtype=np.float64))
    if dist is None:
        t0 = time.perf_counter()
        res = np.asarray(evaluate(0, n_items), dtype=np.float64)
        LAST_SHARD_STATS.clear()
        LAST_SHARD_STATS.update(local_ms=[1e3 * (time.perf_counter() - t0)], collective_ms=0.0)
        return res if extra is None else (res, ex.copy() if scalar_extra else ex[None, :].copy())
    import torch
    R, r = dist.get_world_size(group), dist.get_rank(group)
    lo, hi, per = shard_bounds(n_items, R, r)
    slot = per + len(ex) + 1
    local = np.zeros(slot, dtype=np.float64)
    t0 = time.perf_counter()
    if hi > lo:
        local[:hi - lo] = evaluate(lo, hi)
    t1 = time.perf_counter()
    local[per:per + len(ex)] = ex
    local[slot - 1] = 1e3 * (t1 - t0)
    t = torch.from_numpy(local)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    out = torch.empty(slot * R, dtype=torch.float64, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    full = out.cpu().numpy().reshape(R, slot)
    LAST_SHARD_STATS.clear()
    LAST_SHARD_STATS.update(local_ms=full[:, slot - 1].tolist(), collective_ms=1e3 * (time.perf_counter() - t1))
    res = np.zeros(n_items, dtype=np.float64)
    for k in range(R):
        a, b, _ = shard_bounds(n_items, R, k)
        res[a:b] = full[k, :b - a]
    if extra is None:
        return res
    exs = full[:, per:per + len(ex)].copy()
    return res, (exs[:, 0] if scalar_extra else exs)


def _broadcast_bytes(payload, src=0, group=None):
    """bytes of rank ``src`` -> every rank (two broadcasts: length, then the buffer; device tensors under nccl = RCCL).
    payload None on ``src`` = "I failed before I had anything to send": the length goes out as -1 and EVERY rank gets
    None back straight after the first broadcast, so nobody is left waiting in the second one."""
    import torch
    dist = _dist()
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    me = dist.get_rank(group)
    n = torch.tensor([(-1 if payload is None else len(payload)) if me == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src, group=group)
    if int(n.item()) < 0:
        return None
    if me == src:
        buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src, group=group)
    return payload if me == src else buf.cpu().numpy().tobytes()


# ----------------------------------------------------------------------------------------------- the API
# Optical-flow method of the population path: "lk" (what the reference calls: lucas_kanade, generate_illusion.py:549-550) or
# "farneback" (dense flow sampled on a grid, csrc/farneback_kernels.h).  get_fitnesses_neat keeps the reference's signature,
# so the choice is a module setting; evaluate_population also takes it as an argument.
FLOW_METHOD = "lk"

# Where a multi-rank run takes its genomes from (one rank per GPU, torch.distributed initialised):
#   "rank0"      rank 0 flattens ITS population once and broadcasts the ~1 KB/genome wire arrays; every rank evaluates its
#                contiguous slice of THAT batch.  Rank 0 is authoritative (north_star: "all-gather ... back to the rank-0
#                reproduction step"), so the result is right even when the ranks' own NEAT runs were never seeded alike
#                (the reference and neat-python never seed `random`).
#   "replicated" every rank flattens only its own slice of its own copy of the population (no broadcast); a digest of the
#                population rides in the fitness all-gather and a mismatch raises -- the ranks MUST run identically seeded.
GENOME_SOURCE = os.environ.get("EIGEN_GENOME_SOURCE", "rank0")


def _engine_for(model_name, w, h, channels, max_batch, flow):
    flow = flow or FLOW_METHOD
    return get_engine(model_name, w, h, channels, max_batch=max_batch, **({} if flow == "lk" else {"flow": flow}))


def evaluate_batch(structure, gb, n_in, model_name, w, h, channels, c_dim=3, gradient=1, bg=1,
                   pairing=PAIR_POPULATION, max_batch=None, flow=None):
    """Fitness (float64 array) of an already flattened genome batch on the local GPU, in chunks of max_batch."""
    channels = [int(c) for c in channels]
    if channels[0] != c_dim:
        raise ValueError("channels[0]=%d but c_dim=%d: PredNet's input channels are the image channels" % (channels[0], c_dim))
    eng = _engine_for(model_name, w, h, channels, max_batch, flow)
    _set_grid(eng, structure, w, h, n_in)
    out = np.zeros(gb.n_genomes, dtype=np.float64)
    for i in range(0, gb.n_genomes, eng.max_batch):
        j = min(gb.n_genomes, i + eng.max_batch)
        out[i:j] = eng.eval_population(gb.slice(i, j), int(structure), bg=bg, gradient=gradient, pairing=pairing)
    return out


def evaluate_population(structure, genomes, model_name, config, w, h, channels, c_dim=3, gradient=1, bg=1,
                        pairing=PAIR_POPULATION, max_batch=None, flow=None):
    """Fitness (float64 array) of a list of genome objects on the local GPU, in chunks of max_batch."""
    n_in = len(config.genome_config.input_keys)
    c_out = c_dim if gradient == 1 else 1
    gb = GenomeBatch(genomes, config, c_out, n_leaves=n_in)  # Q6: the first c_dim outputs are rendered
    return evaluate_batch(structure, gb, n_in, model_name, w, h, channels, c_dim=c_dim, gradient=gradient, bg=bg,
                          pairing=pairing, max_batch=max_batch, flow=flow)


_warned_diverged = [False]


def population_digest(genomes):
    """Cheap fingerprint of a population (keys, sizes and one weight / bias per genome; < 0.2 ms for 256 genomes): two
    NEAT runs that were not seeded alike differ in it from the first generation on."""
    import zlib
    parts = []
    for g in genomes:
        c = next(iter(g.connections.values()), None)
        n = next(iter(g.nodes.values()), None)
        parts.append((getattr(g, "key", None), len(g.nodes), len(g.connections),
                      None if c is None else (c.weight, c.enabled), None if n is None else n.bias))
    return float(zlib.crc32(repr(parts).encode()))


def population_fitness(structure, genomes, model_name, config, w, h, channels, c_dim=3, gradient=1, max_batch=None,
                       flow=None, source=None):
    """Fitness of the WHOLE population list on every rank: the population is sharded over the ranks (contiguous slices in
    list order), every rank evaluates its slice on its GPU, ONE all-gather of float64 scalars returns the full vector
    (SURVEY 8(e)).  Single process: plain evaluate_population."""
    dist = _dist()
    kw = dict(c_dim=c_dim, gradient=gradient, max_batch=max_batch, flow=flow)
    if dist is None:
        return evaluate_population(structure, genomes, model_name, config, w, h, channels, **kw)
    source = source or GENOME_SOURCE
    if source == "replicated":
        scores, digests = sharded_map(len(genomes), lambda lo, hi: evaluate_population(
            structure, genomes[lo:hi], model_name, config, w, h, channels, **kw), extra=population_digest(genomes))
        if not np.all(digests == digests[0]):
            raise EngineError("the ranks hold different populations (digests %s): with EIGEN_GENOME_SOURCE=replicated every "
                              "rank must run the same seeded NEAT (random.seed(<const>) before neat.Population); or use the "
                              "default source 'rank0'" % digests.tolist())
        return scores
    if source != "rank0":
        raise ValueError("GENOME_SOURCE must be 'rank0' or 'replicated', got %r" % (source,))
    n_in = len(config.genome_config.input_keys)
    payload, err = b"", None
    if dist.get_rank() == 0:
        try:  # a cyclic / invalid genome raises HERE, on rank 0 only: tell the others instead of leaving them in the broadcast
            payload = GenomeBatch(genomes, config, c_dim if gradient == 1 else 1, n_leaves=n_in).to_bytes()
        except Exception as e:  # noqa: BLE001
            payload, err = None, e
    wire = _broadcast_bytes(payload)
    if wire is None:
        if err is not None:
            raise err
        raise EngineError("rank 0 could not flatten its population (see its traceback): no genomes were broadcast")
    gb = GenomeBatch.from_bytes(wire)
    scores, digests = sharded_map(gb.n_genomes, lambda lo, hi: evaluate_batch(structure, gb.slice(lo, hi), n_in, model_name, w, h, channels, **kw),
                                  extra=population_digest(genomes))
    # rank 0 is authoritative; a rank whose OWN population differs from rank 0's (same length or not) is told so once
    me = dist.get_rank()
    if me != 0 and digests[me] != digests[0] and not _warned_diverged[0]:
        print("eigen: rank %d's population is not rank 0's (digest %.0f vs %.0f): rank 0 is authoritative and its fitness values "
              "are assigned here by list index; seed `random` identically on every rank (INTEGRATION.md section 1) to keep the "
              "replicas in step" % (me, digests[me], digests[0]))
        _warned_diverged[0] = True
    return scores


def get_fitnesses_neat(structure, population, model_name, config, w, h, channels,
                       id=0, c_dim=3, best_dir=".", gradient=1):
    """Drop-in for generate_illusion.get_fitnesses_neat: sets ``genome.fitness`` for every (id, genome) pair."""
    print("Calculating fitnesses of populations: ", len(population))
    genomes = [g for _, g in population]
    scores = population_fitness(structure, genomes, model_name, config, w, h, channels, c_dim=c_dim, gradient=gradient)
    if len(scores) != len(population):  # source "rank0" on a rank whose own NEAT run diverged from rank 0's
        if not _warned_diverged[0]:
            print("eigen: this rank's population (%d genomes) is not rank 0's (%d): rank 0 is authoritative; seed `random` "
                  "identically on every rank to keep the replicas in step" % (len(population), len(scores)))
            _warned_diverged[0] = True
        scores = np.resize(scores, len(population)) if len(scores) else np.zeros(len(population))
    best_score, best_illusion, best_genome = 0, 0, None
    for i, (_, genome) in enumerate(population):
        genome.fitness = float(scores[i])
        if scores[i] >= best_score:  # last maximal genome wins (generate_illusion.py:625)
            best_illusion, best_score, best_genome = i, float(scores[i]), genome
    print("best", best_score, best_illusion)
    dist = _dist()
    if best_genome is not None and best_dir is not None and (dist is None or dist.get_rank() == 0):
        try:
            save_best_artifacts(structure, best_genome, model_name, config, w, h, channels, c_dim, best_dir, gradient)
        except ImportError:
            pass  # PIL missing: artefacts are cosmetic, fitness is the contract
    return scores


def render_images(structure, genomes, model_name, config, w, h, channels, c_dim=3, gradient=1, bg=1, max_batch=None):
    """uint8 [n, C, H, W] images of get_image_from_cppn (generate_illusion.py:372-460) for a list of genomes."""
    import torch
    # the engine the fitness path uses (any device batch of it: a render needs no workspace of its own)
    eng = get_engine(model_name, w, h, [int(c) for c in channels], max_batch=max_batch, any_batch=True,
                     **({} if FLOW_METHOD == "lk" else {"flow": FLOW_METHOD}))
    n_in = len(config.genome_config.input_keys)
    _set_grid(eng, structure, w, h, n_in)
    out = []
    for i in range(0, len(genomes), eng.max_batch):
        chunk = genomes[i:i + eng.max_batch]
        gb = GenomeBatch(chunk, config, c_dim if gradient in (1, 2) else 1, n_leaves=n_in)
        d = torch.empty((len(chunk), c_dim, h, w), dtype=torch.uint8, device="cuda")
        eng.render_cppn(gb, d, bg=bg, gradient=gradient)
        torch.cuda.synchronize()
        out.append(d.cpu().numpy())
    return np.concatenate(out)


def get_equilum_image_from_cppn(structure, genome, model_name, config, w, h, channels, bg=1):
    """The reference's HSV renderer (generate_illusion.py:333-367; dead there: its colorsys call raises): output nodes 0..2 are
    h, s, v, converted per pixel on the device (cppn_render_kernel mode 4).  Returns a PIL image like the reference does.
    ``structure`` stands for the reference's ``inputs`` grid dict (the grid is built and cached on the device)."""
    from PIL import Image
    chw = render_images(structure, [genome], model_name, config, w, h, channels, c_dim=3, gradient=2, bg=bg)[0]
    return Image.fromarray(np.ascontiguousarray(chw.transpose(1, 2, 0)))


def save_best_artifacts(structure, genome, model_name, config, w, h, channels, c_dim, best_dir, gradient,
                        enhanced_size=800, flow_scale=20.0):
    """best.png, best_black_bg.png, best_flow.png and the 800x800 enhanced.png of generate_illusion.py:650-673.
    The renders run on the device (the reference's pure-Python enhanced_image_grid + per-pixel loops take ~12 s);
    best_flow.png is a plain overlay of the flow vectors (the upstream drawing style is not pinned anywhere)."""
    import torch
    from PIL import Image, ImageDraw
    os.makedirs(best_dir, exist_ok=True)
    mode = "RGB" if c_dim == 3 else "L"
    to_pil = lambda img: Image.fromarray(img.transpose(1, 2, 0) if c_dim == 3 else img[0], mode)
    white = render_images(structure, [genome], model_name, config, w, h, channels, c_dim, gradient, 1)[0]
    to_pil(white).save(os.path.join(best_dir, "best.png"), "PNG")
    black = render_images(structure, [genome], model_name, config, w, h, channels, c_dim, gradient, 0)[0]
    to_pil(black).save(os.path.join(best_dir, "best_black_bg.png"), "PNG")
    # flow overlay: population pairing (prediction@20 -> first extension) and flow method, as the fitness used
    eng = get_engine(model_name, w, h, [int(c) for c in channels], any_batch=True, **({} if FLOW_METHOD == "lk" else {"flow": FLOW_METHOD}))
    d = torch.from_numpy(np.ascontiguousarray(white[None])).cuda()
    _, vecs = eng.eval_images(d, 1, int(structure), pairing=PAIR_POPULATION)
    np.save(os.path.join(best_dir, "best_flow_vectors.npy"), np.asarray(vecs[0], dtype=np.float32).reshape(-1, 4))
    flow = to_pil(white).convert("RGB")
    draw = ImageDraw.Draw(flow)
    for x, y, dx, dy in vecs[0]:
        draw.line([(x, y), (x + flow_scale * dx, y + flow_scale * dy)], fill=(255, 0, 0), width=1)
        draw.ellipse([x - 1, y - 1, x + 1, y + 1], fill=(255, 255, 0))
    flow.save(os.path.join(best_dir, "best_flow.png"), "PNG")
    # enhanced image: 3x3 + 2x2 circles on a render-only engine (one PredNet layer keeps its workspaces tiny); written for
    # every structure, as the reference does (generate_illusion.py:665-671; Bands / Free get ring coordinates with theta 0)
    if enhanced_size:
        e = enhanced_size
        key = ("render", _local_device(), e, c_dim)
        reng = _engines.get(key)
        if reng is None:
            reng = Engine(e, e, [c_dim], 1, device=_local_device())
            reng._grid_key = None
            _engines[key] = reng
        gk = ("enhanced", int(structure), e)
        if reng._grid_key != gk:
            g = grids.enhanced_image_grid(e, e, structure)
            reng.set_grid([g["x_mat"], g["y_mat"]])
            reng._grid_key = gk
        gb = GenomeBatch([genome], config, c_dim if gradient == 1 else 1, n_leaves=len(config.genome_config.input_keys))
        dimg = torch.empty((1, c_dim, e, e), dtype=torch.uint8, device="cuda")
        reng.render_cppn(gb, dimg, bg=1, gradient=gradient)
        torch.cuda.synchronize()
        to_pil(dimg.cpu().numpy()[0]).save(os.path.join(best_dir, "enhanced.png"), "PNG")


def _read_image_chw(image_path, c_dim, w, h):
    """read_image of chainer_prednet (PIL -> CHW uint8; gray via convert('L') when c_dim == 1), centre crop."""
    from PIL import Image
    im = Image.open(image_path)
    im = im.convert("L") if c_dim == 1 else im.convert("RGB")
    a = np.asarray(im)
    if a.ndim == 2:
        a = a[:, :, None]
    top, left = (a.shape[0] - h) // 2, (a.shape[1] - w) // 2
    if top < 0 or left < 0:
        raise ValueError("image %s is %dx%d, smaller than the requested %dx%d" % (image_path, a.shape[1], a.shape[0], w, h))
    return np.ascontiguousarray(a[top:top + h, left:left + w].transpose(2, 0, 1))


def get_vectors(image_path, model_name, channels, w, h):
    """fitness_calculator.get_vectors: flow vectors original image -> 2nd extended prediction, np.ndarray (n, 4)
    float, or ``[None]`` when Lucas-Kanade tracks nothing (fitness_calculator.py:496-502).  ``image_path`` may
    also be a uint8 array (H,W[,C])."""
    import torch
    channels = [int(c) for c in channels]
    c_dim = channels[0]
    if isinstance(image_path, np.ndarray):
        a = image_path if image_path.ndim == 3 else image_path[:, :, None]
        if a.ndim != 3 or a.shape[2] != c_dim:
            raise ValueError("image array must be (H, W) or (H, W, %d) uint8, got shape %s" % (c_dim, image_path.shape))
        top, left = (a.shape[0] - h) // 2, (a.shape[1] - w) // 2  # centre crop, as for files
        if top < 0 or left < 0:
            raise ValueError("image array is %dx%d, smaller than the requested %dx%d" % (a.shape[1], a.shape[0], w, h))
        img = np.ascontiguousarray(a[top:top + h, left:left + w].transpose(2, 0, 1).astype(np.uint8))
    else:
        img = _read_image_chw(image_path, c_dim, w, h)
    eng = get_engine(model_name, w, h, channels, any_batch=True)
    d = torch.from_numpy(img[None]).cuda()
    _, vecs = eng.eval_images(d, 1, int(StructureType.Free), pairing=PAIR_SINGLE)
    return np.asarray(vecs[0], dtype=np.float64) if len(vecs[0]) else [None]


def calculate_fitness(structure, vectors, image_path, w, h):
    """fitness_calculator.calculate_fitness on the device scorer.  Deviation (Q15): where the reference raises
    UnboundLocalError (too few plausible vectors) this returns 0.0, the value get_fitnesses_neat would assign."""
    import torch
    if int(structure) not in (0, 1, 2, 3):
        raise NameError("name 'good_vectors' is not defined")  # what the reference does for an unknown structure
    return _device_score(structure, vectors, w, h)


def _device_score(structure, vectors, w, h):
    import torch
    v = np.asarray(vectors, dtype=np.float64).reshape(-1, 4) if (len(vectors) and vectors[0] is not None) else np.zeros((0, 4))
    eng = _score_engine(w, h, max(len(v), 1))
    K = eng.K
    if len(v) > K:
        raise EngineError("%d vectors exceed the scorer capacity %d" % (len(v), K))
    vec = np.zeros((1, K, 4), dtype=np.float32)
    vec[0, :len(v)] = v
    dv = torch.from_numpy(vec).cuda()
    dc = torch.tensor([len(v)], dtype=torch.int32, device="cuda")
    df = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.score(int(structure), dv, dc, 1, df, width=int(w), height=int(h))
    torch.cuda.synchronize()
    return float(df.cpu().numpy()[0])


def inside_outside_score(vectors, width, height):
    """fitness_calculator.inside_outside_score (:219-304) on the device scorer.  The reference only reaches it through an
    else branch that raises NameError, so it is offered under its own name rather than through calculate_fitness."""
    v = np.array(vectors, dtype=np.float64).reshape(-1, 4) if (len(vectors) and vectors[0] is not None) else np.zeros((0, 4))
    # The reference indexes numpy arrays of shape (w, h) with i = int(x / step), j = int(y / step) (:235-236): an index past the
    # end raises IndexError, a negative one wraps Python-style.  The device scorer only uses a position to find its cell, so the
    # same semantics are applied here: out-of-range raises, a wrapped vector is moved to the centre of the cell it wraps to.
    step = width / 5
    nw, nh = int(width / step) + 1, int(height / step) + 1
    for r in v:
        for k, n in ((0, nw), (1, nh)):
            c = int(r[k] / step)
            if c >= n or c < -n:
                raise IndexError("index %d is out of bounds for axis %d with size %d" % (c, k, n))
            if c < 0:
                r[k] = (c + n + 0.5) * step
    return _device_score(4, v, width, height)


def _score_engine(w, h, n):
    key = ("score", _local_device())
    eng = _engines.get(key)
    if eng is None:
        eng = Engine(32, 32, [1, 1], 1, device=_local_device(), max_corners=128)  # scorer only: tiny workspaces
        _engines[key] = eng
    return eng


# ------------------------------------------------------------------- stage-level entry points (import shims)
def _aux_engine(kind, w, h, c_dim, **kw):
    """Render-/flow-only engine: one PredNet layer keeps the workspaces tiny; no weights are ever set."""
    key = (kind, _local_device(), w, h, c_dim, tuple(sorted(kw.items())))
    eng = _engines.get(key)
    if eng is None:
        eng = Engine(w, h, [c_dim], 1, device=_local_device(), **kw)
        eng._grid_key = None
        _engines[key] = eng
    return eng


def cppn_node_planes(genome, config, planes, n_outputs=None):
    """Raw float64 output-node values of one genome over arbitrary input planes: what the node objects returned by
    PyTorch-NEAT's create_cppn compute when called as ``node(x=inp_x, y=inp_y)`` (generate_illusion.py:395,406,443).
    planes: list of equal-length float64 arrays (x, y[, r, bias]).  Returns float64 [n_outputs, N]."""
    import torch
    planes = [np.ascontiguousarray(np.asarray(p, dtype=np.float64).reshape(-1)) for p in planes]
    n = planes[0].size
    n_out = int(n_outputs or len(config.genome_config.output_keys))
    eng = _aux_engine("nodes", n, 1, 1)
    eng.set_grid(planes)
    gb = GenomeBatch([genome], config, n_out, n_leaves=len(planes))
    d = torch.empty((1, n_out, n), dtype=torch.float64, device="cuda")
    eng.eval_cppn_nodes(gb, d)
    torch.cuda.synchronize()
    return d.cpu().numpy()[0]


def prednet_predictions(images, model_name, channels, w, h, n_repeat=20, n_ext=2):
    """Quantised prediction frames of test_prednet for a batch of constant-image sequences: uint8
    [n, n_repeat + n_ext, C, H, W] (frame t = prediction after t+1 steps; the last n_ext are the self-fed ones)."""
    import torch
    channels = [int(c) for c in channels]
    images = np.ascontiguousarray(images, dtype=np.uint8)
    if images.ndim != 4 or images.shape[1:] != (channels[0], h, w):
        raise ValueError("images must be uint8 [n, %d, %d, %d] (n, C, H, W), got %s" % (channels[0], h, w, images.shape))
    n_steps = n_repeat + n_ext
    eng = get_engine(model_name, w, h, channels, n_repeat=n_repeat, n_ext=n_ext)
    out = np.empty((len(images), n_steps) + images.shape[1:], dtype=np.uint8)
    for i in range(0, len(images), eng.max_batch):
        chunk = images[i:i + eng.max_batch]
        d = torch.from_numpy(chunk).cuda()
        fr = torch.empty((len(chunk), n_steps) + chunk.shape[1:], dtype=torch.uint8, device="cuda")
        eng.prednet_rollout(d, len(chunk), n_steps, 0, fr)
        torch.cuda.synchronize()
        out[i:i + len(chunk)] = fr.cpu().numpy()
    return out


def flow_vectors(img0, img1, method="lk"):
    """Flow vectors [x, y, dx, dy] (float32 [n, 4]) between two uint8 CHW images of equal size.  method "lk": Lucas-Kanade
    (Optical_Flow_Analyzer lucas_kanade; generate_illusion.py:549-550, fitness_calculator.py:498); "farneback": dense
    Farneback flow sampled on a grid."""
    import torch
    a, b = np.array(img0, dtype=np.uint8, order="C"), np.array(img1, dtype=np.uint8, order="C")  # writable copies for torch
    if a.shape != b.shape or a.ndim != 3:
        raise ValueError("flow_vectors needs two CHW uint8 images of the same shape, got %s and %s" % (a.shape, b.shape))
    c, h, w = a.shape
    eng = _aux_engine("flow", w, h, c, **({} if method == "lk" else {"flow": method}))
    d0, d1 = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    vec = torch.zeros((1, eng.K, 4), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.flow(d0, a.size, d1, b.size, 1, vec, cnt)
    torch.cuda.synchronize()
    return vec.cpu().numpy()[0, :int(cnt.cpu().numpy()[0])]
