"""Flatten NEAT genomes into the engine's wire format (include/eigen_engine.h: eigen_genome_batch).

Host-side counterpart of ``pytorch_neat.cppn.create_cppn`` as the reference calls it
(/root/reference/generate_illusion.py:384-389, 436-441: ``leaf_names=["x","y"]``, no ``output_activation``):
which connections are expressed, in which order their terms are summed, and what a node without inputs
evaluates to are decided here; the HIP kernel (csrc/cppn_kernel.h) only executes the resulting program.

Semantics kept (UPSTREAM-RECALL):
  * ``required_for_output`` reachability runs over ALL connection keys (enabled or not);
  * enabled connections are taken in ``genome.connections`` dict order; a connection is skipped when
    neither end is required, or when its source is an output node;
  * a node with no expressed inputs evaluates to ``torch.full(shape, bias)`` in torch's default dtype
    float32; products ``w * x`` with such a constant are float32 too.  Here such nodes are folded on the
    host with numpy float32 arithmetic and each of their outgoing edges becomes an edge from the constant-1
    leaf whose weight is the float32 product, so the device never sees them.
"""
import numpy as np

ACT_IDS = {"sigmoid": 0, "tanh": 1, "abs": 2, "gauss": 3, "identity": 4, "sin": 5, "relu": 6}


def _required_for_output(inputs, outputs, connections):
    """neat.graphs.required_for_output: layer by layer, t = sources of connections into the set seen so far that are not
    in it yet.  (Only the layer added last can contribute new sources, so each connection is looked at once.)"""
    incoming = {}
    for (a, b) in connections:
        incoming.setdefault(b, []).append(a)
    inputs = set(inputs)
    required = set(outputs)
    seen = set(outputs)
    last = list(seen)
    while True:
        t = set()
        for b in last:
            for a in incoming.get(b, ()):
                if a not in seen:
                    t.add(a)
        if not t:
            break
        layer = t - inputs
        if not layer:
            break
        required |= layer
        seen |= t
        last = t
    return required


def _f32_act(name, x):
    x = np.float32(x)
    with np.errstate(all="ignore"):
        if name == "sigmoid":
            return np.float32(1.0) / (np.float32(1.0) + np.exp(-(np.float32(5) * x)))
        if name == "tanh":
            return np.tanh(np.float32(2.5) * x)
        if name == "abs":
            return np.abs(x)
        if name == "gauss":
            return np.exp(np.float32(-5.0) * (x * x))
        if name == "identity":
            return x
        if name == "sin":
            return np.sin(x)
        if name == "relu":
            return np.maximum(x, np.float32(0))
    raise ValueError(name)


def flatten_genome(genome, config, n_leaves=2):
    """-> dict(act, bias, resp, edge_off, edge_src, edge_w, out_node) for ONE genome (numpy arrays).

    Leaves are numbered in ``config.genome_config.input_keys`` order; ``edge_src = -(i+1)`` is leaf i and
    ``-(n_leaves+1)`` the constant 1.0."""
    act, bias, resp, edge_off, edge_src, edge_w, out_node = _flatten_lists(genome, config, n_leaves)
    return dict(act=np.asarray(act, np.uint8), bias=np.asarray(bias, np.float64), resp=np.asarray(resp, np.float64),
                edge_off=np.asarray(edge_off, np.int32), edge_src=np.asarray(edge_src, np.int32),
                edge_w=np.asarray(edge_w, np.float64), out_node=np.asarray(out_node, np.int32))


def _flatten_lists(genome, config, n_leaves=2):
    """flatten_genome's work on plain Python lists (GenomeBatch concatenates these and converts once per batch)."""
    gc = config.genome_config
    in_keys, out_keys = list(gc.input_keys), list(gc.output_keys)
    if len(in_keys) != n_leaves:
        raise ValueError("PyTorch-NEAT asserts len(leaf_names) == len(input_keys): %d leaf planes, %d input keys"
                         % (n_leaves, len(in_keys)))
    leaf_of = {k: i for i, k in enumerate(in_keys)}
    required = _required_for_output(in_keys, out_keys, genome.connections)
    node_inputs = {o: [] for o in out_keys}
    out_set = set(out_keys)
    for cg in genome.connections.values():
        if not cg.enabled:
            continue
        i, o = cg.key
        if (o not in required and i not in required) or i in out_set:
            continue
        lst = node_inputs.get(o)
        if lst is None:
            lst = node_inputs[o] = []
        lst.append((i, cg.weight))
        if i not in node_inputs:
            node_inputs[i] = []

    # topological order by depth-first post-order from the outputs (feed_forward = True: acyclic)
    order, state = [], {}
    for root in out_keys:
        stack = [(root, 0)]
        while stack:
            n, ci = stack.pop()
            if n in leaf_of or state.get(n) == 2:
                continue
            conns = node_inputs[n]
            if ci == 0:
                if state.get(n) == 1:
                    raise ValueError("genome %r has a cycle through node %r" % (getattr(genome, "key", None), n))
                state[n] = 1
            if ci < len(conns):
                stack.append((n, ci + 1))
                child = conns[ci][0]
                if child not in leaf_of and state.get(child) != 2:
                    if state.get(child) == 1:
                        raise ValueError("genome %r has a cycle through node %r" % (getattr(genome, "key", None), child))
                    stack.append((child, 0))
            else:
                state[n] = 2
                order.append(n)

    const32 = {}   # node -> np.float32 constant
    index = {}
    act, bias, resp, edge_off, edge_src, edge_w = [], [], [], [0], [], []
    ONE = -(n_leaves + 1)
    for n in order:
        node = genome.nodes[n]
        conns = node_inputs[n]
        if node.aggregation != "sum" and conns:
            raise ValueError("aggregation %r unsupported (neat_configs/*.txt:17 configure 'sum' only)" % node.aggregation)
        if node.activation not in ACT_IDS:
            raise ValueError("activation %r is not in PyTorch-NEAT's str_to_activation" % node.activation)
        if not conns:
            const32[n] = np.float32(node.bias)
            continue
        if const32 and all(i in const32 for i, _ in conns):  # whole sub-graph is float32 constants: fold it
            with np.errstate(all="ignore"):
                pre = None
                for i, w in conns:
                    t = np.float32(w) * const32[i]
                    pre = t if pre is None else np.float32(pre + t)
                z = np.float32(np.float32(node.response) * pre) + np.float32(node.bias)
                const32[n] = np.float32(_f32_act(node.activation, np.float32(z)))
            continue
        index[n] = len(act)
        act.append(ACT_IDS[node.activation]); bias.append(float(node.bias)); resp.append(float(node.response))
        # Python's sum() adds left to right: a LEADING run of float32 constants is accumulated in float32 before the
        # first float64 term promotes the running sum; later float32 terms are promoted one by one.
        lead = 0
        if const32:
            while lead < len(conns) and conns[lead][0] in const32:
                lead += 1
        if lead:
            with np.errstate(all="ignore"):
                pre = None
                for i, w in conns[:lead]:
                    t = np.float32(w) * const32[i]
                    pre = t if pre is None else np.float32(pre + t)
            edge_src.append(ONE); edge_w.append(float(pre))
        for i, w in (conns[lead:] if lead else conns):
            li = leaf_of.get(i)
            if li is not None:
                edge_src.append(-(li + 1)); edge_w.append(float(w))
            elif i in const32:
                with np.errstate(all="ignore"):
                    edge_src.append(ONE); edge_w.append(float(np.float32(w) * const32[i]))
            else:
                edge_src.append(index[i]); edge_w.append(float(w))
        edge_off.append(len(edge_src))
    out_node = []
    for o in out_keys:
        if o in const32:  # constant output plane: identity(1 * (k * 1.0) + 0) == k
            index[o] = len(act)
            act.append(ACT_IDS["identity"]); bias.append(0.0); resp.append(1.0)
            edge_src.append(ONE); edge_w.append(float(const32[o])); edge_off.append(len(edge_src))
        out_node.append(index[o])
    return act, bias, resp, edge_off, edge_src, edge_w, out_node


def _marshal_python(genomes):
    """The genome objects as plain arrays (input of eigen_flatten_genomes) -- the specification of csrc/genome_walk.c."""
    conn_off, node_off = [0], [0]
    cin, cout, cw, cen, nkey, nact, nagg, nbias, nresp = [], [], [], [], [], [], [], [], []
    act_ids = ACT_IDS
    for g in genomes:
        cv = list(g.connections.values())
        keys = [c.key for c in cv]
        cin += [k[0] for k in keys]; cout += [k[1] for k in keys]
        cw += [c.weight for c in cv]; cen += [c.enabled for c in cv]
        conn_off.append(len(cin))
        nv = list(g.nodes.values())
        nkey += list(g.nodes.keys())
        nact += [act_ids.get(n.activation, 255) for n in nv]
        nagg += [n.aggregation == "sum" for n in nv]
        nbias += [n.bias for n in nv]; nresp += [n.response for n in nv]
        node_off.append(len(nkey))
    a_i32 = lambda x: np.asarray(x, np.int32)
    return (a_i32(conn_off), a_i32(cin), a_i32(cout), np.asarray(cw, np.float64), np.asarray(cen, np.uint8), a_i32(node_off),
            a_i32(nkey), np.asarray(nact, np.uint8), np.asarray(nagg, np.uint8), np.asarray(nbias, np.float64), np.asarray(nresp, np.float64))


_walker = [None]


def _marshal(genomes):
    """_marshal_python's arrays through the C walker (csrc/genome_walk.c, built by __graft_entry__.build()) when it is there:
    one call instead of ~28 K Python-level attribute reads per 256 genomes."""
    if _walker[0] is None:
        try:
            from . import _genome_walk
            _walker[0] = _genome_walk.walk
        except ImportError:
            _walker[0] = False
    if not _walker[0]:
        return _marshal_python(genomes)
    G = len(genomes)
    nc = sum(len(g.connections) for g in genomes)
    nn = sum(len(g.nodes) for g in genomes)
    out = (np.empty(G + 1, np.int32), np.empty(nc, np.int32), np.empty(nc, np.int32), np.empty(nc, np.float64), np.empty(nc, np.uint8),
           np.empty(G + 1, np.int32), np.empty(nn, np.int32), np.empty(nn, np.uint8), np.empty(nn, np.uint8), np.empty(nn, np.float64),
           np.empty(nn, np.float64))
    _walker[0](genomes, ACT_IDS, *out)
    return out


class GenomeBatch:
    """Concatenation of flattened genomes = the arrays behind ``eigen_genome_batch``."""

    def __init__(self, genomes, config, c_out, n_leaves=2, native=None):
        """native: None = use libeigen_hip.so's eigen_flatten_genomes when the library is built (the graph work of 256 genomes
        in ~1 ms instead of ~15 ms of Python), False = the Python specification below, True = require the library."""
        self.n_genomes, self.c_out = len(genomes), c_out
        if native is not False and genomes and self._init_native(genomes, config, c_out, n_leaves, required=bool(native)):
            return
        self._init_python(genomes, config, c_out, n_leaves)

    _FIELDS = (("node_off", np.int32), ("edge_off", np.int32), ("node_act", np.uint8), ("node_bias", np.float64),
               ("node_resp", np.float64), ("edge_src", np.int32), ("edge_w", np.float64), ("out_node", np.int32))

    @classmethod
    def _from_arrays(cls, n_genomes, c_out, arrays):
        gb = cls.__new__(cls)
        gb.n_genomes, gb.c_out = int(n_genomes), int(c_out)
        for (name, dt), a in zip(cls._FIELDS, arrays):
            setattr(gb, name, np.ascontiguousarray(a, dtype=dt))
        return gb

    def slice(self, lo, hi):
        """Genomes [lo, hi) as a batch of their own (offsets rebased): what one rank evaluates of a population that was
        flattened once (fitness.population_fitness, source "rank0")."""
        lo, hi = max(0, int(lo)), min(self.n_genomes, int(hi))
        if lo == 0 and hi == self.n_genomes:
            return self
        n0, n1 = int(self.node_off[lo]), int(self.node_off[hi])
        e0, e1 = int(self.edge_off[n0]), int(self.edge_off[n1])
        return self._from_arrays(hi - lo, self.c_out, (
            self.node_off[lo:hi + 1] - n0, self.edge_off[n0:n1 + 1] - e0, self.node_act[n0:n1], self.node_bias[n0:n1],
            self.node_resp[n0:n1], self.edge_src[e0:e1], self.edge_w[e0:e1], self.out_node[lo * self.c_out:hi * self.c_out]))

    def to_bytes(self):
        """One flat buffer (int64 header + the eight arrays, each padded to 8 bytes) for the rank-0 -> all broadcast."""
        arrs = [np.ascontiguousarray(getattr(self, n), dtype=dt) for n, dt in self._FIELDS]
        head = np.asarray([self.n_genomes, self.c_out] + [a.size for a in arrs], np.int64)
        parts = [head.tobytes()]
        for a in arrs:
            b = a.tobytes()
            parts.append(b + b"\0" * (-len(b) % 8))
        return b"".join(parts)

    @classmethod
    def from_bytes(cls, buf):
        buf = memoryview(buf).cast("B")
        nf = len(cls._FIELDS)
        head = np.frombuffer(buf[:8 * (2 + nf)], np.int64)
        off, arrs = 8 * (2 + nf), []
        for (name, dt), n in zip(cls._FIELDS, head[2:]):
            nb = int(n) * np.dtype(dt).itemsize
            arrs.append(np.frombuffer(buf[off:off + nb], dt).copy())
            off += nb + (-nb % 8)
        return cls._from_arrays(head[0], head[1], arrs)

    def digest(self):
        import zlib
        return zlib.crc32(self.to_bytes())

    def _init_python(self, genomes, config, c_out, n_leaves):
        node_off, act, bias, resp, edge_off, edge_src, edge_w, out_node = [0], [], [], [], [0], [], [], []
        for g in genomes:
            a, b, r, eo, es, ew, on = _flatten_lists(g, config, n_leaves)
            if len(on) < c_out:
                raise ValueError("genome %r has %d outputs, %d are rendered" % (getattr(g, "key", None), len(on), c_out))
            base = len(edge_src)
            act += a; bias += b; resp += r; edge_src += es; edge_w += ew
            edge_off += [base + o for o in eo[1:]]
            out_node += on[:c_out]
            node_off.append(len(act))
        self.node_off = np.asarray(node_off, np.int32)
        self.edge_off = np.asarray(edge_off, np.int32)
        self.node_act = np.asarray(act, np.uint8)
        self.node_bias, self.node_resp = np.asarray(bias, np.float64), np.asarray(resp, np.float64)
        self.edge_src, self.edge_w = np.asarray(edge_src, np.int32), np.asarray(edge_w, np.float64)
        self.out_node = np.asarray(out_node, np.int32)

    def _init_native(self, genomes, config, c_out, n_leaves, required=False):
        """Marshal the genomes into plain arrays and let the library do the graph work; genomes it declines (a node whose inputs
        are all constants needs numpy's float32 activations; invalid genomes need their exception) go through _flatten_lists."""
        import ctypes
        try:
            from .engine import load_library
            lib = load_library()
        except Exception:
            if required:
                raise
            return False
        gc = config.genome_config
        in_keys, out_keys = list(gc.input_keys), list(gc.output_keys)
        if len(in_keys) != n_leaves:
            raise ValueError("PyTorch-NEAT asserts len(leaf_names) == len(input_keys): %d leaf planes, %d input keys"
                             % (n_leaves, len(in_keys)))
        if len(out_keys) < c_out:
            raise ValueError("genome %r has %d outputs, %d are rendered" % (getattr(genomes[0], "key", None), len(out_keys), c_out))
        G, n_out = len(genomes), len(out_keys)
        conn_off, cin, cout, cw, cen, node_off, nkey, nact, nagg, nbias, nresp = _marshal(genomes)
        a_i32 = lambda x: np.asarray(x, np.int32)
        ik, ok = a_i32(in_keys), a_i32(out_keys)
        cap_nodes, cap_edges = len(nkey) + G * n_out + 1, len(cin) + len(nkey) + G * n_out + 1
        o_node_off, o_edge_off = np.zeros(G + 1, np.int32), np.zeros(cap_nodes + 1, np.int32)
        o_act, o_bias, o_resp = np.zeros(cap_nodes, np.uint8), np.zeros(cap_nodes, np.float64), np.zeros(cap_nodes, np.float64)
        o_src, o_w = np.zeros(cap_edges, np.int32), np.zeros(cap_edges, np.float64)
        o_out, o_status = np.zeros((G, n_out), np.int32), np.zeros(G, np.uint8)
        ptr = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        rc = lib.eigen_flatten_genomes(ctypes.c_int32(G), ctypes.c_int32(len(in_keys)), ptr(ik), ctypes.c_int32(n_out), ptr(ok),
                                       ptr(conn_off), ptr(cin), ptr(cout), ptr(cw), ptr(cen), ptr(node_off), ptr(nkey), ptr(nact), ptr(nagg),
                                       ptr(nbias), ptr(nresp), ctypes.c_int32(cap_nodes), ctypes.c_int32(cap_edges), ptr(o_node_off),
                                       ptr(o_edge_off), ptr(o_act), ptr(o_bias), ptr(o_resp), ptr(o_src), ptr(o_w), ptr(o_out), ptr(o_status))
        if rc != 0:
            raise RuntimeError("eigen_flatten_genomes failed: %s" % lib.eigen_last_error().decode())
        nn = int(o_node_off[G])
        ne = int(o_edge_off[nn])
        if not o_status.any():
            self.node_off, self.edge_off = o_node_off, o_edge_off[:nn + 1].copy()
            self.node_act, self.node_bias, self.node_resp = o_act[:nn].copy(), o_bias[:nn].copy(), o_resp[:nn].copy()
            self.edge_src, self.edge_w = o_src[:ne].copy(), o_w[:ne].copy()
            self.out_node = np.ascontiguousarray(o_out[:, :c_out]).reshape(-1)
            return True
        # splice: declined genomes come from the Python specification (which also raises for invalid ones); runs of
        # consecutive genomes the library did flatten are copied as one block each
        act, bias, resp, esrc, ew, eoff, outn, noff = [], [], [], [], [], [np.zeros(1, np.int32)], [], [np.zeros(1, np.int32)]
        n_tot = e_tot = 0
        gi = 0
        bad = np.flatnonzero(o_status).tolist() + [G]
        for nxt in bad:
            if nxt > gi:  # block of genomes gi..nxt-1 straight from the library's arrays
                n0, n1 = int(o_node_off[gi]), int(o_node_off[nxt])
                e0, e1 = int(o_edge_off[n0]), int(o_edge_off[n1])
                act.append(o_act[n0:n1]); bias.append(o_bias[n0:n1]); resp.append(o_resp[n0:n1])
                esrc.append(o_src[e0:e1]); ew.append(o_w[e0:e1])
                eoff.append(o_edge_off[n0 + 1:n1 + 1] - e0 + e_tot)
                noff.append(o_node_off[gi + 1:nxt + 1] - n0 + n_tot)
                outn.append(o_out[gi:nxt, :c_out].reshape(-1))
                n_tot += n1 - n0; e_tot += e1 - e0
            if nxt < G:
                a_, b_, r_, eo, es, w_, on = _flatten_lists(genomes[nxt], config, n_leaves)
                if len(on) < c_out:
                    raise ValueError("genome %r has %d outputs, %d are rendered" % (getattr(genomes[nxt], "key", None), len(on), c_out))
                act.append(np.asarray(a_, np.uint8)); bias.append(np.asarray(b_, np.float64)); resp.append(np.asarray(r_, np.float64))
                esrc.append(np.asarray(es, np.int32)); ew.append(np.asarray(w_, np.float64))
                eoff.append(np.asarray(eo[1:], np.int32) + e_tot)
                outn.append(np.asarray(on[:c_out], np.int32))
                n_tot += len(a_); e_tot += len(es)
                noff.append(np.asarray([n_tot], np.int32))
            gi = nxt + 1
        cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs).astype(dt, copy=False)) if xs else np.zeros(0, dt)
        self.node_off, self.edge_off = cat(noff, np.int32), cat(eoff, np.int32)
        self.node_act, self.node_bias, self.node_resp = cat(act, np.uint8), cat(bias, np.float64), cat(resp, np.float64)
        self.edge_src, self.edge_w = cat(esrc, np.int32), cat(ew, np.float64)
        self.out_node = cat(outn, np.int32)
        return True
