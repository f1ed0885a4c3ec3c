"""ctypes binding of the C-ABI engine library (include/eigen_engine.h -> libeigen_hip.so).

The library is the ONLY compute path of this package: if it is missing or cannot be loaded the import of
:class:`Engine` users fails loudly -- there is no PyTorch/CPU fallback.  torch is used by callers for device
memory and streams only; this module passes raw pointers.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EIGEN_HIP_LIB") or os.path.join(_HERE, "libeigen_hip.so")  # EIGEN_HIP_LIB: A/B builds (scripts/)
ABI_VERSION = 4  # include/eigen_engine.h: EIGEN_ABI_VERSION
MAX_LAYERS = 8

PAIR_POPULATION, PAIR_SINGLE = 0, 1

EXPORTS = ["eigen_abi_version", "eigen_gate_order", "eigen_winograd_mask", "eigen_last_error", "eigen_config_defaults", "eigen_create", "eigen_destroy",
           "eigen_set_prednet_weights", "eigen_set_grid", "eigen_render_cppn", "eigen_eval_cppn_nodes", "eigen_prednet_rollout", "eigen_flow",
           "eigen_score", "eigen_eval_population", "eigen_eval_images", "eigen_test_conv", "eigen_time_conv", "eigen_test_det_math",
           "eigen_get_timings", "eigen_conv_profile", "eigen_debug_corners", "eigen_debug_dense_flow", "eigen_prednet_flops_per_step", "eigen_flatten_genomes"]


class EigenConfig(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("n_layers", ctypes.c_int32), ("channels", ctypes.c_int32 * MAX_LAYERS), ("max_batch", ctypes.c_int32),
                ("n_repeat", ctypes.c_int32), ("n_ext", ctypes.c_int32), ("requant_feedback", ctypes.c_int32),
                ("lk_max_corners", ctypes.c_int32), ("lk_block_size", ctypes.c_int32), ("lk_win", ctypes.c_int32),
                ("lk_max_level", ctypes.c_int32), ("lk_max_iter", ctypes.c_int32), ("flow_method", ctypes.c_int32),
                ("lk_quality_level", ctypes.c_double), ("lk_min_distance", ctypes.c_double),
                ("lk_epsilon", ctypes.c_double), ("lk_min_eig_thr", ctypes.c_double),
                ("fb_levels", ctypes.c_int32), ("fb_winsize", ctypes.c_int32), ("fb_iterations", ctypes.c_int32),
                ("fb_poly_n", ctypes.c_int32), ("fb_step", ctypes.c_int32), ("reserved1", ctypes.c_int32),
                ("fb_poly_sigma", ctypes.c_double)]


class GenomeBatchC(ctypes.Structure):
    _fields_ = [("n_genomes", ctypes.c_int32), ("c_out", ctypes.c_int32),
                ("node_off", ctypes.c_void_p), ("edge_off", ctypes.c_void_p), ("node_act", ctypes.c_void_p),
                ("node_bias", ctypes.c_void_p), ("node_resp", ctypes.c_void_p), ("edge_src", ctypes.c_void_p),
                ("edge_w", ctypes.c_void_p), ("out_node", ctypes.c_void_p)]


FLOW_METHODS = {"lk": 0, "farneback": 1}  # eigen_flow_method


class EngineError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen libeigen_hip.so (built by ``__graft_entry__.build()``); raises if it is absent."""
    global _lib
    if _lib is None or path is not None:
        p = path or LIB_PATH
        # PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime of the process (device tensors and
        # streams come from torch), so make sure it is loaded before this library's dependency is resolved.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(p):
            raise EngineError("HIP engine library %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % p)
        lib = ctypes.CDLL(p)
        lib.eigen_last_error.restype = ctypes.c_char_p
        lib.eigen_prednet_flops_per_step.restype = ctypes.c_double
        lib.eigen_prednet_flops_per_step.argtypes = [ctypes.c_void_p]
        for name in EXPORTS:
            getattr(lib, name)  # AttributeError if the ABI drifted
        if lib.eigen_abi_version() != ABI_VERSION:  # EigenConfig below mirrors eigen_config of exactly this version
            raise EngineError("%s has ABI version %d, this package binds version %d: rebuild it" % (p, lib.eigen_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise EngineError("eigen engine error %d: %s" % (rc, load_library().eigen_last_error().decode()))


def _ptr(x):
    """Raw address of a numpy array / torch tensor / int."""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if isinstance(x, np.ndarray):
        return ctypes.c_void_p(x.ctypes.data)
    return ctypes.c_void_p(x.data_ptr())  # torch tensor


def _stream_arg(stream):
    if stream is None:
        return None
    return ctypes.c_void_p(int(getattr(stream, "cuda_stream", stream)))


class Engine:
    """One engine handle = one GPU rank.  Sizes are fixed at creation (workspaces live in HBM)."""

    def __init__(self, width, height, channels, max_batch, device=0, n_repeat=20, n_ext=2, requant_feedback=False, flow="lk", **lk):
        self.lib = load_library()
        cfg = EigenConfig()
        self.lib.eigen_config_defaults(ctypes.byref(cfg))
        cfg.device, cfg.width, cfg.height, cfg.n_layers, cfg.max_batch = device, width, height, len(channels), max_batch
        for i, c in enumerate(channels):
            cfg.channels[i] = c
        cfg.n_repeat, cfg.n_ext, cfg.requant_feedback = n_repeat, n_ext, int(requant_feedback)
        if flow not in FLOW_METHODS:
            raise ValueError("flow must be one of %s" % sorted(FLOW_METHODS))
        cfg.flow_method = FLOW_METHODS[flow]
        for k, v in lk.items():  # lk_* (Lucas-Kanade) or fb_* (Farneback) parameters, prefix optional for lk
            name = k if k.startswith("fb_") else "lk_" + k
            if not hasattr(cfg, name):
                raise TypeError("unknown flow parameter %r" % k)
            setattr(cfg, name, v)
        self.cfg = cfg
        self.width, self.height, self.channels, self.max_batch = width, height, list(channels), max_batch
        self.c_dim, self.K = channels[0], cfg.lk_max_corners
        self._h = ctypes.c_void_p()
        _check(self.lib.eigen_create(ctypes.byref(cfg), ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.eigen_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------------------------------
    def set_weights(self, weights):
        from .weights import tensor_names, tensor_shapes
        names = tensor_names(len(self.channels))
        shapes = tensor_shapes(self.channels, self.width, self.height)
        arrs = []
        for n in names:
            a = np.ascontiguousarray(weights[n], dtype=np.float32)
            if a.shape != shapes[n]:
                raise ValueError("tensor %r has shape %s, expected %s" % (n, a.shape, shapes[n]))
            arrs.append(a)
        tab = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        _check(self.lib.eigen_set_prednet_weights(self._h, tab, ctypes.c_int32(len(arrs))))

    def set_grid(self, planes):
        planes = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.float64).reshape(-1) for p in planes]))
        if planes.shape[1] != self.width * self.height:
            raise ValueError("grid planes must have H*W = %d values" % (self.width * self.height))
        self.n_planes = planes.shape[0]
        _check(self.lib.eigen_set_grid(self._h, _ptr(planes), ctypes.c_int32(planes.shape[0])))

    @staticmethod
    def _genome_struct(gb):
        s = GenomeBatchC(gb.n_genomes, gb.c_out, gb.node_off.ctypes.data, gb.edge_off.ctypes.data, gb.node_act.ctypes.data,
                         gb.node_bias.ctypes.data, gb.node_resp.ctypes.data, gb.edge_src.ctypes.data, gb.edge_w.ctypes.data,
                         gb.out_node.ctypes.data)
        return s

    # -- stages --------------------------------------------------------------------------------------
    def render_cppn(self, gb, d_images, bg=1, gradient=1, stream=None):
        s = self._genome_struct(gb)
        _check(self.lib.eigen_render_cppn(self._h, ctypes.byref(s), ctypes.c_int32(bg), ctypes.c_int32(gradient), _ptr(d_images), _stream_arg(stream)))

    def eval_cppn_nodes(self, gb, d_nodes, stream=None):
        """float64 [n, c_out, H*W] raw output-node values (create_cppn node calls, generate_illusion.py:395)."""
        s = self._genome_struct(gb)
        _check(self.lib.eigen_eval_cppn_nodes(self._h, ctypes.byref(s), _ptr(d_nodes), _stream_arg(stream)))

    def _check_images(self, d_images, batch):
        """The C side reads batch * C0 * H * W bytes from the raw pointer: refuse anything smaller."""
        if not 1 <= int(batch) <= self.max_batch:
            return  # the library reports capacity errors itself (EIGEN_ERR_CAPACITY)
        need = int(batch) * self.c_dim * self.height * self.width
        have = getattr(d_images, "numel", None)
        have = have() if callable(have) else getattr(d_images, "size", None)
        if isinstance(have, int) and have < need:
            raise ValueError("image buffer holds %d bytes, %d images of (%d, %d, %d) need %d"
                             % (have, batch, self.c_dim, self.height, self.width, need))

    def prednet_rollout(self, d_images, batch, n_steps, first_out_step, d_frames, stream=None):
        self._check_images(d_images, batch)
        _check(self.lib.eigen_prednet_rollout(self._h, _ptr(d_images), ctypes.c_int32(batch), ctypes.c_int32(n_steps),
                                              ctypes.c_int32(first_out_step), _ptr(d_frames), _stream_arg(stream)))

    def flow(self, d_img0, stride0, d_img1, stride1, batch, d_vectors, d_counts, stream=None):
        _check(self.lib.eigen_flow(self._h, _ptr(d_img0), ctypes.c_int64(stride0), _ptr(d_img1), ctypes.c_int64(stride1),
                                   ctypes.c_int32(batch), _ptr(d_vectors), _ptr(d_counts), _stream_arg(stream)))

    def score(self, structure, d_vectors, d_counts, batch, d_fitness, stream=None, width=0, height=0):
        _check(self.lib.eigen_score(self._h, ctypes.c_int32(int(structure)), ctypes.c_int32(width), ctypes.c_int32(height), _ptr(d_vectors),
                                    _ptr(d_counts), ctypes.c_int32(batch), _ptr(d_fitness), _stream_arg(stream)))

    def eval_population(self, gb, structure, bg=1, gradient=1, pairing=PAIR_POPULATION, stream=None):
        fit = np.zeros(gb.n_genomes, dtype=np.float64)
        s = self._genome_struct(gb)
        _check(self.lib.eigen_eval_population(self._h, ctypes.byref(s), ctypes.c_int32(int(structure)), ctypes.c_int32(bg),
                                              ctypes.c_int32(gradient), ctypes.c_int32(pairing), _ptr(fit), _stream_arg(stream)))
        return fit

    def eval_images(self, d_images, batch, structure, pairing=PAIR_SINGLE, stream=None):
        self._check_images(d_images, batch)
        fit = np.zeros(batch, dtype=np.float64)
        vec = np.zeros((batch, self.K, 4), dtype=np.float32)
        cnt = np.zeros(batch, dtype=np.int32)
        _check(self.lib.eigen_eval_images(self._h, _ptr(d_images), ctypes.c_int32(batch), ctypes.c_int32(int(structure)),
                                          ctypes.c_int32(pairing), _ptr(fit), _ptr(vec), _ptr(cnt), _stream_arg(stream)))
        return fit, [vec[i, :cnt[i]].copy() for i in range(batch)]

    # -- test / measurement hooks --------------------------------------------------------------------
    def test_conv(self, d_srcs, cins, ups, h_weights, cout, H, W, batch, d_out, stream=None):
        n = len(d_srcs)
        st = (ctypes.c_void_p * n)(*[int(s.data_ptr()) for s in d_srcs])
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w in h_weights]
        wt = (ctypes.c_void_p * n)(*[w.ctypes.data for w in ws])
        ci = (ctypes.c_int32 * n)(*cins)
        up = (ctypes.c_int32 * n)(*ups)
        _check(self.lib.eigen_test_conv(self._h, ctypes.c_int32(n), st, ci, up, wt, ctypes.c_int32(cout), ctypes.c_int32(H),
                                        ctypes.c_int32(W), ctypes.c_int32(batch), _ptr(d_out), _stream_arg(stream)))

    def time_conv(self, d_srcs, cins, ups, h_weights, cout, H, W, batch, d_out, iters=5, stream=None):
        n = len(d_srcs)
        st = (ctypes.c_void_p * n)(*[int(s.data_ptr()) for s in d_srcs])
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w in h_weights]
        wt = (ctypes.c_void_p * n)(*[w.ctypes.data for w in ws])
        ci = (ctypes.c_int32 * n)(*cins)
        up = (ctypes.c_int32 * n)(*ups)
        ms = ctypes.c_double(0)
        _check(self.lib.eigen_time_conv(self._h, ctypes.c_int32(n), st, ci, up, wt, ctypes.c_int32(cout), ctypes.c_int32(H),
                                        ctypes.c_int32(W), ctypes.c_int32(batch), _ptr(d_out), ctypes.c_int32(iters), ctypes.byref(ms),
                                        _stream_arg(stream)))
        return ms.value

    def test_det_math(self, d_x, n, d_exp, d_sig, d_tanh, stream=None):
        _check(self.lib.eigen_test_det_math(self._h, _ptr(d_x), ctypes.c_int32(n), _ptr(d_exp), _ptr(d_sig), _ptr(d_tanh), _stream_arg(stream)))

    def timings(self):
        ms = np.zeros(6, dtype=np.float64)
        _check(self.lib.eigen_get_timings(self._h, _ptr(ms)))
        return dict(render_ms=ms[0], prednet_ms=ms[1], flow_ms=ms[2], score_ms=ms[3], conv_ms=ms[4], conv_launches=int(ms[5]))

    def conv_profile(self, enable, reset=True):
        out = np.zeros((6 * MAX_LAYERS, 8), dtype=np.float64)
        n = ctypes.c_int32(0)
        _check(self.lib.eigen_conv_profile(self._h, ctypes.c_int32(int(enable)), ctypes.c_int32(int(reset)), _ptr(out),
                                           ctypes.c_int32(out.shape[0]), ctypes.byref(n)))
        rows = []
        for r in out[:n.value]:
            rows.append(dict(layer=int(r[0]), epi={1: "lstm", 2: "convA", 3: "convP", 4: "lstm", 5: "up4", 6: "up4"}.get(int(r[1]) & 15, "raw"), step0=bool((int(r[1]) >> 4) & 1),
                             wino=bool(int(r[1]) & 32),  # Winograd form (csrc/conv_wino4.h): flops_per_image counts ITS multiply-adds
                             NI=int(r[2]), TW=int(r[3]),
                             launches=int(r[4]), ms=float(r[5]), flops_per_image=float(r[6]), n_nblk=int(r[7])))
        return rows

    def debug_dense_flow(self, batch, stream=None):
        """float32 [batch, 2, H, W]: the dense Farneback field of the last flow call (engines created with flow="farneback")."""
        f = np.zeros((batch, 2, self.height, self.width), np.float32)
        _check(self.lib.eigen_debug_dense_flow(self._h, ctypes.c_int32(batch), _ptr(f), _stream_arg(stream)))
        return f

    def debug_corners(self, batch, stream=None):
        c = np.zeros((batch, self.K, 2), np.float32); n = np.zeros(batch, np.int32)
        nx = np.zeros((batch, self.K, 2), np.float32); st = np.zeros((batch, self.K), np.uint8)
        _check(self.lib.eigen_debug_corners(self._h, ctypes.c_int32(batch), _ptr(c), _ptr(n), _ptr(nx), _ptr(st), _stream_arg(stream)))
        return c, n, nx, st

    def flops_per_step(self):
        return float(self.lib.eigen_prednet_flops_per_step(self._h))
