"""`from pytorch_neat.pytorch_neat.cppn import create_cppn` (generate_illusion.py:14).

create_cppn(genome, config, leaf_names, node_names) returns one callable per output key of the config; the reference
calls each as ``node(x=inp_x, y=inp_y)`` with flat float64 torch tensors and converts the result with ``.numpy()``
(generate_illusion.py:384-397, 436-445).  Here a call evaluates the genome's pruned graph for that output on the
device (float64, same connection order and activations as PyTorch-NEAT) and returns a CPU float64 tensor.  All
outputs of one genome over the same inputs are evaluated in one launch and cached, so the colour loop over the three
nodes costs one device pass.
"""
import numpy as np


class _Shared:
    def __init__(self, genome, config, leaf_names):
        self.genome, self.config, self.leaf_names = genome, config, list(leaf_names)
        self.key, self.values = None, None

    def evaluate(self, inputs):
        import torch
        from evolutionary_illusion_generator_amd import fitness
        missing = [n for n in self.leaf_names if n not in inputs]
        if missing:
            raise KeyError("create_cppn node call: missing inputs %s" % missing)
        arrs = [np.asarray(inputs[n].detach().cpu().numpy() if isinstance(inputs[n], torch.Tensor) else inputs[n], dtype=np.float64)
                for n in self.leaf_names]
        key = tuple((a.ctypes.data, a.shape, float(a.reshape(-1)[0]) if a.size else 0.0, float(a.reshape(-1)[-1]) if a.size else 0.0) for a in arrs)
        if key != self.key:
            shape = arrs[0].shape
            self.values = fitness.cppn_node_planes(self.genome, self.config, arrs).reshape((-1,) + shape)
            self.key = key
            self._keep = arrs  # keeps the buffers (and therefore the data pointers of the key) alive
        return self.values


class Node:
    """One CPPN output: ``node(x=..., y=...)`` -> float64 tensor shaped like the inputs."""

    def __init__(self, shared, index, name=None):
        self._shared, self._index, self.name = shared, index, name

    def __call__(self, **inputs):
        import torch
        return torch.from_numpy(np.array(self._shared.evaluate(inputs)[self._index]))

    def __repr__(self):
        return "Node(output %d of genome %r, on the HIP engine)" % (self._index, getattr(self._shared.genome, "key", None))


def create_cppn(genome, config, leaf_names, node_names, output_activation=None):
    if output_activation is not None:
        raise NotImplementedError("create_cppn shim: output_activation is not used by the reference and not supported")
    n_in = len(config.genome_config.input_keys)
    # PyTorch-NEAT asserts the leaf list against the config's inputs
    assert len(leaf_names) == n_in, "create_cppn: %d leaf names for a config with %d inputs" % (len(leaf_names), n_in)
    shared = _Shared(genome, config, leaf_names)
    names = list(node_names) if node_names else [None] * len(config.genome_config.output_keys)
    return [Node(shared, i, names[i] if i < len(names) else None) for i in range(len(config.genome_config.output_keys))]
