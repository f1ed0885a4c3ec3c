"""Seeded synthetic NEAT genomes shaped like neat-python's DefaultGenome (duck-typed).

neat-python is not installed in the build image or on the GPU box, so benchmarks and tests
use these objects.  They expose exactly the attributes the fitness path reads:
``genome.nodes[k].{bias,response,activation,aggregation}``, ``genome.connections[(i,o)].{key,weight,enabled}``,
``genome.fitness`` and ``config.genome_config.{input_keys,output_keys}``.

The generator follows the ``[DefaultGenome]`` section of /root/reference/neat_configs/circles.txt:10-68:
``num_inputs=2, num_hidden=20, initial_connection = partial_nodirect 0.8`` (inputs->hidden and
hidden->outputs, 80 % of them kept), weights ~ N(0.1, 1) and biases ~ N(0, 1) clipped to +-30,
response = 1, aggregation = sum.  To emulate a population after some generations of mutation
(activation_mutate_rate 0.5, conn_add_prob 0.5, enabled_mutate_rate 0.1) activations are drawn uniformly
from ``activation_options`` and a few forward hidden->hidden links and disabled links are added.
"""
import random
from types import SimpleNamespace

ACTIVATION_OPTIONS = ("sin", "sigmoid", "gauss", "tanh", "relu", "abs")  # neat_configs/circles.txt:12


class NodeGene(SimpleNamespace):
    pass


class ConnectionGene(SimpleNamespace):
    pass


class Genome:
    def __init__(self, key):
        self.key = key
        self.nodes = {}
        self.connections = {}
        self.fitness = None

    def size(self):
        return len(self.nodes), sum(1 for c in self.connections.values() if c.enabled)


def make_config(num_inputs=2, num_outputs=3):
    gc = SimpleNamespace(input_keys=[-i - 1 for i in range(num_inputs)], output_keys=list(range(num_outputs)),
                         num_inputs=num_inputs, num_outputs=num_outputs)
    return SimpleNamespace(genome_config=gc)


def _clip(v, lo=-30.0, hi=30.0):
    return max(lo, min(hi, v))


def make_genome(key, config, seed, num_hidden=20, fraction=0.8, extra_links=6, disabled_fraction=0.05,
                activations=ACTIVATION_OPTIONS, weight_mean=0.1, weight_std=1.0, bias_std=1.0):
    rng = random.Random(seed)
    gc = config.genome_config
    g = Genome(key)
    hidden = [len(gc.output_keys) + i for i in range(num_hidden)]
    for k in list(gc.output_keys) + hidden:
        g.nodes[k] = NodeGene(key=k, bias=_clip(rng.gauss(0.0, bias_std)), response=1.0,
                              activation=rng.choice(activations), aggregation="sum")
    links = []
    if hidden:
        links += [(i, h) for i in gc.input_keys for h in hidden]
        links += [(h, o) for h in hidden for o in gc.output_keys]
    else:
        links += [(i, o) for i in gc.input_keys for o in gc.output_keys]
    rng.shuffle(links)
    links = links[:int(round(len(links) * fraction))]
    for _ in range(extra_links):  # forward-only hidden->hidden links keep the graph feed-forward
        if len(hidden) >= 2:
            a, b = sorted(rng.sample(hidden, 2))
            if (a, b) not in links:
                links.append((a, b))
    for key_ in links:
        g.connections[key_] = ConnectionGene(key=key_, weight=_clip(rng.gauss(weight_mean, weight_std)),
                                             enabled=rng.random() >= disabled_fraction)
    return g


def make_population(pop_size, config, seed=0, **kw):
    """[(genome_id, genome), ...] as neat.Population hands it to eval_genomes (generate_illusion.py:692-694)."""
    return [(i + 1, make_genome(i + 1, config, seed * 1000003 + i, **kw)) for i in range(pop_size)]
