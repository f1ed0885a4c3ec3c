"""neat_lite -- a small NEAT implementation exposing the slice of neat-python's API that EIGen's driver uses.

SURVEY section 8(f) row 1: neat-python is not installed here or on the GPU box, so the reference's evolution loop
(/root/reference/generate_illusion.py:688-711) cannot run without a stand-in.  This module provides, written from
scratch and much simplified, exactly the names that loop touches::

    config = neat.Config(neat.DefaultGenome, neat.DefaultReproduction, neat.DefaultSpeciesSet, neat.DefaultStagnation, path)
    p = neat.Population(config)              # or  neat.Checkpointer(100).restore_checkpoint(path)
    p.add_reporter(neat.StdOutReporter(True)); p.add_reporter(neat.StatisticsReporter()); p.add_reporter(neat.Checkpointer(100))
    winner = p.run(eval_genomes, 100)        # eval_genomes(list_of_(id, genome), config) sets genome.fitness

and genomes with the attributes the fitness path reads (``nodes[k].bias/response/activation/aggregation``,
``connections[(i, o)].key/weight/enabled``, ``config.genome_config.input_keys/output_keys``).  It reads the same INI keys as
/root/reference/neat_configs/*.txt.  It is NOT bit-compatible with neat-python's random stream and is host-side, O(pop)
Python: outside the accelerated path and outside the parity claims.
"""
import configparser
import math
import pickle
import random
import time

__all__ = ["Config", "DefaultGenome", "DefaultReproduction", "DefaultSpeciesSet", "DefaultStagnation", "Population",
           "StdOutReporter", "StatisticsReporter", "Checkpointer"]


class _Section(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)


def _parse(v):
    s = v.strip()
    if s.lower() in ("true", "false"):
        return s.lower() == "true"
    for t in (int, float):
        try:
            return t(s)
        except ValueError:
            pass
    return s


class DefaultReproduction:
    pass


class DefaultSpeciesSet:
    pass


class DefaultStagnation:
    pass


class Config:
    def __init__(self, genome_type, reproduction_type, species_set_type, stagnation_type, filename):
        cp = configparser.ConfigParser()
        if not cp.read(filename):
            raise FileNotFoundError(filename)
        sec = lambda name: _Section({k: _parse(v) for k, v in cp[name].items()}) if cp.has_section(name) else _Section()
        n = sec("NEAT")
        self.pop_size = n["pop_size"]
        self.fitness_criterion = n.get("fitness_criterion", "max")
        self.fitness_threshold = n.get("fitness_threshold", float("inf"))
        self.no_fitness_termination = n.get("no_fitness_termination", False)
        self.reset_on_extinction = n.get("reset_on_extinction", False)
        g = sec("DefaultGenome")
        g.setdefault("activation_options", g.get("activation_default", "sigmoid"))
        g.setdefault("aggregation_options", g.get("aggregation_default", "sum"))
        g["activation_options"] = str(g["activation_options"]).split()
        g["aggregation_options"] = str(g["aggregation_options"]).split()
        ic = str(g.get("initial_connection", "unconnected")).split()
        g["initial_connection"] = ic[0]
        g["connection_fraction"] = float(ic[1]) if len(ic) > 1 else 1.0
        g["input_keys"] = [-i - 1 for i in range(g["num_inputs"])]
        g["output_keys"] = list(range(g["num_outputs"]))
        self.genome_config = g
        self.genome_type = genome_type
        self.species_set_config = sec("DefaultSpeciesSet")
        st = sec("DefaultStagnation")
        st.setdefault("species_fitness_func", "mean"); st.setdefault("max_stagnation", 15); st.setdefault("species_elitism", 0)
        self.stagnation_config = st
        r = sec("DefaultReproduction")
        r.setdefault("elitism", 0); r.setdefault("survival_threshold", 0.2); r.setdefault("min_species_size", 1)
        self.reproduction_config = r


class NodeGene:
    __slots__ = ("key", "bias", "response", "activation", "aggregation")

    def copy(self):
        n = NodeGene()
        for a in self.__slots__:
            setattr(n, a, getattr(self, a))
        return n


class ConnectionGene:
    __slots__ = ("key", "weight", "enabled")

    def copy(self):
        c = ConnectionGene()
        c.key, c.weight, c.enabled = self.key, self.weight, self.enabled
        return c


def _init_float(g, name, rng):
    v = rng.gauss(g[name + "_init_mean"], g[name + "_init_stdev"])
    return max(g[name + "_min_value"], min(g[name + "_max_value"], v))


def _mutate_float(g, name, v, rng):
    r = rng.random()
    if r < g[name + "_mutate_rate"]:
        v = v + rng.gauss(0.0, g[name + "_mutate_power"])
        return max(g[name + "_min_value"], min(g[name + "_max_value"], v))
    if r < g[name + "_mutate_rate"] + g[name + "_replace_rate"]:
        return _init_float(g, name, rng)
    return v


def _creates_cycle(connections, test):
    i, o = test
    if i == o:
        return True
    visited = {o}
    while True:
        n_added = 0
        for a, b in connections:
            if a in visited and b not in visited:
                if b == i:
                    return True
                visited.add(b)
                n_added += 1
        if n_added == 0:
            return False


class DefaultGenome:
    def __init__(self, key):
        self.key = key
        self.nodes = {}
        self.connections = {}
        self.fitness = None

    # -- construction ------------------------------------------------------------------------------
    @staticmethod
    def _new_node(g, key, rng):
        n = NodeGene()
        n.key = key
        n.bias = _init_float(g, "bias", rng)
        n.response = _init_float(g, "response", rng)
        n.activation = g.get("activation_default", g["activation_options"][0])
        if n.activation == "random":
            n.activation = rng.choice(g["activation_options"])
        n.aggregation = g.get("aggregation_default", "sum")
        return n

    @staticmethod
    def _new_conn(g, key, rng):
        c = ConnectionGene()
        c.key = key
        c.weight = _init_float(g, "weight", rng)
        c.enabled = bool(g.get("enabled_default", True))
        return c

    def configure_new(self, g, rng):
        for k in g["output_keys"]:
            self.nodes[k] = self._new_node(g, k, rng)
        hidden = []
        for _ in range(g.get("num_hidden", 0)):
            k = max(self.nodes) + 1
            self.nodes[k] = self._new_node(g, k, rng)
            hidden.append(k)
        mode = g["initial_connection"]
        direct = not mode.endswith("nodirect")
        links = []
        if mode != "unconnected":
            if hidden:
                links += [(i, h) for i in g["input_keys"] for h in hidden]
                links += [(h, o) for h in hidden for o in g["output_keys"]]
            if direct or not hidden:
                links += [(i, o) for i in g["input_keys"] for o in g["output_keys"]]
            if mode.startswith("partial"):
                rng.shuffle(links)
                links = links[:int(round(len(links) * g["connection_fraction"]))]
        for key in links:
            self.connections[key] = self._new_conn(g, key, rng)

    def configure_crossover(self, p1, p2, rng):
        if p2.fitness > p1.fitness:
            p1, p2 = p2, p1
        for key, c1 in p1.connections.items():
            c2 = p2.connections.get(key)
            self.connections[key] = (c1 if c2 is None or rng.random() < 0.5 else c2).copy()
        for key, n1 in p1.nodes.items():
            n2 = p2.nodes.get(key)
            self.nodes[key] = (n1 if n2 is None or rng.random() < 0.5 else n2).copy()

    # -- mutation ----------------------------------------------------------------------------------
    def mutate(self, g, rng):
        if rng.random() < g.get("node_add_prob", 0):
            self._mutate_add_node(g, rng)
        if rng.random() < g.get("node_delete_prob", 0):
            self._mutate_delete_node(g, rng)
        if rng.random() < g.get("conn_add_prob", 0):
            self._mutate_add_conn(g, rng)
        if rng.random() < g.get("conn_delete_prob", 0) and self.connections:
            del self.connections[rng.choice(list(self.connections))]
        for c in self.connections.values():
            c.weight = _mutate_float(g, "weight", c.weight, rng)
            if rng.random() < g.get("enabled_mutate_rate", 0):
                c.enabled = rng.random() < 0.5
        for n in self.nodes.values():
            n.bias = _mutate_float(g, "bias", n.bias, rng)
            n.response = _mutate_float(g, "response", n.response, rng)
            if rng.random() < g.get("activation_mutate_rate", 0):
                n.activation = rng.choice(g["activation_options"])
            if rng.random() < g.get("aggregation_mutate_rate", 0):
                n.aggregation = rng.choice(g["aggregation_options"])

    def _mutate_add_node(self, g, rng):
        if not self.connections:
            return
        c = rng.choice(list(self.connections.values()))
        k = max(self.nodes) + 1
        self.nodes[k] = self._new_node(g, k, rng)
        c.enabled = False
        i, o = c.key
        a = self._new_conn(g, (i, k), rng); a.weight, a.enabled = 1.0, True
        b = self._new_conn(g, (k, o), rng); b.weight, b.enabled = c.weight, True
        self.connections[a.key], self.connections[b.key] = a, b

    def _mutate_delete_node(self, g, rng):
        cand = [k for k in self.nodes if k not in g["output_keys"]]
        if not cand:
            return
        k = rng.choice(cand)
        for key in [key for key in self.connections if k in key]:
            del self.connections[key]
        del self.nodes[k]

    def _mutate_add_conn(self, g, rng):
        outs = list(self.nodes)
        ins = outs + g["input_keys"]
        o, i = rng.choice(outs), rng.choice(ins)
        key = (i, o)
        if key in self.connections:
            self.connections[key].enabled = True
            return
        if i in g["output_keys"] and o in g["output_keys"]:
            return
        if g.get("feed_forward", True) and _creates_cycle(list(self.connections), key):
            return
        self.connections[key] = self._new_conn(g, key, rng)

    # -- distance ----------------------------------------------------------------------------------
    def distance(self, other, g):
        cd, cw = g.get("compatibility_disjoint_coefficient", 1.0), g.get("compatibility_weight_coefficient", 0.5)

        def part(a, b, dist):
            if not a and not b:
                return 0.0
            disjoint = sum(1 for k in b if k not in a)
            d = 0.0
            for k, x in a.items():
                y = b.get(k)
                if y is None:
                    disjoint += 1
                else:
                    d += dist(x, y) * cw
            return (d + cd * disjoint) / max(len(a), len(b))

        nd = lambda x, y: abs(x.bias - y.bias) + abs(x.response - y.response) + (x.activation != y.activation) + (x.aggregation != y.aggregation)
        cdist = lambda x, y: abs(x.weight - y.weight) + (x.enabled != y.enabled)
        return part(self.nodes, other.nodes, nd) + part(self.connections, other.connections, cdist)

    def size(self):
        return len(self.nodes), sum(1 for c in self.connections.values() if c.enabled)


class _Species:
    def __init__(self, key, generation):
        self.key, self.created, self.last_improved = key, generation, generation
        self.representative, self.members, self.fitness, self.best = None, {}, None, None


class Population:
    def __init__(self, config, seed=None):
        self.config = config
        self.rng = random.Random(seed)
        self.reporters = []
        self.generation = 0
        self._next_genome, self._next_species = 1, 1
        self.species = {}
        self.best_genome = None
        self._fresh_population()
        self._speciate()

    def _fresh_population(self):
        """pop_size new genomes from the population's OWN random stream (also the restart after a complete extinction)."""
        self.population = {}
        for _ in range(self.config.pop_size):
            g = self.config.genome_type(self._next_genome)
            g.configure_new(self.config.genome_config, self.rng)
            self.population[g.key] = g
            self._next_genome += 1

    def add_reporter(self, r):
        self.reporters.append(r)

    def _speciate(self):
        gc, thr = self.config.genome_config, self.config.species_set_config.get("compatibility_threshold", 3.0)
        for s in self.species.values():
            # new representative: the member of the new population closest to the old one
            s.representative = min(self.population.values(), key=lambda g: g.distance(s.representative, gc))
            s.members = {}
        for g in self.population.values():
            best, bd = None, None
            for s in self.species.values():
                d = g.distance(s.representative, gc)
                if d < thr and (bd is None or d < bd):
                    best, bd = s, d
            if best is None:
                best = _Species(self._next_species, self.generation)
                best.representative = g
                self.species[best.key] = best
                self._next_species += 1
            best.members[g.key] = g
        self.species = {k: s for k, s in self.species.items() if s.members}

    def _reproduce(self):
        cfg, rc, st = self.config, self.config.reproduction_config, self.config.stagnation_config
        func = {"max": max, "min": min, "mean": lambda v: sum(v) / len(v)}[st["species_fitness_func"]]
        for s in self.species.values():
            f = func([m.fitness for m in s.members.values()])
            if s.best is None or f > s.best:
                s.best, s.last_improved = f, self.generation
            s.fitness = f
        ranked = sorted(self.species.values(), key=lambda s: s.fitness)
        alive = []
        for idx, s in enumerate(ranked):
            stagnant = self.generation - s.last_improved >= st["max_stagnation"]
            if stagnant and len(ranked) - idx > st["species_elitism"]:
                continue
            alive.append(s)
        if not alive:
            self.species = {}
            return {}
        fits = [m.fitness for s in alive for m in s.members.values()]
        lo, rng_f = min(fits), max(1.0, max(fits) - min(fits))
        adj = [(sum(m.fitness for m in s.members.values()) / len(s.members) - lo) / rng_f for s in alive]
        total = sum(adj)
        min_size = max(rc["min_species_size"], rc["elitism"])
        spawn = [max(min_size, int(round(cfg.pop_size * (a / total if total > 0 else 1.0 / len(alive))))) for a in adj]
        scale = cfg.pop_size / float(sum(spawn))
        spawn = [max(min_size, int(round(n * scale))) for n in spawn]
        new_pop = {}
        self.species = {}
        for s, n in zip(alive, spawn):
            old = sorted(s.members.values(), key=lambda m: m.fitness, reverse=True)
            s.members = {}
            self.species[s.key] = s
            for m in old[:rc["elitism"]]:
                if n <= 0:
                    break
                new_pop[m.key] = m
                n -= 1
            cut = max(2, int(math.ceil(rc["survival_threshold"] * len(old))))
            parents = old[:cut]
            while n > 0:
                p1, p2 = self.rng.choice(parents), self.rng.choice(parents)
                child = cfg.genome_type(self._next_genome)
                self._next_genome += 1
                child.configure_crossover(p1, p2, self.rng)
                child.mutate(cfg.genome_config, self.rng)
                new_pop[child.key] = child
                n -= 1
        return new_pop

    def run(self, fitness_function, n=None):
        k = 0
        while n is None or k < n:
            k += 1
            t0 = time.time()
            for r in self.reporters:
                r.start_generation(self.generation)
            fitness_function(list(self.population.items()), self.config)
            for g in self.population.values():
                if g.fitness is None:
                    raise RuntimeError("Fitness not assigned to genome %d" % g.key)
                if g.fitness != g.fitness or g.fitness in (float("inf"), float("-inf")):
                    # the reference's scorers return NaN for zero-length vectors; NaN makes max()/sorted()
                    # order-dependent, so rank such genomes below everything finite
                    g.fitness = -1e300 if g.fitness != float("inf") else 1e300
            best = max(self.population.values(), key=lambda g: g.fitness)
            if self.best_genome is None or best.fitness > self.best_genome.fitness:
                self.best_genome = best
            for r in self.reporters:
                r.post_evaluate(self.config, self.population, self.species, best, time.time() - t0)
            if not self.config.no_fitness_termination:
                crit = {"max": max, "min": min, "mean": lambda v: sum(v) / len(v)}[self.config.fitness_criterion]
                if crit([g.fitness for g in self.population.values()]) >= self.config.fitness_threshold:
                    break
            self.population = self._reproduce()
            if not self.species:
                if not self.config.reset_on_extinction:
                    raise RuntimeError("complete extinction")
                # neat-python's reset_on_extinction: a new random population; reporters, generation counter, best genome and
                # the seeded random stream are kept (re-running __init__ dropped the reporters and reseeded from the clock)
                self.species = {}
                self._fresh_population()
            self._speciate()
            for r in self.reporters:
                r.end_generation(self.config, self.population, self.species, self.generation, self)
            self.generation += 1
        return self.best_genome


class _Reporter:
    def start_generation(self, generation): pass
    def post_evaluate(self, config, population, species, best, seconds): pass
    def end_generation(self, config, population, species, generation, pop): pass


class StdOutReporter(_Reporter):
    def __init__(self, show_species_detail=False):
        self.detail, self.times = show_species_detail, []

    def start_generation(self, generation):
        print("\n ****** Running generation %d ****** \n" % generation)

    def post_evaluate(self, config, population, species, best, seconds):
        f = [g.fitness for g in population.values()]
        mean = sum(f) / len(f)
        sd = (sum((x - mean) ** 2 for x in f) / len(f)) ** 0.5
        print("Population's average fitness: %.5f stdev: %.5f" % (mean, sd))
        print("Best fitness: %.5f - size: %r - id %d" % (best.fitness, best.size(), best.key))
        self.times.append(seconds)
        print("Population of %d members in %d species; Generation time: %.3f sec (%.3f average)"
              % (len(population), len(species), seconds, sum(self.times[-10:]) / len(self.times[-10:])))


class StatisticsReporter(_Reporter):
    def __init__(self):
        self.most_fit_genomes, self.generation_statistics = [], []

    def post_evaluate(self, config, population, species, best, seconds):
        self.most_fit_genomes.append(best)
        self.generation_statistics.append({sid: {gid: g.fitness for gid, g in s.members.items()} for sid, s in species.items()})

    def get_fitness_mean(self):
        return [sum(f for s in gen.values() for f in s.values()) / max(1, sum(len(s) for s in gen.values())) for gen in self.generation_statistics]

    def best_genome(self):
        return max(self.most_fit_genomes, key=lambda g: g.fitness)


class Checkpointer(_Reporter):
    def __init__(self, generation_interval=100, time_interval_seconds=None, filename_prefix="neat-checkpoint-"):
        self.interval, self.prefix = generation_interval, filename_prefix

    def end_generation(self, config, population, species, generation, pop):
        if self.interval and (generation + 1) % self.interval == 0:
            reporters, pop.reporters = pop.reporters, []
            with open("%s%d" % (self.prefix, generation), "wb") as f:
                pickle.dump(pop, f)
            pop.reporters = reporters

    @staticmethod
    def restore_checkpoint(filename):
        with open(filename, "rb") as f:
            return pickle.load(f)
