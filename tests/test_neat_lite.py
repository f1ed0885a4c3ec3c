"""CPU: the neat-python stand-in used by examples/evolve_illusion.py."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_neat_lite_runs_generations_and_feeds_the_flattener(tmp_path):
    from evolutionary_illusion_generator_amd import genome, neat_lite as neat
    cfg = neat.Config(neat.DefaultGenome, neat.DefaultReproduction, neat.DefaultSpeciesSet, neat.DefaultStagnation,
                      os.path.join(ROOT, "examples", "circles_neat.cfg"))
    assert cfg.pop_size == 64 and cfg.genome_config.input_keys == [-1, -2] and cfg.genome_config.output_keys == [0, 1, 2]
    p = neat.Population(cfg, seed=1)
    stats = neat.StatisticsReporter()
    p.add_reporter(stats)
    p.add_reporter(neat.Checkpointer(2, filename_prefix=str(tmp_path / "ckpt-")))
    calls = []

    def eval_genomes(genomes, config):
        calls.append(len(genomes))
        for gid, g in genomes:
            f = genome.flatten_genome(g, config)          # every evolved genome must be renderable
            assert len(f["out_node"]) == 3 and f["edge_off"][-1] == len(f["edge_src"])
            g.fitness = float(len(f["act"])) / 50.0 + 0.01 * (gid % 7)

    winner = p.run(eval_genomes, 4)
    assert len(calls) == 4 and all(c >= 40 for c in calls)
    assert winner.fitness == max(g.fitness for g in stats.most_fit_genomes)
    n_nodes = [g.size()[0] for g in p.population.values()]
    assert max(n_nodes) > min(n_nodes)                    # structural mutation happened
    acts = {n.activation for g in p.population.values() for n in g.nodes.values()}
    assert len(acts) >= 4 and acts <= set(cfg.genome_config.activation_options)
    q = neat.Checkpointer.restore_checkpoint(str(tmp_path / "ckpt-1"))
    assert isinstance(q, neat.Population) and len(q.population) >= 40
    q.run(eval_genomes, 1)
