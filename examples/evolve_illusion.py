#!/usr/bin/env python3
"""Evolve illusions end to end on the MI355X engine: the flow of the reference's neat_illusion()
(/root/reference/generate_illusion.py:676-711) with neat_lite standing in for neat-python.

    python examples/evolve_illusion.py -o results -g 5 [-m model.npz] [--size small|big|256] [-s 1] [-c 3]
    python -m torch.distributed.run --nproc-per-node 8 examples/evolve_illusion.py ...   (population sharded over GPUs)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from evolutionary_illusion_generator_amd import fitness, neat_lite as neat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", "-m", default="synthetic", help="chainer npz weights, or synthetic[:seed]")
    ap.add_argument("--output_dir", "-o", default="results")
    ap.add_argument("--structure", "-s", type=int, default=1, help="0 Bands, 1 Circles, 2 Free, 3 CirclesFree")
    ap.add_argument("--config", "-cfg", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "circles_neat.cfg"))
    ap.add_argument("--size", "-wh", default="small", help="small (160x120), big (640x480) or N for NxN")
    ap.add_argument("--color_space", "-c", type=int, default=3)
    ap.add_argument("--channels", "-ch", default="3,48,96,192")
    ap.add_argument("--gradient", "-g1", type=int, default=1)
    ap.add_argument("--generations", "-g", type=int, default=3)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    w, h = {"small": (160, 120), "big": (640, 480)}.get(a.size) or (int(a.size), int(a.size))
    channels = [int(c) for c in a.channels.split(",")]
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        torch.distributed.init_process_group("nccl")
    config = neat.Config(neat.DefaultGenome, neat.DefaultReproduction, neat.DefaultSpeciesSet, neat.DefaultStagnation, a.config)

    def eval_genomes(genomes, config):
        fitness.get_fitnesses_neat(a.structure, genomes, a.model, config, w, h, channels, c_dim=a.color_space,
                                   best_dir=a.output_dir, gradient=a.gradient)

    p = neat.Population(config, seed=a.seed)  # identical on every rank: same seed, same all-gathered fitness
    p.add_reporter(neat.StdOutReporter(True))
    p.add_reporter(neat.StatisticsReporter())
    winner = p.run(eval_genomes, a.generations)
    print("winner", winner.key, winner.fitness, winner.size())


if __name__ == "__main__":
    main()
