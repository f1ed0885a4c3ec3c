"""ORACLE (test infrastructure): the whole fitness path on the CPU for one genome / one image.

Order of stages as in get_fitnesses_neat (/root/reference/generate_illusion.py:478-616) and, for the single-image
pairing, get_vectors + calculate_fitness (/root/reference/fitness_calculator.py:468-548):
  render (oracle.cppn) -> PredNet roll-out (eig_oracle.c) -> Lucas-Kanade (eig_oracle.c) -> score (oracle.scores).
"""
import numpy as np

import oracle
from oracle import cppn, scores

PAIR_POPULATION, PAIR_SINGLE = 0, 1


def image_vectors(img, weights, channels, w, h, pairing=PAIR_POPULATION, n_repeat=20, requant=False, lk_params=None, flow="lk",
                  fb_params=None):
    """img uint8 [C,H,W] -> flow vectors float32 [n,4].  flow: "lk" (the reference's call) or "farneback" (oracle/farneback.c)."""
    n_ext = 1 if pairing == PAIR_POPULATION else 2
    frames = oracle.prednet_rollout(weights, channels, w, h, img, n_repeat=n_repeat, n_ext=n_ext, requant=requant)
    fn = (lambda a, b: oracle.lucas_kanade(a, b, lk_params)) if flow == "lk" else (lambda a, b: oracle.farneback(a, b, fb_params))
    if pairing == PAIR_POPULATION:  # prediction@20 -> first extension (generate_illusion.py:543-550)
        return fn(frames[n_repeat - 1], frames[n_repeat])
    return fn(img, frames[n_repeat + 1])  # original -> 2nd extension (fitness_calculator.py:493-498)


def image_fitness(img, weights, channels, w, h, structure, pairing=PAIR_POPULATION, **kw):
    v = image_vectors(img, weights, channels, w, h, pairing=pairing, **kw)
    return scores.fitness_from_vectors(structure, v.astype(np.float64), w, h)


def render_chw(genome, config, grid, c_dim, w, h, bg=1, gradient=1):
    r = cppn.render(grid, genome, config, c_dim, w, h, bg=bg, gradient=gradient)
    return np.ascontiguousarray(r.transpose(2, 0, 1) if r.ndim == 3 else r[None])


def genome_fitness(genome, config, grid, weights, channels, w, h, structure, pairing=PAIR_POPULATION, bg=1, gradient=1, **kw):
    img = render_chw(genome, config, grid, channels[0], w, h, bg=bg, gradient=gradient)
    return image_fitness(img, weights, channels, w, h, structure, pairing=pairing, **kw)
