"""CPPN coordinate grids, vectorised (host side, computed once per (structure, W, H) and uploaded).

Product-side counterpart of ``create_grid`` / ``fill_circle`` (/root/reference/generate_illusion.py:196-317, 38-117).
The grid is shared by every genome of every generation (generate_illusion.py:501), so it is not on the per-genome
critical path; it is built with numpy in the same float64 operation order as the reference's scalar loops, which
makes it bit-identical to them (tests/test_oracle_golden.py checks against fixtures produced by the reference itself).

Deviations: Bands planes are returned as (H, W) -- the reference returns (1, H*W, 1)
arrays that its own renderer cannot index -- and sizes with W % 10 != 0 or H % 4 != 0, where the reference
raises, are handled by leaving the remainder columns / rows at coordinate 0.
"""
import math
from enum import IntEnum

import numpy as np


class StructureType(IntEnum):  # generate_illusion.py:25-29
    Bands = 0
    Circles = 1
    Free = 2
    CirclesFree = 3


def _ring_edges():
    e = np.zeros(10)
    e[9] = 1
    for i in range(2, 11):
        e[10 - i] = e[10 - i + 1] * 1.5
    return e / e[0]


def _polar(x, y):
    with np.errstate(divide="ignore", invalid="ignore"):
        theta = np.where(x == 0, math.pi / 2.0, np.arctan(y * 1.0 / x))
    return np.where(x < 0, theta + math.pi, theta)


def ring_grid(x, y, max_radius, direction=1, circles_free=False, no_theta=False):
    """Vectorised fill_circle: x, y are float64 arrays of offsets from the circle centre -> (r, theta).
    no_theta: fill_circle called with a structure that is neither Circles nor CirclesFree (generate_illusion.py:69-105 has no
    branch for Bands / Free, so theta keeps its initial 0 while r is still the ring coordinate)."""
    edges = _ring_edges()
    r_total = np.sqrt(x * x + y * y)
    inside = r_total <= max_radius / 2
    radius = np.minimum(1, r_total / (max_radius / 2))
    r = np.full(x.shape, -1.0)
    ring = np.zeros(x.shape, dtype=np.int64)
    found = np.zeros(x.shape, dtype=bool)
    for i in range(1, 9):
        hit = (~found) & (radius > edges[i])
        ri = (radius - edges[i]) / (edges[i - 1] - edges[i])
        if direction < 0:
            ri = 1 - ri
        r = np.where(hit, ri, r)
        ring = np.where(hit, 9 - i, ring)
        found |= hit
    theta = _polar(x, y)
    theta = np.where(ring % 2 == 1, theta + math.pi / 4.0, theta)
    if not circles_free:
        theta = theta % (math.pi / 6.0)
    if direction < 0:
        theta = (math.pi / 6.0) - theta
    blank = (r > 0.9) | (r < 0.1) | (~inside)
    if no_theta:
        return np.where(blank, -1.0, r / 0.8), np.zeros(x.shape)
    return np.where(blank, -1.0, r / 0.8), np.where(blank, 0.0, theta)


def create_grid(structure, x_res=32, y_res=32, scaling=1.0):
    """Same call signature and dict keys as the reference's create_grid."""
    structure = int(structure)
    xs = np.linspace(-1 * scaling, scaling, num=x_res)
    ys = np.linspace(-1 * scaling, scaling, num=y_res)
    if structure == StructureType.Free:
        x_mat = np.tile(xs[None, :], (y_res, 1))
        y_mat = np.tile(ys[:, None], (1, x_res))
    elif structure == StructureType.Circles:
        xx, yy = np.meshgrid(np.arange(x_res) - x_res / 2, np.arange(y_res) - y_res / 2)
        x_mat, y_mat = ring_grid(xx, yy, y_res, 1)
    elif structure == StructureType.CirclesFree:
        r_len = int(y_res / 6)
        x, y = np.meshgrid(np.arange(x_res) - x_res / 2, np.arange(y_res) - y_res / 2)
        r_total = np.sqrt(x * x + y * y)
        x_mat = (np.minimum(r_total, y_res / 2) % r_len) / r_len
        theta = _polar(x, y)
        theta = np.where((r_total / r_len).astype(np.int64) % 2 == 1, theta + math.pi / 4.0, theta)
        y_mat = np.where(r_total < y_res / 2, theta, 0.0)
    elif structure == StructureType.Bands:
        bands, tiles, gap = 4, 10, 10
        band_h, tile_w = int(y_res / bands), int(x_res / tiles)
        ramp_y = np.concatenate((np.linspace(-scaling / bands, scaling / bands, num=max(band_h - gap, 0)), np.zeros(min(gap, band_h))))
        ramp_x = np.linspace(-scaling / tiles, scaling / tiles, num=tile_w)
        col = np.zeros(x_res)
        col[:tile_w * tiles] = np.tile(ramp_x, tiles)
        row = np.zeros(y_res)
        row[:band_h * bands] = np.tile(ramp_y, bands)
        sign = np.ones(y_res)
        start = band_h
        while band_h > 0 and start < y_res:
            sign[max(0, start - gap):start] = 0
            stop = min(y_res, start + band_h)
            sign[max(stop - gap, 0):stop] = 0
            sign[start:stop] = -sign[start:stop]
            start += 2 * band_h
        x_mat = sign[:, None] * col[None, :]
        y_mat = np.tile(row[:, None], (1, x_res))
    else:
        raise ValueError("unknown structure %r" % (structure,))
    return {"x_mat": np.ascontiguousarray(x_mat, dtype=np.float64), "y_mat": np.ascontiguousarray(y_mat, dtype=np.float64)}


def enhanced_image_grid(x_res, y_res, structure):
    """3x3 circles plus 2x2 overlaid circles with alternating direction (generate_illusion.py:121-193)."""
    free = int(structure) == StructureType.CirclesFree
    flat = int(structure) not in (StructureType.Circles, StructureType.CirclesFree)  # Bands / Free: theta stays 0
    y_step, x_step = int(y_res / 3), int(x_res / 3)
    x_mat = np.ones((y_res, x_res)) * -1
    y_mat = np.ones((y_res, x_res)) * -1
    centers = [[x_step * x + x_step / 2, y_step * y + y_step / 2] for y in range(3) for x in range(3)]
    centers += [[x_step * x + x_step, y_step * y + x_step] for y in range(2) for x in range(2)]
    for row in range(3):
        for col in range(3):
            index = row * 3 + col
            direction = -1 if index % 2 == 0 else 1
            rx = col * x_step + np.arange(x_step)
            ry = row * y_step + np.arange(y_step)
            gx, gy = np.meshgrid(rx, ry)
            r, t = ring_grid(gx - centers[index][0], gy - centers[index][1], y_step, direction, free, flat)
            x_mat[np.ix_(ry, rx)] = r
            y_mat[np.ix_(ry, rx)] = t
    for row in range(2):
        for col in range(2):
            index = 9 + row * 2 + col
            direction = -1 if index % 2 == 0 else 1
            rx = col * x_step + np.arange(x_step) + int(x_step / 2)
            ry = row * y_step + np.arange(y_step) + int(y_step / 2)
            gx, gy = np.meshgrid(rx, ry)
            x, y = gx - centers[index][0], gy - centers[index][1]
            r, t = ring_grid(x, y, y_step, direction, free, flat)
            m = np.sqrt(x * x + y * y) < x_step / 2
            sub_x, sub_y = x_mat[np.ix_(ry, rx)], y_mat[np.ix_(ry, rx)]
            x_mat[np.ix_(ry, rx)] = np.where(m, r, sub_x)
            y_mat[np.ix_(ry, rx)] = np.where(m, t, sub_y)
    return {"x_mat": x_mat, "y_mat": y_mat}
