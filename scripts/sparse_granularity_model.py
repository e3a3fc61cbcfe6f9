#!/usr/bin/env python
"""CPU model of the range-aware contraction's work at BASELINE config 5 for different granularities (no GPU): stations in the library's
Hilbert order (mik_station_order), point blocks = compact patches of 128 grid cells (what the device sort produces) or row segments
(the caller's order), active K tiles of 16 or 8 stations; work of the symmetric form ~ (active stations)^2 / 2 per point.
Answers: what finer K tiles / smaller point blocks could still save."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, synth  # noqa: E402
from pykrige_amd import _lib  # noqa: E402

cfg = CONFIGS[5]
(x, y), _ = synth(cfg["seed"], cfg["n"], 2)
o = _lib.station_order(x, y)
xs, ys = x[o], y[o]
rng = np.random.default_rng(1)
r = cfg["params"][1]
nx = 4096
h = 1.0 / (nx - 1)
rows = []
for shape, (pw, ph) in (("row segment 128 x 1", (128, 1)), ("patch 16 x 8", (16, 8)), ("patch 8 x 8 (64 points)", (8, 8)), ("single point", (1, 1))):
    res = {}
    for tile in (16, 8, 1):
        tot_exact, tot_tiles, nblk = 0.0, 0.0, 0
        for _ in range(300):
            cx, cy = rng.uniform(0.0, 1.0 - pw * h), rng.uniform(0.0, 511 * h - ph * h)  # the first GPU's slab of the 4096^2 grid
            px = cx + h * np.arange(pw)
            py = cy + h * np.arange(ph)
            gx, gy = np.meshgrid(px, py)
            d2 = (xs[None, :] - gx.ravel()[:, None]) ** 2 + (ys[None, :] - gy.ravel()[:, None]) ** 2
            act = (d2 <= r * r).any(axis=0)
            nt = (act.size + tile - 1) // tile
            pad = np.zeros(nt * tile, bool)
            pad[:act.size] = act
            tiles_on = pad.reshape(nt, tile).any(axis=1).sum()
            tot_exact += act.sum()
            tot_tiles += tiles_on * tile
            nblk += 1
        res[tile] = (tot_exact / nblk, tot_tiles / nblk)
    rows.append((shape, res))
base = rows[1][1][16][1]
print("BASELINE config 5 (N = 8000, spherical range %.2f), first slab; stations in Hilbert order; 300 random blocks per line" % r)
print("%-28s %14s %22s %22s   work ~ n^2 relative to (patch 16 x 8, 16-station tiles)" % ("point block", "stations in range", "in active 16-tiles", "in active 8-tiles"))
for shape, res in rows:
    print("%-28s %14.0f %22.0f %22.0f   %5.2f (16)  %5.2f (8)  %5.2f (exact)" % (
        shape, res[16][0], res[16][1], res[8][1], (res[16][1] / base) ** 2, (res[8][1] / base) ** 2, (res[1][1] / base) ** 2))
