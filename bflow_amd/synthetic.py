"""Deterministic synthetic inputs shaped like the reference's datasets (no dataset / network access here).

All generators use numpy RandomState with explicit seeds so that the reference (in the build container),
the CPU oracle and the HIP path (on the GPU box) see bit-identical inputs.  Shapes follow
SURVEY.md section 8(d):  DSEC voxel grids are (B, 2*5-1=9, 480, 640) (data/dsec/subsequence/twostep.py:77-83),
MultiFlow grids are (B, 41+25-1=65, H, W) (data/multiflow2d/sample.py:41-46).
"""
from __future__ import annotations

import numpy as np


def voxel_grid(batch: int, channels: int, height: int, width: int, seed: int = 1234, density: float = 0.3,
               first_sample: int = 0) -> np.ndarray:
    """(B,C,H,W) float32: Bernoulli(density) support x N(0,1) values, then zero-mean/unit-std over the
    non-zero entries per sample (what norm_voxel_grid, data/utils/representations.py:9-18, leaves behind).
    Sample i of the global batch depends only on (seed, first_sample + i), so any sharding of a global batch
    over ranks sees the same data."""
    out = np.empty((batch, channels, height, width), dtype=np.float32)
    for i in range(batch):
        rs = np.random.RandomState((seed + 7919 * (first_sample + i)) % (2 ** 31 - 1))
        vals = rs.standard_normal((channels, height, width)).astype(np.float32)
        keep = rs.uniform(size=(channels, height, width)) < density
        g = np.where(keep, vals, np.float32(0))
        nz = g != 0
        if nz.any():
            m = g[nz].mean(dtype=np.float64)
            s = g[nz].std(dtype=np.float64, ddof=1)
            g[nz] = ((g[nz] - m) / s).astype(np.float32)
        out[i] = g
    return out


def image_pair(batch: int, height: int, width: int, seed: int = 4321, first_sample: int = 0):
    """Two (B,3,H,W) uint8 images (reference time, target time): models/raft_spline/raft.py:131-134."""
    a = np.empty((batch, 3, height, width), dtype=np.uint8)
    b = np.empty_like(a)
    for i in range(batch):
        rs = np.random.RandomState((seed + 104729 * (first_sample + i)) % (2 ** 31 - 1))
        a[i] = rs.randint(0, 256, (3, height, width)).astype(np.uint8)
        b[i] = rs.randint(0, 256, (3, height, width)).astype(np.uint8)
    return a, b


def gt_flow(batch: int, height: int, width: int, seed: int = 99, sigma: float = 5.0, first_sample: int = 0) -> np.ndarray:
    """(B,2,H,W) float32 ground-truth flow ~ N(0, sigma px) for the EPE reduction."""
    out = np.empty((batch, 2, height, width), dtype=np.float32)
    for i in range(batch):
        rs = np.random.RandomState((seed + 15485863 * (first_sample + i)) % (2 ** 31 - 1))
        out[i] = (rs.standard_normal((2, height, width)) * sigma).astype(np.float32)
    return out


def events(n_events: int, height: int, width: int, t_start: int, t_end: int, seed: int = 7, int_xy: bool = False,
           border: float = 1.5):
    """Synthetic event stream: sorted int64 microsecond timestamps uniform over [t_start, t_end], polarity in
    {0,1}, x/y uniform (float32 incl. a margin outside the sensor for the DSEC rectified path,
    data/dsec/subsequence/base.py:137-143; int16 inside the sensor for the MultiFlow path, sample.py:112-212)."""
    rs = np.random.RandomState(seed)
    t = np.sort(rs.randint(t_start, t_end + 1, n_events).astype(np.int64))
    pol = rs.randint(0, 2, n_events).astype(np.int8)
    if int_xy:
        x = rs.randint(0, width, n_events).astype(np.int16)
        y = rs.randint(0, height, n_events).astype(np.int16)
    else:
        x = rs.uniform(-border, width - 1 + border, n_events).astype(np.float32)
        y = rs.uniform(-border, height - 1 + border, n_events).astype(np.float32)
    return x, y, pol, t
