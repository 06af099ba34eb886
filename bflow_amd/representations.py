"""Event voxel grid on the GPU -- drop-in for data/utils/representations.py:9-111 (VoxelGrid.convert, norm_voxel_grid).

The reference runs `convert` on the CPU inside DataLoader workers (24 masked put_ passes per grid, single-threaded by
its own torch.set_num_threads(1)); here the events are binned by grid tile, accumulated per tile in LDS in 64-bit fixed point and
written once (K1: deterministic, no atomics on memory, no zero fill), then normalised by a three-pass reduction (K2).  Inputs must be
GPU tensors."""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import hip


def norm_voxel_grid(voxel_grid: torch.Tensor) -> torch.Tensor:
    """In place: (v - mean) / std over the non-zero entries (unbiased std; mean-only if std == 0)  -- representations.py:9-18."""
    if voxel_grid.numel() > 0:
        hip.voxel_norm(voxel_grid)
    return voxel_grid


class EventRepresentation:
    def convert(self, x, y, pol, time, t_from: Optional[int] = None, t_to: Optional[int] = None):
        raise NotImplementedError


class VoxelGrid(EventRepresentation):
    def __init__(self, channels: int, height: int, width: int):
        assert channels > 1 and height > 1 and width > 1
        self.nb_channels, self.height, self.width = channels, height, width

    def _get_dt(self, t0_center: int, t1_center: int):
        assert t1_center > t0_center
        return (t1_center - t0_center) / (self.nb_channels - 1)

    def get_extended_time_window(self, t0_center: int, t1_center: int):
        """representations.py:35-39: one extra bin on either side."""
        dt = self._get_dt(t0_center, t1_center)
        return math.floor(t0_center - dt), math.ceil(t1_center + dt)

    def convert(self, x: torch.Tensor, y: torch.Tensor, pol: torch.Tensor, time: torch.Tensor,
                t0_center: Optional[int] = None, t1_center: Optional[int] = None) -> torch.Tensor:
        """representations.py:64-111.  float x/y -> tri-linear (8 neighbours); integer x/y -> temporal-linear (2 neighbours)."""
        assert x.shape == y.shape == pol.shape == time.shape and x.ndim == 1
        assert type(t0_center) == type(t1_center)
        assert not torch.is_floating_point(time)
        if not (x.is_cuda and y.is_cuda and pol.is_cuda and time.is_cuda):
            raise hip.BflowHipError("VoxelGrid.convert (bflow_amd) needs GPU tensors; the CPU restatement is oracle-only")
        if t0_center is None:      # representations.py:78-79 (costs one device->host read; pass the centres to avoid it)
            t0_center, t1_center = int(time[0]), int(time[-1])
        if torch.is_floating_point(x):
            xs, ys = x.float().contiguous(), y.float().contiguous()
        else:
            assert not torch.is_floating_point(y)
            # int8 / uint8 / int16 coordinates are widened or kept (no value can change); anything wider goes to the int32 kernel, so a
            # coordinate > 32767 is never wrapped into the grid by a narrowing cast (the reference indexes with x.long())
            narrow = x.dtype in (torch.int8, torch.uint8, torch.int16) and y.dtype in (torch.int8, torch.uint8, torch.int16)
            xy_dtype = torch.int16 if narrow else torch.int32
            if not narrow:
                lim = 2 ** 31 - 1
                x, y = x.clamp(-lim, lim), y.clamp(-lim, lim)      # int64 outliers saturate (and are then dropped by the flat-index rule)
            xs, ys = x.to(xy_dtype).contiguous(), y.to(xy_dtype).contiguous()
        grid = torch.empty((self.nb_channels, self.height, self.width), dtype=torch.float32, device=x.device)   # K1 writes every cell
        hip.voxel_grid(xs, ys, pol.to(torch.int8).contiguous(), time.to(torch.int64).contiguous(), int(t0_center), int(t1_center), grid)
        return grid
