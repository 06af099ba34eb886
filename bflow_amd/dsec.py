"""DSEC two-step sample assembly on the GPU -- SURVEY 8(f-1).

What `TwoStepSubSequence.__getitem__` (data/dsec/subsequence/twostep.py:44-100) and `BaseSubSequence` (base.py:121-204) do on one
CPU DataLoader worker per sample -- pick the current / previous flow interval, cut the event stream around each (one extra bin on
either side), rectify the raw sensor coordinates through the sequence's map, build two voxel grids, drop the temporal slice they
share, normalise -- with the arithmetic in HIP kernels:

    raw events (uint16 x / y, 0/1 polarity, int64 us)  --bflow_voxel_grid_rectified-->  grid_prev, grid_cur   (K1 + map gather)
    max |grid_prev[-1] - grid_cur[0]| < 0.5             --bflow_maxabs_diff-->              the reference's consistency assert
    [grid_prev | grid_cur[1:]]                          --bflow_voxel_norm-->               (2*bins - 1, H, W) network input (K2)

File I/O stays with the caller: `events` is any object with `window(t_start_us, t_end_us) -> (x, y, p, t)` tensors (GPU or host)
covering [t_start, t_end); `EventStream` wraps an in-memory, time-sorted stream and reproduces EventSlicer's offset arithmetic
(data/dsec/eventslicer.py:99-158).  The host-side window bookkeeping mirrors the reference line by line (asserts included).
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import hip, voxel_cache
from .representations import VoxelGrid, norm_voxel_grid


def event_window_indices(time_array: np.ndarray, time_start_us: int, time_end_us: int) -> Tuple[int, int]:
    """EventSlicer.get_time_indices_offsets (eventslicer.py:99-158): [i0, i1) with time_start <= t[i0:i1] < time_end."""
    assert time_array.ndim == 1
    if time_array[-1] < time_start_us:
        return time_array.size, time_array.size
    return int(np.searchsorted(time_array, time_start_us, side="left")), int(np.searchsorted(time_array, time_end_us, side="left"))


def twostep_windows(forward_flow_timestamps, index: int) -> List[Tuple[int, int]]:
    """twostep.py:49-66: (ts_from, ts_to) of the current interval and of the previous one (extrapolated when index == 0)."""
    out: List[Tuple[int, int]] = []
    ts_from = ts_to = None
    for idx in (index, index - 1):
        if 0 <= idx < len(forward_flow_timestamps):
            ts_from, ts_to = int(forward_flow_timestamps[idx][0]), int(forward_flow_timestamps[idx][1])
        else:
            assert idx == index - 1
            assert ts_from is not None and ts_to is not None
            dt = ts_to - ts_from
            ts_to = ts_from
            ts_from = ts_from - dt
        out.append((ts_from, ts_to))
    return out


class EventStream:
    """A time-sorted raw event stream held in memory (host numpy timestamps for the window search, payload on the GPU)."""

    def __init__(self, x: np.ndarray, y: np.ndarray, p: np.ndarray, t: np.ndarray, device="cuda"):
        assert x.shape == y.shape == p.shape == t.shape and t.ndim == 1 and t.size > 0
        self.t_host = np.ascontiguousarray(t, dtype=np.int64)
        self.x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint16)).to(device)
        self.y = torch.from_numpy(np.ascontiguousarray(y, dtype=np.uint16)).to(device)
        self.p = torch.from_numpy(np.ascontiguousarray(p, dtype=np.uint8)).to(device)
        self.t = torch.from_numpy(self.t_host).to(device)

    def get_start_time_us(self) -> int:
        return int(self.t_host[0])

    def get_final_time_us(self) -> int:
        return int(self.t_host[-1])

    def window(self, t_start_us: int, t_end_us: int):
        assert t_start_us < t_end_us
        i0, i1 = event_window_indices(self.t_host, t_start_us, t_end_us)
        return self.x[i0:i1], self.y[i0:i1], self.p[i0:i1], self.t[i0:i1]


class TwoStepAssembler:
    """EV_REPR of TwoStepSubSequence.__getitem__ from raw events.  num_bins / height / width / rectify_map as BaseSubSequence holds
    them (base.py:55-84); extended_voxel_grid = version 1 (one extra bin of events on either side, base.py:196-200)."""

    def __init__(self, num_bins: int, height: int, width: int, rectify_map, normalize_voxel_grid: bool = True, merge_grids: bool = True,
                 extended_voxel_grid: bool = True, device="cuda", voxel_grid_dir=None):
        assert num_bins >= 1
        self.num_bins, self.height, self.width = num_bins, height, width
        self.voxel_grid = VoxelGrid(num_bins, height, width)
        rm = torch.as_tensor(np.asarray(rectify_map), dtype=torch.float32) if not isinstance(rectify_map, torch.Tensor) else rectify_map.float()
        assert tuple(rm.shape) == (height, width, 2), tuple(rm.shape)                       # base.py:140
        self.rectify_events_map = rm.contiguous().to(device)
        self.normalize, self.merge_grids, self.version = normalize_voxel_grid, merge_grids, 1 if extended_voxel_grid else 0
        self.device = device
        self._bad = torch.zeros(1, dtype=torch.int32, device=device)
        self.keep_last_window = True      # assemble(): reuse the last sample's current-window grid as this sample's previous one when it is the same window
        self._kept = None
        # base.py:93-104 `load_voxel_grid`: per-window grids are cached as blosc-zstd HDF5 files (bflow_amd/voxel_cache.py)
        self.voxel_grid_dir = None
        if voxel_grid_dir is not None:
            self.voxel_grid_dir = Path(voxel_grid_dir)
            if not self.voxel_grid_dir.exists():
                os.mkdir(self.voxel_grid_dir)
            else:
                assert self.voxel_grid_dir.is_dir()

    # base.py:160-204 -------------------------------------------------------------------------------------------------
    def time_window(self, events, ts_from: int, ts_to: int):
        """(t_start, t_end, t0_center, t1_center) of the window [ts_from, ts_to]: the extended window of version 1, clamped to the recording,
        with the reference's sanity asserts (base.py:165-191).  Version 0's centres are the first / last event's times (None here)."""
        if self.version == 1:
            t_start, t_end = self.voxel_grid.get_extended_time_window(ts_from, ts_to)
            assert (ts_from - t_start) < 50000, f"ts_from: {ts_from}, t_start: {t_start}"
            assert (t_end - ts_to) < 50000, f"t_end: {t_end}, ts_to: {ts_to}"
            t0c, t1c = ts_from, ts_to
        else:
            t_start, t_end, t0c, t1c = ts_from, ts_to, None, None
        start_us, final_us = events.get_start_time_us(), events.get_final_time_us()
        assert t_start > start_us - 50000, "Do not request more than 50 ms before the minimum time. Otherwise, something might be wrong."
        assert t_end < final_us + 50000, "Do not request more than 50 ms past the maximum time. Otherwise, something might be wrong."
        t_start, t_end = max(t_start, start_us), min(t_end, final_us)
        assert t_start < t_end
        return t_start, t_end, t0c, t1c

    def window_descriptor(self, events: "EventStream", ts_from: int, ts_to: int) -> Tuple[int, int, int, int]:
        """{first event, event count, t0_center, t1_center} of the window: what bflow_voxel_grid_rectified_window reads from device memory
        (extended voxel grids only: version 0 takes its centres from the events themselves)."""
        assert self.version == 1, "device-side windows need extended_voxel_grid (the centres are the window's own timestamps)"
        t_start, t_end, t0c, t1c = self.time_window(events, ts_from, ts_to)
        i0, i1 = event_window_indices(events.t_host, t_start, t_end)
        return i0, i1 - i0, t0c, t1c

    def construct_voxel_grid(self, events, ts_from: int, ts_to: int) -> torch.Tensor:
        t_start, t_end, t0c, t1c = self.time_window(events, ts_from, ts_to)
        x, y, p, t = events.window(t_start, t_end)
        if t0c is None:                                                                     # version 0: centres = first / last event
            t0c, t1c = int(t[0]), int(t[-1])
        grid = torch.empty((self.num_bins, self.height, self.width), dtype=torch.float32, device=self.device)   # K1 writes every cell
        dev = lambda a, dt: a.to(device=self.device, dtype=dt).contiguous()
        hip.voxel_grid_rectified(dev(x, torch.uint16), dev(y, torch.uint16), dev(p, torch.uint8), dev(t, torch.int64),
                                    self.rectify_events_map, t0c, t1c, grid, self._bad)
        return grid

    # base.py:205-222 -------------------------------------------------------------------------------------------------
    def get_voxel_grid(self, events, ts_from: int, ts_to: int, file_index: Optional[int] = None) -> torch.Tensor:
        """`_get_voxel_grid`: with a cache directory the grid of file `{file_index:06d}.h5` is loaded if it exists, else built and saved."""
        if self.voxel_grid_dir is None or file_index is None:
            return self.construct_voxel_grid(events, ts_from, ts_to)
        f = voxel_cache.dsec_voxel_grid_file(self.voxel_grid_dir, file_index)
        cached = voxel_cache.h5_to_np_array(f) if f.exists() else None     # None: unreadable / truncated file = a miss, rebuilt and overwritten
        if cached is None:
            grid = self.construct_voxel_grid(events, ts_from, ts_to)
            voxel_cache.np_array_to_h5(grid.cpu().numpy(), f)
            return grid
        return torch.from_numpy(cached).to(self.device)

    # twostep.py:44-92 ------------------------------------------------------------------------------------------------
    def assemble(self, events, forward_flow_timestamps, index: int, check: bool = True, flow_file_index: Optional[int] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`flow_file_index` (the index in the flow file's name, twostep.py:41) selects the cache files: it for the current window,
        it - 2 for the previous one (100-ms steps, twostep.py:63-64).  `out`: a caller-owned (2 * bins - 1, H, W) buffer for the merged grid
        (a frame stream double-buffers it, bflow_amd/pipeline.py EventFramePipeline)."""
        (cf, ct), (pf, pt) = twostep_windows(forward_flow_timestamps, index)
        self._bad.zero_()     # the counter is per sample: one bad sample must not fail (or hide in) the following ones
        ev_cur = self.get_voxel_grid(events, cf, ct, flow_file_index)
        # In a 100-ms-step sequence the previous window of sample k + 1 is the current window of sample k (twostep.py:63-64): its grid is kept in
        # memory (one 5-bin grid; the reference keeps the same per-window grids on disk, base.py:93-104) -- one K1 per sample instead of two.
        key_prev = (id(events), int(pf), int(pt))
        keep = self.keep_last_window and self.merge_grids          # (the unmerged form normalises the two grids in place)
        if keep and self._kept is not None and self._kept[0] == key_prev:
            ev_prev = self._kept[1]
        else:
            ev_prev = self.get_voxel_grid(events, pf, pt, None if flow_file_index is None else flow_file_index - 2)
        self._kept = ((id(events), int(cf), int(ct)), ev_cur) if keep else None
        if check:
            # base.py:141-142: raw coordinates must lie inside the map (one device->host read, like the reference's x.max())
            assert int(self._bad) == 0, f"{int(self._bad)} events outside the {self.height}x{self.width} rectification map"
        if self.merge_grids:
            if check:
                d = float(hip.maxabs_diff(ev_prev[-1], ev_cur[0]))
                assert d < 0.5, f"{d}"                                                      # twostep.py:83
            if out is None:
                out = torch.empty((2 * self.num_bins - 1, self.height, self.width), dtype=torch.float32, device=self.device)
            assert tuple(out.shape) == (2 * self.num_bins - 1, self.height, self.width) and out.is_contiguous()
            if self.normalize:
                # torch.cat((ev_repr_0, ev_repr_1[1:])) and norm_voxel_grid of the result (twostep.py:77-85) as ONE statistics pass over the
                # two per-window grids + one pass that writes the merged, normalised grid (bflow_voxel_merge_norm; round 5: two copies + K2)
                return hip.voxel_merge_norm(ev_prev, ev_cur[1:], out)
            out[:self.num_bins].copy_(ev_prev)                                              # torch.cat((ev_repr_0, ev_repr_1[1:]))
            out[self.num_bins:].copy_(ev_cur[1:])
            return out
        grids = [norm_voxel_grid(g) for g in (ev_prev, ev_cur)] if self.normalize else [ev_prev, ev_cur]
        return torch.stack(grids)
