"""hipGraph capture/replay of RAFTSpline.forward.

At batch 1 a forward is ~400 small launches (5 encoder passes + 12 update iterations of ~22 kernels); the CPU cannot
issue them fast enough, so the forward is captured ONCE per input signature into a hipGraph and replayed.  Capture uses
torch's stream-capture plumbing (torch.cuda.CUDAGraph == hipGraph on ROCm), which also gives the captured region a
private memory pool: every intermediate (correlation volume, pyramid, workspaces) becomes a static buffer of the graph.
Our kernels are launched through the C ABI on torch's current stream, so they are recorded like any other node; the
look-up kernel's plane table and Bezier coefficients travel as kernel arguments and are frozen into the graph, which is
valid because they only depend on the signature (shapes, config) and on buffers owned by the graph.
"""
from __future__ import annotations

import gc
from typing import Dict, Optional, Tuple

import torch
from . import corr as _corr      # the kernel-selection switches are part of a captured graph's identity


class _Captured:
    def __init__(self, model, voxel_grid, images, iters, flow_init, test_mode):
        self.static_voxel = None if voxel_grid is None else voxel_grid.clone()
        self.static_images = None if images is None else [x.clone() for x in images]
        self.static_init = None if flow_init is None else flow_init.clone()
        # warm-up on a side stream: MIOpen algorithm search, lazy module state, allocator growth
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                model._forward_impl(self.static_voxel, self.static_images, iters, self.static_init, test_mode)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # The cyclic garbage collector must not run inside the capture: if it finalises an OLDER captured graph there (a model
        # that went out of scope, e.g. the previous test's), hipGraphExecDestroy is refused while the stream is capturing and
        # torch's destructor aborts the process.  Collect now, keep the collector off until the capture has ended.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            # (a HIGH-priority capture stream was tried so that the forked context encoder yields to the feature encoder: the replay took
            # 8.8 ms instead of 3.7 -- priority queues serialise badly on this runtime; default priority everywhere)
            with torch.cuda.graph(self.graph):
                self.low, self.ups = model._forward_impl(self.static_voxel, self.static_images, iters, self.static_init, test_mode)
        finally:
            if gc_was_on:
                gc.enable()

    @staticmethod
    def _same_buffer(src: torch.Tensor, dst: torch.Tensor) -> bool:
        """True only when `src` IS the static buffer's memory in the static buffer's layout: same address AND same strides (a transposed /
        strided view that merely starts at the same address holds different contents and must be copied)."""
        return src is dst or (src.data_ptr() == dst.data_ptr() and src.stride() == dst.stride() and src.shape == dst.shape
                              and src.dtype == dst.dtype)

    def replay(self, voxel_grid, images, flow_init):
        # (a frame that already lives in the static buffer is not copied onto itself)
        if voxel_grid is not None and not self._same_buffer(voxel_grid, self.static_voxel):
            self.static_voxel.copy_(voxel_grid)
        if images is not None:
            for dst, src in zip(self.static_images, images):
                if not self._same_buffer(src, dst):
                    dst.copy_(src)
        if flow_init is not None and not self._same_buffer(flow_init, self.static_init):
            self.static_init.copy_(flow_init)
        self.graph.replay()
        return self.low, self.ups


class GraphCache:
    """One captured graph per (shapes, dtypes, iters, test_mode, has flow_init).  Outputs are the graph's static buffers:
    they are overwritten by the next replay of the same signature (clone them to keep them)."""

    MAX_GRAPHS = 8      # signatures kept; a graph owns all of its intermediates (>= 0.4 GB per DSEC-sized sample)

    def __init__(self, model):
        self.model = model
        self._graphs: Dict[Tuple, _Captured] = {}
        self._weights_key = None

    def _weights_signature(self):
        """Storage + in-place version of every parameter and buffer.  The engine's packed weights (and folded BatchNorm terms) are
        frozen into a captured graph; load_state_dict / optimizer steps / .to() after the capture must invalidate it."""
        m = self.model
        return tuple((t.data_ptr(), t._version) for t in list(m.parameters()) + list(m.buffers()))

    @staticmethod
    def _sig(t: Optional[torch.Tensor]):
        return None if t is None else (tuple(t.shape), t.dtype, t.device.index)

    def run(self, voxel_grid, images, iters: int, flow_init, test_mode: bool):
        key = (self._sig(voxel_grid), None if images is None else tuple(self._sig(x) for x in images), int(iters),
               self._sig(flow_init), bool(test_mode), self.model.resolved_corr_precision(), _corr.FUSE_POOL1)
        wkey = self._weights_signature()
        if wkey != self._weights_key:
            self.clear()                         # destroyed here, outside any capture
            self._weights_key = wkey
        cap = self._graphs.pop(key, None)
        if cap is None:
            while len(self._graphs) >= self.MAX_GRAPHS:
                torch.cuda.synchronize()                        # (a graph is only destroyed when its last replay has finished, see clear())
                self._graphs.pop(next(iter(self._graphs)))      # least recently used first (dict order = recency, see below)
            cap = _Captured(self.model, voxel_grid, images, iters, flow_init, test_mode)
        self._graphs[key] = cap                  # (re-)insert at the end: most recently used
        return cap.replay(voxel_grid, images, flow_init)

    def clear(self):
        """Destroys the captured graphs -- after the device has drained: releasing a hipGraphExec whose last replay is still running is not
        safe on this runtime (intermittent segmentation fault in a later graph launch)."""
        if self._graphs:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self._graphs.clear()

    def __del__(self):
        try:
            self.clear()
        except Exception:
            pass
