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
from operator import attrgetter
from typing import Dict, Optional, Tuple

import torch
from . import corr as _corr      # the kernel-selection switches are part of a captured graph's identity


_VERSION = attrgetter("_version")


class _Captured:
    def __init__(self, model, voxel_grid, images, iters, flow_init, test_mode):
        # The graph and its static buffers outlive this call and are written in place by later replays (`static_voxel.copy_`): they must be
        # ordinary tensors even when the FIRST forward runs under torch.inference_mode() (val.py:75) -- an inference tensor cannot be
        # updated in place outside inference mode, i.e. by a later replay under plain no_grad.
        with torch.inference_mode(False), torch.no_grad():
            self._capture(model, voxel_grid, images, iters, flow_init, test_mode)

    def _capture(self, model, voxel_grid, images, iters, flow_init, test_mode):
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


class WeightsWatch:
    """Tells a graph cache whether the model's weights are still the ones frozen into its graphs -- in O(1) per call."""

    def __init__(self, model):
        self.model = model
        self._tensors = None          # parameters + buffers, listed once per weight generation (the module-tree walk costs ~0.3 ms)
        self._gen = None              # model._weights_gen this list / signature was taken at
        self._full = None             # (data_ptr, _version) of every tensor: the exact identity of the frozen weights
        self._vsum = None             # sum of the in-place version counters (monotone: any in-place edit changes it)

    def _walk(self) -> bool:
        """The full signature: storage + in-place version of every parameter and buffer.  The engine's packed weights (and folded
        BatchNorm terms) are frozen into a captured graph; whatever changes them must invalidate it.  True = changed."""
        m = self.model
        every = list(m.parameters()) + list(m.buffers())
        # (tensors created under torch.inference_mode() carry no version counter: they cannot be edited in place outside it either)
        self._tensors = [t for t in every if not t.is_inference()]
        full = tuple((t.data_ptr(), -1 if t.is_inference() else t._version) for t in every)
        changed = full != self._full
        self._full = full
        self._vsum = sum(map(_VERSION, self._tensors))
        self._gen = getattr(m, "_weights_gen", None)
        return changed

    def changed(self) -> bool:
        """Per call: ONE integer comparison -- `model._weights_gen`, bumped by RAFTSpline.load_state_dict / _apply (.to(), .cuda(), .half())
        / train() -- and, as the net under in-place edits that no hook sees (`p.data.mul_()`, an optimiser stepped in eval mode), the sum of
        the version counters over the CACHED tensor list (C-level map, no module-tree walk: ~14 us for 169 tensors against ~300 us for
        the walk).  The full (data_ptr, version) walk runs only when one of the two differs."""
        if self._tensors is None or getattr(self.model, "_weights_gen", None) != self._gen:
            return self._walk()
        if sum(map(_VERSION, self._tensors)) != self._vsum:
            return self._walk()
        return False


class GraphCache:
    """One captured graph per (shapes, dtypes, iters, test_mode, has flow_init).  Outputs are the graph's static buffers:
    they are overwritten by the next replay of the same signature (clone them to keep them)."""

    MAX_GRAPHS = 8      # signatures kept; a graph owns all of its intermediates (>= 0.4 GB per DSEC-sized sample)

    def __init__(self, model):
        self.model = model
        self._graphs: Dict[Tuple, _Captured] = {}
        self._weights = WeightsWatch(model)
        self.replays = 0              # graph replays served (tests, bench: "was this forward a replay?")
        self.captures = 0

    @staticmethod
    def _sig(t: Optional[torch.Tensor]):
        return None if t is None else (tuple(t.shape), t.dtype, t.device.index)

    def run(self, voxel_grid, images, iters: int, flow_init, test_mode: bool):
        key = (self._sig(voxel_grid), None if images is None else tuple(self._sig(x) for x in images), int(iters),
               self._sig(flow_init), bool(test_mode), self.model.resolved_corr_precision(), _corr.FUSE_POOL1)
        if self._weights.changed():
            self.clear()                         # destroyed here, outside any capture
        cap = self._graphs.pop(key, None)
        if cap is None:
            while len(self._graphs) >= self.MAX_GRAPHS:
                torch.cuda.synchronize()                        # (a graph is only destroyed when its last replay has finished, see clear())
                self._graphs.pop(next(iter(self._graphs)))      # least recently used first (dict order = recency, see below)
            cap = _Captured(self.model, voxel_grid, images, iters, flow_init, test_mode)
            self.captures += 1
        self.replays += 1
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
