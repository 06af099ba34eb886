"""Several independent forwards as parallel branches of ONE hipGraph (`ConcurrentRunner`).

Why: the GRU loop of a forward is a chain of ~130 dependent launches of <= 320 workgroups (batch 1) that leaves most CUs idle, and
even at batch 8 its launches do not fill the chip all the time.  Independent inputs -- the micro-batches of a rank's shard of the
global batch (BASELINE configs[3]: 64 frames in micro-batches of 8), or consecutive frames of a validation stream -- can fill that
idle capacity, but only inside one graph: hipGraph replays launched from different streams execute one after the other on this
runtime (tools/multistream_probe.py), while parallel branches of one graph do run concurrently.

    capture stream :  forward(input 0)
    stream 1 .. S-1:  forward(input s)          all forked from the capture stream, joined at the end

Measured on MI355X (tools/two_frame_probe.py; one replay = two forwards): batch 1: 6.63 ms vs 8.10 ms one after the other
(302 vs 247 frames/s); batch 8: 44.4 ms vs 48.3 ms (360 vs 332 frames/s).  The forwards' own side-stream branches are switched off
inside the capture: a stream forked from a forked stream crashes hipGraph instantiation on this runtime, and with several forwards in
flight they do not pay (flat 3-stream variant: 287 frames/s at batch 1).  What was also built and measured, and is NOT kept: a
software pipeline with ONE frame per replay (frame k's encoders + volume next to frame k-1's GRU loop, double-buffered state): 3.83-3.90 ms
per frame against 4.12 ms -- the encoder phase already fills the chip, and the power-limited clock under matrix-core load stretches
the latency-bound GRU chain running next to it; GRU loop next to GRU loop is the overlap that pays.

The arithmetic of every forward is unchanged: outputs are bit-identical to `model(voxel_grid=...)`
(tests/test_hip_parity.py::test_concurrent_runner_matches_sequential).  Returned tensors are the graph's static buffers: valid until
the next call (clone to keep).
"""
from __future__ import annotations

import gc
from typing import List, Optional, Sequence, Tuple

import torch

from . import hip
from .bezier import BezierCurves


class ConcurrentRunner:
    def __init__(self, model, iters: int = 12, streams: int = 2):
        # Two parallel chains is what every captured forward of this package has (capture stream + one side stream).  Graph executables
        # with MORE parallel chains crash inside hipGraphLaunch (hip::Graph::UpdateStreams, SIGSEGV) once other multi-stream graph
        # executables have been destroyed in the same process (ROCm 7.2 runtime; reproduced by the GPU test-suite, not by a fresh process).
        if streams not in (1, 2):
            raise ValueError("ConcurrentRunner: 1 or 2 forwards in flight (3+ parallel graph branches are not safe on this runtime)")
        self.model, self.iters, self.streams = model, iters, streams
        self._graph = None
        self._sig = None
        from .graph import WeightsWatch
        self._weights = WeightsWatch(model)

    def _capture(self, voxels: Sequence[torch.Tensor]):
        m, dev = self.model, voxels[0].device
        assert not m.training, "inference only"
        with torch.inference_mode(False):
            self._static = [v.clone() for v in voxels]
        self._side = [torch.cuda.Stream(device=dev) for _ in range(self.streams - 1)]
        was = hip.BRANCHING
        hip.BRANCHING = False                 # flat: every stream forks from the capture stream, none from a forked one
        try:
            # (inference_mode(False): the static inputs are written in place by every later call, whatever mode that call runs in -- graph.py)
            with torch.inference_mode(False), torch.no_grad():
                for v in self._static:        # warm-up outside the capture (lazy packing, allocator growth)
                    m._forward_impl(v, None, self.iters, None, True)
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                gc.collect()
                gc_on = gc.isenabled()
                gc.disable()                  # (graph.py: a collection inside a capture can abort the process)
                try:
                    with torch.cuda.graph(g):
                        cur = torch.cuda.current_stream(dev)
                        outs = [None] * self.streams
                        for s, side in enumerate(self._side, start=1):
                            side.wait_stream(cur)
                            with torch.cuda.stream(side):
                                outs[s] = m._forward_impl(self._static[s], None, self.iters, None, True)
                        outs[0] = m._forward_impl(self._static[0], None, self.iters, None, True)
                        for side in self._side:
                            cur.wait_stream(side)
                finally:
                    if gc_on:
                        gc.enable()
        finally:
            hip.BRANCHING = was
        self._graph, self._outs = g, [(low, ups[-1]) for low, ups in outs]

    def __call__(self, voxels: Sequence[torch.Tensor]) -> List[Tuple[BezierCurves, BezierCurves]]:
        """`streams` voxel grids of one shape -> [(low-resolution curves, full-resolution curves)] in the same order."""
        assert len(voxels) == self.streams
        if not all(v.is_cuda for v in voxels):
            raise hip.BflowHipError("ConcurrentRunner runs on MI355X only: move the inputs to the GPU")
        sig = (tuple(tuple(v.shape) for v in voxels), voxels[0].dtype, voxels[0].device.index)
        stale = self._weights.changed()       # (graph.py: one integer + a 14-us version sum per call, the full walk only on a mismatch)
        with torch.cuda.device(voxels[0].device), torch.no_grad():
            if self._graph is None or stale or sig != self._sig:
                self.close()                  # an older graph is destroyed here, outside any capture and after its last replay has finished
                self._capture(voxels)
                self._sig = sig
            for dst, src in zip(self._static, voxels):
                dst.copy_(src)
            self._graph.replay()
        return [(BezierCurves(low), BezierCurves(up)) for low, up in self._outs]

    def close(self):
        """Releases the captured graph.  The device is synchronised first: destroying a hipGraphExec whose last replay is still running
        is not safe on this runtime (an intermittent segmentation fault in a LATER graph launch was traced to it)."""
        if self._graph is not None:
            torch.cuda.synchronize(self._static[0].device)
            self._graph = None
            self._outs = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EventFramePipeline:
    """Raw events -> flow for a STREAM of frames: the voxel-grid assembly of frame k + 1 (2 x K1 + merge + K2, bflow_amd/dsec.py; ~0.2 ms of
    HBM-bound launches) runs on its own stream next to the GRU loop of frame k (a chain of small launches that leaves most CUs idle), the
    forward of frame k + 1 waits for it through an event.  Two grid buffers alternate; the forward is whatever `model(...)` does (a graph
    replay under eval + inference_mode).  Results are the model's own return values.

    SURVEY 8(f-1) / twostep.py:44-100 feeding raft.py:101-200; measured by bench.py `pipeline_from_events`."""

    def __init__(self, model, assembler, iters: int = 12):
        self.model, self.asm, self.iters = model, assembler, iters
        dev = assembler.device
        self.side = torch.cuda.Stream(device=dev)
        shape = (2 * assembler.num_bins - 1, assembler.height, assembler.width)
        self.bufs = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.free = [None, None]          # event behind the last forward that read the buffer
        self.k = 0

    def __call__(self, events, forward_flow_timestamps, index: int, check: bool = False):
        slot = self.k & 1
        self.k += 1
        main = torch.cuda.current_stream()
        if self.k == 1:
            self.side.wait_stream(main)        # (first call: whatever produced the event arrays has been enqueued on `main`)
        with torch.cuda.stream(self.side):
            if self.free[slot] is not None:
                self.side.wait_event(self.free[slot])
            grid = self.asm.assemble(events, forward_flow_timestamps, index, check=check, out=self.bufs[slot])
            self.ready[slot].record(self.side)
        main.wait_event(self.ready[slot])
        out = self.model(voxel_grid=grid[None], iters=self.iters, test_mode=True)
        ev = torch.cuda.Event()
        ev.record(main)
        self.free[slot] = ev
        return out
