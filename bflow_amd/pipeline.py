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


class EventFrameGraph:
    """Raw events -> flow for the frames of ONE resident recording, one hipGraph replay per frame.  K1 takes its window {first event, count,
    centres} from a device descriptor the host rewrites before every replay (bflow_voxel_grid_rectified_window); its launches and workspace
    are planned for `max_events` per window.  Two forms (profiles/r06_pipeline_graph.txt):

    * serial (default): replay = 2 x K1 + merge + K2 of frame k, then frame k's forward, no frame of latency:
            out = g(forward_flow_timestamps, index)
      (Measured equal to eager assembly + forward replay, +-0.02 ms: in steady state the host enqueues the eager launches while the previous
      replay runs, so there are no gaps to remove -- what this form adds is the next point, inside one launch per frame.)
      Consecutive frames of a 100-ms-step stream share a window (frame k + 1's previous window IS frame k's current one, twostep.py:63-64):
      its 5-bin grid is kept and the replay runs ONE K1 (`reuse_windows`, decided on the host by comparing the window descriptors; the
      reference caches the same per-window grids on disk, base.py:93-104).  Any other frame order falls back to both windows.
    * `overlap=True`: the assembly of frame k + 1 as a BRANCH of the graph that runs frame k's forward, on the forward's own side stream
      behind the context encoder (idle during the batch-1 GRU loop; the graph keeps two parallel chains -- more are not safe on this
      runtime: ConcurrentRunner); two graphs alternate (forward on grid[p], assembly into grid[1 - p]):
            out = g.submit(forward_flow_timestamps, index)     # curves of the PREVIOUS submission (None at first);  last = g.flush()
      Built, bit-identical, and SLOWER than the serial form (258 vs 274 frames/s for eager assembly + replay on one box): K1's launches are
      sized to fill the chip (512 workgroups of 16 waves = every wave slot), so next to them each of the GRU loop's ~120 dependent
      launches waits for slots -- the loop stretches by more than the assembly takes alone.  Kept for the measurement.
      (`EventFramePipeline` above, eager assembly on a second stream, gains nothing either: eager launches do not run next to a replay.)

    Bit-identical to `model(voxel_grid=assembler.assemble(...)[None])`: K1's fixed-point accumulation does not depend on the chunking, the
    forward is the same launches (tests/test_hip_parity.py::test_event_frame_graph_matches_eager).  The reference's per-sample asserts
    (coordinates inside the map, agreement of the shared slice: base.py:141, twostep.py:83) are host reads and are not part of a replay;
    `bad_events()` returns the out-of-map counter of the last assembly.

    SURVEY 8(f-1) / twostep.py:44-100 feeding raft.py:101-200; measured by bench.py `pipeline_from_events`."""

    def __init__(self, model, assembler, events, iters: int = 12, max_events: Optional[int] = None, overlap: bool = False,
                 reuse_windows: bool = True):
        assert assembler.merge_grids and assembler.normalize and assembler.version == 1 and assembler.voxel_grid_dir is None, \
            "EventFrameGraph: merged, normalised, extended voxel grids built from the events (the DSEC two-step default)"
        self.model, self.asm, self.events, self.iters, self.overlap = model, assembler, events, iters, bool(overlap)
        dev = torch.device(assembler.device)
        self.device = dev
        C, H, W = assembler.num_bins, assembler.height, assembler.width
        self.max_events = int(max_events if max_events is not None else min(events.t_host.size, 1 << 22))
        with torch.inference_mode(False):      # static buffers: rewritten in place by every later call, whatever mode that call runs in (graph.py)
            self.ws = hip.voxel_workspace(self.max_events, C, H, W, True, dev)
            self.norm_ws = hip.voxel_norm_workspace(dev)
            self.win = torch.zeros((2, 4), dtype=torch.int64, device=dev)          # [current, previous] window of the frame being assembled
            self._pin = [torch.zeros((2, 4), dtype=torch.int64).pin_memory() for _ in range(4)]
            self.parts = [torch.empty((C, H, W), dtype=torch.float32, device=dev) for _ in range(2)]
            self.grid = [torch.empty((2 * C - 1, H, W), dtype=torch.float32, device=dev) for _ in range(2)]
        self._pin_ev = [None] * 4
        self._graphs = [None, None]
        self._outs = [None, None]
        self.reuse_windows = bool(reuse_windows)
        self._serial = {}           # serial form: previous window reused? -> (graph, static outputs)
        self._last_cur = None       # descriptor of the window whose grid parts[1] holds (the last frame's current window)
        self._rows = None
        self.k1_launch_sets = 0     # K1 launch sequences replayed so far (2 per frame, 1 where the previous window was reused)
        self._p = 0                 # grid[_p] holds the frame whose forward runs next
        self._pending = False
        self._n = 0
        from .graph import WeightsWatch
        self._weights = WeightsWatch(model)

    # -- the assembly as capturable launches: windows from self.win, everything else static
    def _assemble(self, out: torch.Tensor, reuse_prev: bool = False, keep: bool = False):
        """The current window (descriptor row 0) -> parts[0]; the previous one (row 1) -> parts[1], unless `reuse_prev`: then parts[1] already
        holds it (it was the current window of the frame before: twostep.py:63-64, 100-ms steps).  `keep`: the current window's grid is copied
        to parts[1] behind the merge, for the next frame (6 MB device to device: ONE captured variant serves every consecutive frame --
        alternating two executables with swapped slots instead measured slower than assembling both windows)."""
        a, ev = self.asm, self.events
        a._bad.zero_()
        cur, prev = self.parts[0], self.parts[1]
        hip.voxel_grid_rectified_window(ev.x, ev.y, ev.p, ev.t, self.win[0], self.max_events, a.rectify_events_map, cur, self.ws, a._bad)
        if not reuse_prev:
            hip.voxel_grid_rectified_window(ev.x, ev.y, ev.p, ev.t, self.win[1], self.max_events, a.rectify_events_map, prev, self.ws, a._bad)
        hip.voxel_merge_norm(prev, cur[1:], out, workspace=self.norm_ws)      # cat((previous, current[1:])), twostep.py:77-85
        if keep:
            prev.copy_(cur)

    def _write_windows(self, forward_flow_timestamps, index: int):
        from .dsec import twostep_windows
        (cf, ct), (pf, pt) = twostep_windows(forward_flow_timestamps, index)
        rows = [self.asm.window_descriptor(self.events, cf, ct), self.asm.window_descriptor(self.events, pf, pt)]
        self._rows = rows
        for r in rows:
            assert r[1] <= self.max_events, f"window of {r[1]} events: EventFrameGraph was planned for {self.max_events} (max_events)"
        slot = self._n & 3
        self._n += 1
        if self._pin_ev[slot] is not None:
            self._pin_ev[slot].synchronize()           # the copy out of this staging buffer (four submissions ago) has run
        self._pin[slot].copy_(torch.tensor(rows, dtype=torch.int64))
        with torch.inference_mode(False):
            self.win.copy_(self._pin[slot], non_blocking=True)
        e = torch.cuda.Event()
        e.record()
        self._pin_ev[slot] = e

    def _capture(self, p: int, reuse_prev: bool = False, keep: bool = False):
        m = self.model
        assert not m.training, "inference only"
        with torch.inference_mode(False), torch.no_grad():
            m._forward_impl(self.grid[p][None], None, self.iters, None, True)          # warm-up outside the capture (lazy packing, allocator growth)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            gc.collect()
            gc_on = gc.isenabled()
            gc.disable()                  # (graph.py: a collection inside a capture can abort the process)
            try:
                with torch.cuda.graph(g):
                    if self.overlap:
                        fr = m._encode(self.grid[p][None], None, None)
                        with hip.Branch(True) as br:      # the forward's own side stream: idle from here to the last iteration's mask head
                            self._assemble(self.grid[1 - p])
                        low, ups = m._iterate(fr, self.iters, True)
                        br.join()
                    else:
                        self._assemble(self.grid[p], reuse_prev, keep)
                        low, ups = m._forward_impl(self.grid[p][None], None, self.iters, None, True)
            finally:
                if gc_on:
                    gc.enable()
        return g, (low, ups[-1])

    def __call__(self, forward_flow_timestamps, index: int):
        """Serial form (`overlap=False`, the default): ONE replay = assembly of frame `index` + its forward; returns its curves."""
        assert not self.overlap, "overlap=True: use submit() / flush()"
        with torch.cuda.device(self.device), torch.no_grad():
            if self._weights.changed():
                self.close()
            self._write_windows(forward_flow_timestamps, index)
            # The previous window of this frame is the current window of the frame before (a 100-ms-step stream, twostep.py:63-64): its 5-bin
            # grid is still in parts[last slot] -- ONE K1 per frame instead of two.  (The reference caches exactly these per-window grids,
            # on disk: base.py:93-104.)  Two captured variants: both windows, or the current one only.
            reuse = self.reuse_windows and self._last_cur is not None and self._rows[1] == self._last_cur
            if not self._serial:
                with torch.inference_mode(False):
                    self._assemble(self.grid[0])       # a real grid under the capture's warm-up forward
                reuse = False                          # (parts[1] now holds THIS frame's previous window, not a kept one)
            if reuse not in self._serial:
                self._serial[reuse] = self._capture(0, reuse, self.reuse_windows)
            g, (low, up) = self._serial[reuse]
            g.replay()
            self._last_cur = self._rows[0] if self.reuse_windows else None
            self.k1_launch_sets += 1 if reuse else 2
            return BezierCurves(low.clone()), BezierCurves(up.clone())

    def submit(self, forward_flow_timestamps, index: int):
        """Overlapped form (`overlap=True`): queues frame `index` (its two windows are assembled next to the GRU loop of the previously
        submitted frame) and returns that previous frame's (low-resolution curves, full-resolution curves), or None for the first submission."""
        assert self.overlap, "overlap=False: call the object"
        with torch.cuda.device(self.device), torch.no_grad():
            if self._weights.changed():
                self.close()
            self._write_windows(forward_flow_timestamps, index)
            if not self._pending:
                with torch.inference_mode(False):
                    self._assemble(self.grid[self._p])
                self._pending = True
                return None
            p = self._p
            if self._graphs[p] is None:
                # the windows just written belong to the NEXT frame; the capture's warm-up forward reads grid[p] only
                self._graphs[p], self._outs[p] = self._capture(p)
            self._graphs[p].replay()
            low, up = self._outs[p]
            self._p = 1 - p
            return BezierCurves(low.clone()), BezierCurves(up.clone())

    def flush(self):
        """The curves of the last submitted frame (a plain forward on its grid: nothing is left to assemble)."""
        if not self._pending:
            return None
        self._pending = False
        with torch.cuda.device(self.device):
            return self.model(voxel_grid=self.grid[self._p][None], iters=self.iters, test_mode=True)

    def bad_events(self) -> int:
        """Events of the last assembled frame whose raw coordinates lie outside the rectification map (one device -> host read)."""
        return int(self.asm._bad)

    def close(self):
        if any(g is not None for g in self._graphs) or self._serial:
            torch.cuda.synchronize(self.device)        # (ConcurrentRunner.close: a graph is destroyed only after its last replay has finished)
        self._graphs, self._outs = [None, None], [None, None]
        self._serial, self._last_cur = {}, None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
