"""Stage timers with the reference's hook names (models/raft_spline/raft.py:116-186, utils/timers.py:11-78).

The reference's CudaTimer synchronises the device and reads the wall clock around every stage; here each stage is
bracketed by hipEvents on torch's current stream, so the measurement does not serialise the pipeline.  Like the
reference's summary, the first `skip_warmup` samples of every stage are dropped."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List

import torch

STAGES = ("fnet_ev", "fnet_img", "cnet", "corr computation", "all iters", "1 iter", "get_flow (per iter)",
          "corr lookup (per iter)", "update (per iter)")


class StageTimer:
    def __init__(self, skip_warmup: int = 2):
        self.skip_warmup = skip_warmup
        self._open: Dict[str, List[torch.cuda.Event]] = defaultdict(list)
        self._pairs: Dict[str, List] = defaultdict(list)

    def start(self, name: str):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._open[name].append(ev)

    def stop(self, name: str):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._pairs[name].append((self._open[name].pop(), ev))

    def summary_ms(self, per_forward_counts: Dict[str, int] = None) -> Dict[str, float]:
        """Mean milliseconds per stage occurrence (after dropping warm-up samples)."""
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self._pairs.items():
            vals = [a.elapsed_time(b) for a, b in pairs]
            skip = self.skip_warmup * (per_forward_counts or {}).get(name, 1)
            vals = vals[skip:] if len(vals) > skip else vals
            out[name] = sum(vals) / max(len(vals), 1)
        return out

    def reset(self):
        self._open.clear()
        self._pairs.clear()


class StampTimer:
    """Stage times of a CAPTURED forward: `model._probe = StampTimer(device)` makes the forward enqueue one bflow_clock_stamp launch (the
    device's 100 MHz wall clock into a slot) at every stage boundary, also inside a hipGraph; `stage_ms()` reads the slots of the last
    replay.  Unlike hipEvents around eager launches (StageTimer) this times the replay `value` is measured on -- each stamp is a
    one-thread launch of its own (~1.5 us on the chain), the only perturbation."""

    def __init__(self, device, slots: int = 256):
        self.slots = torch.zeros(slots, dtype=torch.int64, device=device)
        self.names: List[str] = []

    def __call__(self, name: str):
        from . import hip
        if name not in self.names:
            self.names.append(name)
        hip.clock_stamp(self.slots, self.names.index(name))

    def read_us(self) -> Dict[str, float]:
        torch.cuda.synchronize()
        t = self.slots.cpu().tolist()
        t0 = min(t[i] for i in range(len(self.names)))
        return {n: (t[i] - t0) / 100.0 for i, n in enumerate(self.names)}

    def stage_ms(self) -> Dict[str, float]:
        """The reference's hook names (raft.py:116-186) from the stamps of the last replay; per-iteration stages = mean over iterations."""
        u = self.read_us()
        out = {}

        def span(name, a, b):
            if a in u and b in u:
                out[name] = (u[b] - u[a]) / 1e3
        span("cnet", "cnet.begin", "cnet.end")
        span("fnet_ev", "fnet.begin", "fnet.end")
        span("corr computation", "fnet.end", "corr.end")
        span("all iters", "joined", "iters.end")
        its = sorted(int(n[4:-6]) for n in u if n.startswith("iter") and n.endswith(".begin"))
        if len(its) >= 2:
            begins = [u[f"iter{k}.begin"] for k in its]
            out["1 iter"] = (begins[-1] - begins[0]) / (len(its) - 1) / 1e3
            look = [u[f"iter{k}.lookup_end"] - u[f"iter{k}.begin"] for k in its if f"iter{k}.lookup_end" in u]
            if look:
                out["corr lookup (per iter)"] = sum(look) / len(look) / 1e3
                out["update (per iter)"] = out["1 iter"] - out["corr lookup (per iter)"]
        out["forward"] = max(u.values()) / 1e3
        return out
