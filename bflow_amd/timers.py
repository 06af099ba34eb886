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
