"""End-point error on the GPU -- drop-in for epe_masked / EPE of utils/metrics.py:30-49,196-213.

The per-pixel error and its masked mean are one HIP reduction (K15).  `EPE` keeps the torchmetrics state
(`epe`: float64 sum of per-batch means, `total`: int64 count, dist_reduce_fx="sum") and syncs it with ONE all-gather of
the 2-element record over RCCL (bflow_amd/dist.py) instead of depending on torchmetrics."""
from __future__ import annotations

from typing import Optional

import torch

from . import hip


def epe_masked(source: torch.Tensor, target: torch.Tensor, valid_mask: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """mean over (valid) pixels of ||source - target||_2 along dim 1; None if the mask is empty (metrics.py:196-213).
    Returns a 0-dim float32 GPU tensor.  The empty-mask test costs one device->host read, as in the reference (:209)."""
    assert source.ndim > 2 and source.shape == target.shape
    acc = torch.zeros(2, dtype=torch.float64, device=source.device)
    if valid_mask is not None:
        assert valid_mask.shape[0] == target.shape[0] and valid_mask.ndim == target.ndim - 1 and valid_mask.dtype == torch.bool
        assert tuple(valid_mask.shape[1:]) == tuple(source.shape[2:])
        valid_mask = valid_mask.contiguous()
    hip.epe_accumulate(source.float().contiguous(), target.float().contiguous(), valid_mask, acc)
    if valid_mask is not None and float(acc[1]) == 0:
        return None
    return (acc[0] / acc[1]).float()


class EPE:
    """metrics.py:30-49 without torchmetrics: update() adds one batch mean; compute() = sum / count."""

    def __init__(self, device=None):
        self.epe = torch.tensor(0, dtype=torch.float64, device=device)
        self.total = torch.tensor(0, dtype=torch.int64, device=device)

    def update(self, source, target, valid_mask=None):
        e = epe_masked(source, target, valid_mask)
        if e is not None:
            self.epe = self.epe.to(e.device) + e.double()
            self.total = self.total.to(e.device) + 1

    def state(self) -> torch.Tensor:
        """(2,) float64 record [epe_sum, count] -- what is exchanged between ranks."""
        return torch.stack([self.epe.double(), self.total.double()])

    def compute(self) -> torch.Tensor:
        assert int(self.total) > 0
        return (self.epe / self.total).float()
