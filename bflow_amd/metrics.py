"""Flow metrics on the GPU -- drop-in for utils/metrics.py (EPE, EPE_MULTI, AE, AE_MULTI, NPE and their functional forms).

The per-pixel errors and their masked sums are HIP reductions (K15 `bflow_epe_accumulate`; `bflow_flow_metrics_accumulate` produces the
EPE / angular-error / n-pixel-error sums of one (prediction, ground truth, mask) triple in a single pass).  The metric classes keep the
torchmetrics state (float64 sum of per-batch values + int64 count, dist_reduce_fx="sum") and sync it with ONE all-gather of the
2-element record over RCCL (bflow_amd/dist.py) instead of depending on torchmetrics."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

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


# ----------------------------------------------------------------------------------------------------------------------
# the rest of utils/metrics.py
# ----------------------------------------------------------------------------------------------------------------------
def _check_triple(source, target, valid_mask):
    assert source.ndim > 2 and source.shape == target.shape
    if valid_mask is not None:
        assert valid_mask.shape[0] == target.shape[0] and valid_mask.ndim == target.ndim - 1 and valid_mask.dtype == torch.bool
        assert tuple(valid_mask.shape[1:]) == tuple(source.shape[2:])
        valid_mask = valid_mask.contiguous()
    return source.float().contiguous(), target.float().contiguous(), valid_mask


def flow_error_sums(source: torch.Tensor, target: torch.Tensor, valid_mask: Optional[torch.Tensor] = None,
                    n_pixels: Sequence[float] = (1.0, 2.0, 3.0)) -> torch.Tensor:
    """(6,) float64 GPU tensor [sum epe, #valid, sum angular error (rad), #n-pixel errors x3] of one batch, in ONE pass."""
    source, target, valid_mask = _check_triple(source, target, valid_mask)
    acc = torch.zeros(6, dtype=torch.float64, device=source.device)
    hip.flow_metrics_accumulate(source, target, valid_mask, n_pixels, acc)
    return acc


def ae_masked(source: torch.Tensor, target: torch.Tensor, valid_mask: Optional[torch.Tensor] = None, degrees: bool = True) -> torch.Tensor:
    """Mean angular error between the (u, v, 1) vectors over the (valid) pixels (metrics.py:259-296).  An empty mask gives NaN
    (0 / 0), as in the reference."""
    acc = flow_error_sums(source, target, valid_mask)
    ae = acc[2] / acc[1]
    if degrees:
        ae = ae / math.pi * 180
    return ae.float()


def n_pixel_error_masked(source: torch.Tensor, target: torch.Tensor, valid_mask: Optional[torch.Tensor], n_pixels: float) -> torch.Tensor:
    """Percentage of (valid) pixels whose error exceeds n_pixels AND 5 % of the ground-truth magnitude (metrics.py:160-193)."""
    acc = flow_error_sums(source, target, valid_mask, (n_pixels, n_pixels, n_pixels))
    if valid_mask is not None:
        assert float(acc[1]) > 0                                        # metrics.py:172
    return (acc[3] / acc[1] * 100).float()


def epe_masked_multi(source_lst: List[torch.Tensor], target_lst: List[torch.Tensor],
                     valid_mask_lst: Optional[List[torch.Tensor]] = None) -> Optional[torch.Tensor]:
    """metrics.py:216-239: mean of the per-prediction masked EPE over the predictions with a non-empty mask."""
    num_preds = len(source_lst)
    assert num_preds > 0
    assert len(target_lst) == num_preds, len(target_lst)
    if valid_mask_lst is not None:
        assert len(valid_mask_lst) == num_preds, len(valid_mask_lst)
    else:
        valid_mask_lst = [None] * num_preds
    epe_sum, den = 0, 0
    for src, tgt, vm in zip(source_lst, target_lst, valid_mask_lst):
        e = epe_masked(src, tgt, vm)
        if e is not None:
            epe_sum = epe_sum + e
            den += 1
    if den == 0:
        return None
    return epe_sum / den


def ae_masked_multi(source_lst: List[torch.Tensor], target_lst: List[torch.Tensor], valid_mask_lst: Optional[List[torch.Tensor]] = None,
                    degrees: bool = True) -> torch.Tensor:
    """metrics.py:241-256."""
    num_preds = len(source_lst)
    assert num_preds > 0
    assert len(target_lst) == num_preds, len(target_lst)
    if valid_mask_lst is not None:
        assert len(valid_mask_lst) == num_preds, len(valid_mask_lst)
    else:
        valid_mask_lst = [None] * num_preds
    total = 0
    for src, tgt, vm in zip(source_lst, target_lst, valid_mask_lst):
        total = total + ae_masked(src, tgt, vm, degrees)
    return total / num_preds


def predictions_from_lin_assumption(source: torch.Tensor, target_timestamps: List[float]) -> List[torch.Tensor]:
    """metrics.py:298-305: the final flow scaled linearly in time."""
    assert max(target_timestamps) <= 1
    assert 0 <= min(target_timestamps)
    return [ts * source for ts in target_timestamps]


class _MeanOfBatches:
    """State of every metric of utils/metrics.py: float64 sum of per-batch values + int64 number of batches."""

    def __init__(self, device=None):
        self.value = torch.tensor(0, dtype=torch.float64, device=device)
        self.total = torch.tensor(0, dtype=torch.int64, device=device)

    def _add(self, v: Optional[torch.Tensor]):
        if v is not None:
            self.value = self.value.to(v.device) + v.double()
            self.total = self.total.to(v.device) + 1

    def state(self) -> torch.Tensor:
        return torch.stack([self.value.double(), self.total.double()])

    def compute(self) -> torch.Tensor:
        assert int(self.total) > 0
        return (self.value / self.total).float()


class AE(_MeanOfBatches):
    """metrics.py:90-110."""

    def __init__(self, degrees: bool = True, device=None):
        super().__init__(device)
        self.degrees = degrees

    def update(self, source, target, valid_mask=None):
        self._add(ae_masked(source, target, valid_mask, degrees=self.degrees))


class NPE(_MeanOfBatches):
    """metrics.py:138-158."""

    def __init__(self, n_pixels: float, device=None):
        super().__init__(device)
        assert n_pixels > 0
        self.n_pixels = n_pixels

    def update(self, source, target, valid_mask=None):
        self._add(n_pixel_error_masked(source, target, valid_mask, self.n_pixels))


class EPE_MULTI(_MeanOfBatches):
    """metrics.py:51-88 incl. the trajectory-length filter."""

    def __init__(self, min_traj_len=None, max_traj_len=None, device=None):
        super().__init__(device)
        self.min_traj_len = min_traj_len
        self.max_traj_len = max_traj_len

    @staticmethod
    def compute_traj_len(target: List[torch.Tensor]) -> torch.Tensor:
        return hip.traj_len(torch.stack([t.float() for t in target], dim=0).contiguous())

    def update(self, source: List[torch.Tensor], target: List[torch.Tensor], valid_mask: Optional[List[torch.Tensor]] = None):
        if self.min_traj_len is not None or self.max_traj_len is not None:
            traj_len = self.compute_traj_len(target)
            valid_len = torch.ones(traj_len.shape, dtype=torch.bool, device=traj_len.device)
            if self.min_traj_len is not None:
                valid_len &= (traj_len >= self.min_traj_len)
            if self.max_traj_len is not None:
                valid_len &= (traj_len <= self.max_traj_len)
            if valid_mask is None:
                valid_mask = [valid_len.clone() for _ in range(len(target))]
            else:
                valid_mask = [valid_mask[idx] & valid_len for idx in range(len(target))]
        self._add(epe_masked_multi(source, target, valid_mask))


class AE_MULTI(_MeanOfBatches):
    """metrics.py:113-136."""

    def __init__(self, degrees: bool = True, device=None):
        super().__init__(device)
        self.degrees = degrees

    def update(self, source: List[torch.Tensor], target: List[torch.Tensor], valid_mask: Optional[List[torch.Tensor]] = None):
        self._add(ae_masked_multi(source, target, valid_mask, degrees=self.degrees))


class SingleFlowMetrics:
    """The `single_metrics` MetricCollection of modules/raft_spline.py:33-39 ('epe', 'ae', '1pe', '2pe', '3pe') fed by ONE
    reduction pass per batch instead of five."""

    KEYS = ("epe", "ae", "1pe", "2pe", "3pe")

    def __init__(self, prefix: str = "", device=None):
        self.prefix = prefix
        self.m = {k: _MeanOfBatches(device) for k in self.KEYS}

    def __call__(self, source, target, valid_mask=None):
        """update + the batch values, like MetricCollection.forward."""
        acc = flow_error_sums(source, target, valid_mask, (1.0, 2.0, 3.0))
        n = acc[1]
        empty = valid_mask is not None and float(n) == 0
        vals = {"epe": None if empty else (acc[0] / n).float(), "ae": (acc[2] / n / math.pi * 180).float()}
        if valid_mask is not None:
            assert not empty                                            # n_pixel_error_masked asserts num_valid > 0
        for i, k in enumerate(("1pe", "2pe", "3pe")):
            vals[k] = (acc[3 + i] / n * 100).float()
        for k, v in vals.items():
            self.m[k]._add(v)
        return {self.prefix + k: v for k, v in vals.items() if v is not None}

    def compute(self):
        return {self.prefix + k: m.compute() for k, m in self.m.items()}

    def state(self) -> torch.Tensor:
        """(5, 2) float64: what is exchanged between ranks (sum over ranks, then value / total)."""
        return torch.stack([self.m[k].state() for k in self.KEYS])
