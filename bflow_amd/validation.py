"""Validation harness of the RAFT-spline path -- SURVEY 8(f-3).

Mirrors what the reference does around the network at evaluation time, without Lightning / torchmetrics:
  * `DataLoading`, `DataSetType`           -- the batch keys of data/utils/keys.py (same member names, so reference batches work);
  * `InputPadder`                          -- modules/utils.py:48-83 (replicate padding to a multiple of 8 and its inverse);
  * `Validator.validation_step(batch)`     -- modules/raft_spline.py:190-319: DSEC branch (single forward flow at tau = 1) and
                                              MultiFlow branch (M flows at the ground-truth timestamps, multi metrics and the
                                              linear-assumption baseline), same outputs and metric names ('val/epe', 'val/ae',
                                              'val/1pe', 'val/2pe', 'val/3pe', 'val/epe_multi', 'val/ae_multi', 'val/epe_multi_lin',
                                              'val/ae_multi_lin').
All arithmetic runs in HIP kernels (bflow_amd/metrics.py, hip.pad_replicate); this file is orchestration.
"""
from __future__ import annotations

from enum import Enum, IntEnum, auto
from typing import Any, Dict, List

import torch

from . import hip
from .metrics import AE_MULTI, EPE_MULTI, SingleFlowMetrics, predictions_from_lin_assumption


class DataSetType(IntEnum):        # data/utils/keys.py:3-5
    DSEC = auto()
    MULTIFLOW2D = auto()


class DataLoading(Enum):           # data/utils/keys.py:7-16
    FLOW = auto()
    FLOW_TIMESTAMPS = auto()
    FLOW_VALID = auto()
    FILE_INDEX = auto()
    EV_REPR = auto()
    BIN_META = auto()
    IMG = auto()
    IMG_TIMESTAMPS = auto()
    DATASET_TYPE = auto()


def _get(batch: Dict[Any, Any], key: DataLoading, default=None):
    """Batches keyed by this enum, by the reference's enum (same member names) or by the member name."""
    for k, v in batch.items():
        if k is key or getattr(k, "name", k) == key.name:
            return v
    return default


class InputPadder:
    """Pads the last two dimensions to multiples of `min_size` with replicated edges (modules/utils.py:48-83).

    Deviation, on purpose: the reference's `requires_padding` (:56-61) starts from `answer = False` and only ANDs into it, so it
    never pads and non-multiple-of-8 inputs trip the network's shape asserts; here it answers the question it asks."""

    def __init__(self, min_size: int = 8, no_top_padding: bool = False):
        assert min_size > 0
        self.min_size = min_size
        self.no_top_padding = no_top_padding
        self._pad = None

    def requires_padding(self, input_tensor: torch.Tensor) -> bool:
        ht, wd = input_tensor.shape[-2:]
        return not (ht % self.min_size == 0 and wd % self.min_size == 0)

    def pad(self, input_tensor: torch.Tensor) -> torch.Tensor:
        ht, wd = input_tensor.shape[-2:]
        pad_ht = (((ht // self.min_size) + 1) * self.min_size - ht) % self.min_size
        pad_wd = (((wd // self.min_size) + 1) * self.min_size - wd) % self.min_size
        if self.no_top_padding:
            pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]                      # RAFT: KITTI
        else:
            pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]   # RAFT: Sintel (default)
        if self._pad is None:
            self._pad = pad
        else:
            assert self._pad == pad
        return hip.pad_replicate(input_tensor.float().contiguous(), self._pad)

    def unpad(self, input_tensor: torch.Tensor) -> torch.Tensor:
        ht, wd = input_tensor.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return input_tensor[..., c[0]:c[1], c[2]:c[3]]


def reduce_ev_repr(ev_repr: torch.Tensor) -> torch.Tensor:
    """modules/utils.py:36-45: the time-summed event representation kept for visualisation."""
    assert isinstance(ev_repr, torch.Tensor) and ev_repr.ndim == 4 and ev_repr.is_cuda
    return torch.sum(ev_repr, dim=1)


class Validator:
    """`RAFTSplineModule`'s evaluation half (modules/raft_spline.py:17-58,190-319) around a bflow_amd.RAFTSpline."""

    def __init__(self, net, config_model: Dict[str, Any], device=None):
        self.net = net
        self.num_iter_test = config_model["num_iter"]["test"]
        self.use_images = config_model["use_boundary_images"]
        self.use_events = config_model["use_events"]
        self._input_padder = InputPadder(min_size=8, no_top_padding=False)
        self.val_single_metrics = SingleFlowMetrics(prefix="val/", device=device)
        self.val_epe_multi, self.val_ae_multi = EPE_MULTI(device=device), AE_MULTI(degrees=True, device=device)
        self.val_epe_multi_lin, self.val_ae_multi_lin = EPE_MULTI(device=device), AE_MULTI(degrees=True, device=device)
        self.logged: List[Dict[str, torch.Tensor]] = []

    def forward(self, voxel_grid, images, iters, test_mode: bool):
        out = self.net(voxel_grid=voxel_grid, images=images, iters=iters, test_mode=test_mode)
        if getattr(self.net, "_graph_mode", None) == "on":
            # under explicit hipGraph replay (enable_hipgraph(); the default "auto" mode returns private copies itself) the curves wrap the graph's static output buffers, which the NEXT replay overwrites; the reference
            # hands out detached copies (@to_cpu, modules/raft_spline.py:190), so what a validation step returns must survive later steps
            out = tuple(c.detach(clone=True) for c in out) if isinstance(out, tuple) else [c.detach(clone=True) for c in out]
        return out

    def validation_step(self, batch: Dict[Any, Any], batch_idx: int = 0) -> Dict[str, Any]:
        flow_gt = _get(batch, DataLoading.FLOW)
        flow_gt_valid = _get(batch, DataLoading.FLOW_VALID)
        ev_repr = _get(batch, DataLoading.EV_REPR)
        images = _get(batch, DataLoading.IMG) if self.use_images else None
        dataset_type = _get(batch, DataLoading.DATASET_TYPE)[0]
        output: Dict[str, Any] = {}
        log: Dict[str, torch.Tensor] = {}

        if int(dataset_type) == int(DataSetType.DSEC):
            combined_bins = ev_repr.shape[1]
            assert combined_bins % 2 == 1, f"combined_bins={combined_bins}"
            num_bins = combined_bins // 2 + 1
            assert num_bins >= 1
            ev_repr_previous, ev_repr_current = ev_repr[:, 0:num_bins, ...], ev_repr[:, -num_bins:, ...]
            requires_padding = self._input_padder.requires_padding(ev_repr)
            if requires_padding:
                ev_repr = self._input_padder.pad(ev_repr)
                if images is not None:
                    assert len(images) == 2
                    images = [self._input_padder.pad(x) for x in images]
            _, bezier_up = self.forward(ev_repr if self.use_events else None, images, self.num_iter_test, True)
            flow_pred = bezier_up.get_flow_from_reference(1.0)
            if requires_padding:
                flow_pred = self._input_padder.unpad(flow_pred)
                if images is not None:
                    images = [self._input_padder.unpad(x) for x in images]
            log.update(self.val_single_metrics(flow_pred, flow_gt, flow_gt_valid))
            output.update({"pred": flow_pred, "gt": flow_gt, "gt_valid": flow_gt_valid})
        elif int(dataset_type) == int(DataSetType.MULTIFLOW2D):
            meta = _get(batch, DataLoading.BIN_META)
            nbins_context, nbins_corr = int(meta["nbins_context"][0]), int(meta["nbins_correlation"][0])
            nbins_total = ev_repr.shape[1]
            assert nbins_total == int(meta["nbins_total"][0]) == nbins_context + nbins_corr - 1
            ev_repr_previous, ev_repr_current = ev_repr[:, 0:nbins_corr, ...], ev_repr[:, -nbins_corr:, ...]
            flow_ts = _get(batch, DataLoading.FLOW_TIMESTAMPS)
            requires_padding = self._input_padder.requires_padding(ev_repr)
            if requires_padding:
                ev_repr = self._input_padder.pad(ev_repr)
                ev_repr_previous = self._input_padder.pad(ev_repr_previous)
                ev_repr_current = self._input_padder.pad(ev_repr_current)
                if images is not None:
                    images = [self._input_padder.pad(x) for x in images]
            _, bezier_up = self.forward(ev_repr if self.use_events else None, images, self.num_iter_test, True)
            flow_preds, timestamp_eval_lst = [], []
            for timestamp_batch in flow_ts:
                ts_mean_diff = (timestamp_batch[1:] - timestamp_batch[:-1]).abs().mean().item() if len(timestamp_batch) > 1 else 0.0
                assert 0 <= ts_mean_diff < 0.001, ts_mean_diff          # raft_spline.py:265-267
                timestamp = float(timestamp_batch[0])
                timestamp_eval_lst.append(timestamp)
                pred = bezier_up.get_flow_from_reference(timestamp)
                if requires_padding:
                    pred = self._input_padder.unpad(pred)
                flow_preds.append(pred)
            if requires_padding and images is not None:
                images = [self._input_padder.unpad(x) for x in images]
            log.update(self.val_single_metrics(flow_preds[-1], flow_gt[-1]))
            self.val_epe_multi.update(flow_preds, flow_gt)
            self.val_ae_multi.update(flow_preds, flow_gt)
            lin = predictions_from_lin_assumption(flow_preds[-1], timestamp_eval_lst)
            self.val_epe_multi_lin.update(lin, flow_gt)
            self.val_ae_multi_lin.update(lin, flow_gt)
            output.update({"pred": flow_preds[-1], "gt": flow_gt})
        else:
            raise NotImplementedError
        output["bezier_prediction"] = bezier_up
        if self.use_events:
            output["ev_repr_reduced"] = reduce_ev_repr(ev_repr_current)
            output["ev_repr_reduced_m1"] = reduce_ev_repr(ev_repr_previous)
        if images is not None:
            output["images"] = images
        self.logged.append(log)
        return output

    def compute(self) -> Dict[str, torch.Tensor]:
        """Epoch values of everything that received at least one batch."""
        out = {k: v for k, v in self.val_single_metrics.compute().items()} if int(self.val_single_metrics.m["ae"].total) > 0 else {}
        for name, m in (("val/epe_multi", self.val_epe_multi), ("val/ae_multi", self.val_ae_multi),
                        ("val/epe_multi_lin", self.val_epe_multi_lin), ("val/ae_multi_lin", self.val_ae_multi_lin)):
            if int(m.total) > 0:
                out[name] = m.compute()
        return out

    def state(self) -> torch.Tensor:
        """(9, 2) float64 [value sum, batch count] rows: the record one all-gather exchanges between ranks (bflow_amd/dist.py)."""
        rows = [self.val_single_metrics.m[k].state() for k in SingleFlowMetrics.KEYS]
        rows += [m.state() for m in (self.val_epe_multi, self.val_ae_multi, self.val_epe_multi_lin, self.val_ae_multi_lin)]
        dev = next((r.device for r in rows if r.is_cuda), rows[0].device)      # metrics that never saw a batch still live on the host
        return torch.stack([r.to(dev) for r in rows])
