"""BezierCurves -- drop-in for models/raft_spline/bezier.py:15-216 (state container + curve evaluation + x8 convex
up-sampling).  Same public surface: get_params, get_flow_from_reference(float | list | ndarray), delta_update_params,
create_upsampled, detach / detach_ / cpu / cpu_, from_2view, create_from_specification / create_from_voxel_grid and the
batch_size / degree / dim / height / width / device / dtype / requires_grad properties.

GPU-resident curves evaluate through the HIP kernels (bflow_bezier_eval, bflow_cvx_upsample).  A curve that a caller
moved to the host with .cpu() (the reference's logging callbacks do: callbacks/logger.py:209,270 after
utils/general.py:66-75) is a plain data container; evaluating it there uses ordinary torch host ops -- that is
visualisation plumbing, not part of the accelerated path (RAFTSpline.forward never takes it).
"""
from __future__ import annotations

import math
from typing import List, Union

import numpy as np
import torch

from . import hip


def polynomial_coefficients(times: Union[List[float], np.ndarray], degree: int) -> np.ndarray:
    """(T, degree) fp32: binom(deg,i) (1-t)^(deg-i) t^i, computed in fp64 then cast (bezier.py:141-180)."""
    ts = np.asarray(times, dtype="float64").reshape(-1)
    assert ts.size > 0 and ts.min() >= 0 and ts.max() <= 1
    return hip.bezier_coeffs(ts, degree)


class BezierCurves:
    CTRL_DIM: int = 2  # each control point lives in R^2; P0 == pixel location is implicit (bezier.py:28-31)

    def __init__(self, bezier_params: torch.Tensor):
        assert bezier_params.ndim == 4
        self._params = bezier_params
        self.batch, channels, self.ht, self.wd = bezier_params.shape
        assert channels % 2 == 0
        self.n_ctrl_pts = channels // self.CTRL_DIM + 1
        assert self.n_ctrl_pts > 0

    @staticmethod
    def comb(n: int, k: int) -> int:
        return math.comb(n, k)

    # -- constructors (bezier.py:46-71) --------------------------------------------------------------
    @classmethod
    def create_from_specification(cls, batch_size: int, n_ctrl_pts: int, height: int, width: int, device) -> "BezierCurves":
        assert batch_size > 0 and n_ctrl_pts > 1 and height > 0 and width > 0
        return cls(torch.zeros(batch_size, cls.CTRL_DIM * (n_ctrl_pts - 1), height, width, device=device))

    @classmethod
    def from_2view(cls, flow_tensor: torch.Tensor) -> "BezierCurves":
        assert flow_tensor.shape[1] == 2 == cls.CTRL_DIM
        return cls(flow_tensor)

    @classmethod
    def create_from_voxel_grid(cls, voxel_grid: torch.Tensor, downsample_factor: int = 8, bezier_degree: int = 2) -> "BezierCurves":
        assert isinstance(downsample_factor, int) and downsample_factor >= 1
        batch, _, ht, wd = voxel_grid.shape
        assert ht % 8 == 0 and wd % 8 == 0
        return cls.create_from_specification(batch, bezier_degree + 1, ht // downsample_factor, wd // downsample_factor,
                                             voxel_grid.device)

    # -- container API ------------------------------------------------------------------------------
    @property
    def device(self):
        return self._params.device

    @property
    def dtype(self):
        return self._params.dtype

    @property
    def requires_grad(self):
        return self._params.requires_grad

    @property
    def batch_size(self):
        return self._params.shape[0]

    @property
    def degree(self):
        return self.n_ctrl_pts - 1

    @property
    def dim(self):
        return self._params.shape[1]

    @property
    def height(self):
        return self._params.shape[-2]

    @property
    def width(self):
        return self._params.shape[-1]

    def get_params(self) -> torch.Tensor:
        return self._params

    def _param_view(self) -> torch.Tensor:
        return self._params.view(self.batch, self.CTRL_DIM, self.degree, self.ht, self.wd)

    def detach(self, clone: bool = False, cpu: bool = False) -> "BezierCurves":
        p = self._params.detach()
        if cpu:
            return BezierCurves(p.cpu())
        return BezierCurves(p.clone() if clone else p)

    def detach_(self, cpu: bool = False) -> None:
        self._params = self._params.detach()
        if cpu:
            self._params = self._params.cpu()

    def cpu(self) -> "BezierCurves":
        return BezierCurves(self._params.cpu())

    def cpu_(self) -> None:
        self._params = self._params.cpu()

    def delta_update_params(self, delta_bezier: torch.Tensor) -> None:
        assert delta_bezier.shape == self._params.shape
        self._params = self._params + delta_bezier

    # -- evaluation (bezier.py:165-216) -----------------------------------------------------------------
    def create_upsampled(self, mask: torch.Tensor) -> "BezierCurves":
        """[N, dim, H/8, W/8] -> [N, dim, H, W] by convex combination (bezier.py:81-84, raft_utils/utils.py:33-48)."""
        if torch.is_grad_enabled() and (self._params.requires_grad or mask.requires_grad):
            from .training import cvx_upsample                # differentiable K13 (csrc/backward.hip)
            return BezierCurves(cvx_upsample(self._params, mask))
        return BezierCurves(hip.cvx_upsample(self._params.contiguous(), mask.contiguous()))

    def get_flow_from_reference(self, time: Union[float, int, List[float], np.ndarray]) -> torch.Tensor:
        pv = self._param_view()
        batch, dim, degree, height, width = pv.shape
        scalar = isinstance(time, (int, float))
        if scalar:
            assert 0.0 <= time <= 1.0
            if time == 1:
                return pv[:, :, -1, ...]
            if time == 0:
                return torch.zeros((batch, dim, height, width), dtype=self.dtype, device=self.device)
            time = np.array([time], dtype="float64")
        elif isinstance(time, list):
            time = np.asarray(time, dtype="float64")
        else:
            assert isinstance(time, np.ndarray)
        assert time.dtype == "float64" and time.size > 0 and time.min() >= 0 and time.max() <= 1
        coef = polynomial_coefficients(time, degree)
        if self._params.is_cuda and not (torch.is_grad_enabled() and self._params.requires_grad):
            flows = hip.bezier_eval(self._params.contiguous().float(), coef, add_coords0=False)
        else:  # host-resident container (see module docstring)
            cf = (hip.const_tensor(coef, pv.device) if pv.is_cuda else torch.from_numpy(coef)).to(pv.dtype)
            # sum_p pv[b, d, p] * cf[t, p] as a broadcast product (identical to the reference's einsum up to summation order; no library GEMM)
            flows = (pv.unsqueeze(0) * cf.view(cf.shape[0], 1, 1, degree, 1, 1)).sum(dim=3)
        return flows[0] if scalar else flows
