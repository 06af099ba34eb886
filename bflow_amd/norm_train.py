"""InstanceNorm2d / BatchNorm2d (+ ReLU) forward AND backward on hand-written kernels -- the norm layers of the training path.

The reference trains `torch.nn.InstanceNorm2d` (feature encoders) and `torch.nn.BatchNorm2d` (context encoder) under autograd, each followed
by ReLU except on the down-sampling shortcut (models/raft_utils/extractor.py:5-55,58-125).  `norm_act(module, x, relu)` routes GPU tensors in
training mode through `_NormActFn` (csrc/norm_train.hip: statistics, normalise + activate, and the two-pass backward; BatchNorm's running
statistics are updated by the forward's finalise kernel exactly as torch does: momentum, unbiased variance, `num_batches_tracked`).
Everything else -- CPU tensors (the oracle tests), eval mode, GroupNorm, planes that are not 16-byte aligned -- falls through to the
module itself.  Only x and the per-plane (mean, rstd, scale, shift) are kept for the backward pass; the ReLU mask is recomputed."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hip
from . import split as S

ENABLED = True                      # tools / A-B: False = torch's own norm layers (MIOpen BatchNorm) + a separate ReLU


class _NormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, module, mode: int, relu: bool):
        x = x.float().contiguous()
        B, C, H, W = x.shape
        HW = H * W
        dev = x.device
        stats = S.plane_stats(x)                                         # (B, C, 2) fp64
        coef = torch.empty((4, B, C), dtype=torch.float32, device=dev)   # mean, rstd, scale, shift per plane
        rm = rv = None
        momentum = 0.0
        if mode == 1 and module.track_running_stats and module.running_mean is not None:
            rm, rv = module.running_mean, module.running_var
            momentum = 0.1 if module.momentum is None else float(module.momentum)
        L = hip.lib()
        hip._check(L.bflow_norm_train_finalize(stats.data_ptr(), mode, B, C, HW, float(module.eps), None if gamma is None else hip._dev(gamma.detach(), name="gamma"),
                                               None if beta is None else hip._dev(beta.detach(), name="beta"), None if rm is None else hip._dev(rm, name="running_mean"),
                                               None if rv is None else hip._dev(rv, name="running_var"), momentum, coef[0].data_ptr(), coef[1].data_ptr(),
                                               coef[2].data_ptr(), coef[3].data_ptr(), hip._stream()), "bflow_norm_train_finalize")
        if rm is not None and module.num_batches_tracked is not None:
            module.num_batches_tracked.add_(1)
        y = torch.empty_like(x)
        hip._check(L.bflow_norm_train_apply(x.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), y.data_ptr(), B * C, HW, int(relu), hip._stream()),
                   "bflow_norm_train_apply")
        ctx.mode, ctx.relu, ctx.affine = mode, bool(relu), gamma is not None
        ctx.save_for_backward(x, coef)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, coef = ctx.saved_tensors
        B, C, H, W = x.shape
        HW = H * W
        dy = dy.float().contiguous()
        dev = x.device
        L = hip.lib()
        sums = torch.empty((B, C, 2), dtype=torch.float64, device=dev)
        m, r, sc, sh = (coef[k].data_ptr() for k in range(4))
        hip._check(L.bflow_norm_train_bwd_stats(dy.data_ptr(), x.data_ptr(), m, r, sc, sh, sums.data_ptr(), B * C, HW, int(ctx.relu), hip._stream()),
                   "bflow_norm_train_bwd_stats")
        k = torch.empty((2, B, C), dtype=torch.float32, device=dev)
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev) if ctx.affine else None
        hip._check(L.bflow_norm_train_bwd_finalize(sums.data_ptr(), ctx.mode, B, C, HW, k[0].data_ptr(), k[1].data_ptr(),
                                                   None if dgb is None else dgb[0].data_ptr(), None if dgb is None else dgb[1].data_ptr(), hip._stream()),
                   "bflow_norm_train_bwd_finalize")
        dx = torch.empty_like(x)
        hip._check(L.bflow_norm_train_bwd_apply(dy.data_ptr(), x.data_ptr(), m, r, sc, sh, k[0].data_ptr(), k[1].data_ptr(), dx.data_ptr(), B * C, HW,
                                                int(ctx.relu), hip._stream()), "bflow_norm_train_bwd_apply")
        return dx, (dgb[0] if ctx.affine else None), (dgb[1] if ctx.affine else None), None, None, None


def _supported(module: nn.Module, x: torch.Tensor):
    if not (ENABLED and x.is_cuda and x.dim() == 4 and torch.is_grad_enabled() and module.training):
        return None
    if (x.shape[2] * x.shape[3]) % 4 or x.shape[0] * x.shape[1] > 65535:      # bflow_plane_stats wants 16-B aligned planes
        return None
    if isinstance(module, nn.InstanceNorm2d) and not module.affine and not module.track_running_stats:
        return 0
    if isinstance(module, nn.BatchNorm2d) and module.affine and module.momentum is not None:
        return 1      # (momentum = None means a cumulative average with factor 1 / num_batches_tracked: not implemented here, refused below)
    return None


def norm_act(module: nn.Module, x: torch.Tensor, relu: bool) -> torch.Tensor:
    """[relu](module(x)) for the norm layers of the encoders: on the HIP kernels in GPU training mode, otherwise the module itself."""
    mode = _supported(module, x)
    if mode is None:
        if ENABLED and x.is_cuda and not isinstance(module, nn.Sequential):     # ("none" norm = empty Sequential: nothing to compute)
            raise hip.BflowHipError(f"norm_act({type(module).__name__}) on a GPU tensor outside the HIP training kernels (grad disabled, eval mode, "
                                    "or an unsupported norm layer): there is no library fall-back -- run inference through RAFTSpline.forward "
                                    "in eval() mode, or train with grad enabled")
        y = module(x)
        return torch.relu_(y) if relu else y
    affine = mode == 1
    return _NormActFn.apply(x, module.weight if affine else None, module.bias if affine else None, module, mode, relu)
