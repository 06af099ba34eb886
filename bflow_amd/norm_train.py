"""InstanceNorm2d / BatchNorm2d (+ ReLU) forward AND backward on hand-written kernels -- the norm layers of the training path.

The reference trains `torch.nn.InstanceNorm2d` (feature encoders) and `torch.nn.BatchNorm2d` (context encoder) under autograd, each followed
by ReLU except on the down-sampling shortcut (models/raft_utils/extractor.py:5-55,58-125).  `norm_act(module, x, relu)` routes GPU tensors in
training mode through `_NormActFn` (csrc/norm_train.hip: statistics, normalise + activate, and the two-pass backward; BatchNorm's running
statistics are updated by the forward's finalise kernel exactly as torch does: momentum, unbiased variance, `num_batches_tracked`).
CPU tensors (the oracle tests) run the module itself; GPU cases outside the kernels (eval-mode BatchNorm after freeze_bn(), planes that are
not 16-byte aligned, ...) run as torch element-wise arithmetic (`_norm_elementwise`), never on the vendor library; GroupNorm is refused.  Only x and the per-plane (mean, rstd, scale, shift) are kept for the backward pass; the ReLU mask is recomputed."""
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


def _norm_elementwise(module: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """InstanceNorm2d / BatchNorm2d written with torch reductions + element-wise ops (autograd differentiates them; no MIOpen kernel is
    involved -- `F.batch_norm` / `F.instance_norm` on ROCm would be).  The cases `_NormActFn` does not take on a GPU tensor:
      * BatchNorm2d in eval() mode -- the reference API `RAFTSpline.freeze_bn()` (raft.py:75-78) followed by a training forward: a
        per-channel affine from the running statistics;
      * planes that are not 16-byte aligned (H*W % 4), B*C > 65535, BatchNorm with momentum=None (cumulative average), grad disabled.
    Same arithmetic as torch's modules: biased variance for the normalisation, unbiased for the running update."""
    xf = x.float()
    if isinstance(module, nn.BatchNorm2d):
        use_batch = module.training or module.running_mean is None
        if use_batch:
            mean = xf.mean(dim=(0, 2, 3))
            var = xf.var(dim=(0, 2, 3), unbiased=False)
            if module.training and module.track_running_stats and module.running_mean is not None:
                with torch.no_grad():
                    n = xf.numel() // xf.shape[1]
                    if module.num_batches_tracked is not None:
                        module.num_batches_tracked.add_(1)
                    if module.momentum is None:      # cumulative average: the factor stays on the device (no host read: capturable)
                        f = 1.0 / module.num_batches_tracked.to(mean.dtype)
                        module.running_mean.add_((mean.detach() - module.running_mean) * f)
                        module.running_var.add_((var.detach() * (n / max(n - 1, 1)) - module.running_var) * f)
                    else:
                        f = float(module.momentum)
                        module.running_mean.mul_(1 - f).add_(mean.detach(), alpha=f)
                        module.running_var.mul_(1 - f).add_(var.detach() * (n / max(n - 1, 1)), alpha=f)
        else:
            mean, var = module.running_mean.float(), module.running_var.float()
        scale = torch.rsqrt(var + module.eps)
        if module.affine:
            scale = scale * module.weight.float()
        shift = -mean * scale
        if module.affine:
            shift = shift + module.bias.float()
        return xf * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    assert isinstance(module, nn.InstanceNorm2d) and not module.track_running_stats, type(module)
    mean = xf.mean(dim=(2, 3), keepdim=True)
    var = xf.var(dim=(2, 3), unbiased=False, keepdim=True)
    y = (xf - mean) * torch.rsqrt(var + module.eps)
    if module.affine:
        y = y * module.weight.float().view(1, -1, 1, 1) + module.bias.float().view(1, -1, 1, 1)
    return y


def norm_act(module: nn.Module, x: torch.Tensor, relu: bool) -> torch.Tensor:
    """[relu](module(x)) for the norm layers of the encoders: on the HIP kernels in GPU training mode; the GPU cases those kernels do not
    cover (eval-mode BatchNorm after `freeze_bn()`, unaligned planes, ...) as torch element-wise arithmetic (`_norm_elementwise`), never on
    the vendor library; GroupNorm on a GPU tensor is refused (the TRAINING kernels have no GroupNorm: training.forward_train says so before the
    first launch; the inference engine runs it, extractor.py BasicEncoder._group_stats).
    CPU tensors (the oracle tests) run the module itself."""
    mode = _supported(module, x)
    if mode is None:
        if ENABLED and x.is_cuda and not isinstance(module, nn.Sequential):     # ("none" norm = empty Sequential: nothing to compute)
            if isinstance(module, nn.BatchNorm2d) or (isinstance(module, nn.InstanceNorm2d) and not module.track_running_stats):
                y = _norm_elementwise(module, x)
                return torch.relu_(y) if relu else y
            raise hip.BflowHipError(f"norm_act({type(module).__name__}) on a GPU tensor: only InstanceNorm2d / BatchNorm2d are implemented "
                                    "(there is no library fall-back)")
        y = module(x)
        return torch.relu_(y) if relu else y
    affine = mode == 1
    return _NormActFn.apply(x, module.weight if affine else None, module.bias if affine else None, module, mode, relu)
