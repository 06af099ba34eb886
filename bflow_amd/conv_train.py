"""Conv2d forward AND backward on the hand-written split-fp16 conv engine -- the convolutions of the training path.

The reference trains with `torch.nn.Conv2d` under autograd (models/raft_utils/extractor.py:5-125, models/raft_spline/update.py:8-126,
modules/raft_spline.py:63-188).  Here a convolution module routes GPU tensors under autograd through `_ConvFn`:

  forward   y = conv(x, w) + b                      bflow_conv_split (blocked split tensors, fp16 MFMA x 3, fp32 accumulation)
  dgrad     dx = conv_transpose(dy, w)              the SAME engine: a stride-1 convolution of dy (zero-dilated for stride 2) with
                                                    the filter flipped in space and transposed in (cin, cout)
  wgrad     dw[co, ci, r, q] = sum_k dy[k, co] x[k + (r, q), ci]
                                                    the SAME engine as a batch of 1x1 "convolutions" whose contraction index is the
                                                    PIXEL index: bflow_wgrad_pack re-blocks x (one shifted copy per filter tap) and
                                                    dy so that pixels sit in the 32-wide block position; "image" = (tap, k-chunk),
                                                    "filter" = the packed dy of that k-chunk (bflow_conv_desc_t.weight_sets); the
                                                    k-chunks (split-K, to fill the chip) are summed afterwards
  dbias     sum of dy over (batch, y, x)            torch reduction

Gradients are tiny (1e-4 .. 1e-9) and the split format keeps 22 bits only inside fp16's normal range, so dy is pre-scaled by a
power of two chosen on the device from max|dy| (no host synchronisation) and the results are scaled back -- exact in binary
floating point.  Normalisation layers, activations and the GRU gate arithmetic stay torch element-wise ops under autograd.
CPU tensors (the oracle tests) and inference-mode calls fall through to `torch.nn.Conv2d.forward`; the inference PRODUCT path never
comes here (it runs `forward_split` / `step_split`).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from . import hip
from . import split as S

ENABLED = True                      # tools / A-B timing: False = torch (MIOpen) convolutions under autograd
_TARGET = 8192.0                    # max |dy| after scaling: well inside fp16 (65504), 13 bits of head-room for sums of products


def _pow2_scale(t: torch.Tensor) -> torch.Tensor:
    """2^floor(log2(_TARGET / max|t|)) as a 1-element device tensor (1 for an all-zero / non-finite tensor)."""
    amax = t.detach().abs().amax().float()
    e = torch.floor(torch.log2(_TARGET / amax))
    e = torch.where(torch.isfinite(e), e, torch.zeros_like(e)).clamp(-60.0, 60.0)
    return torch.exp2(e).reshape(1)


class _PackCache:
    """Packed filters keyed on the parameter's storage and in-place version (an optimiser step bumps the version)."""

    def __init__(self):
        self.fwd = S.PackedConvWeight()
        self.bwd = S.PackedConvWeight()
        self._flip_key = None
        self._flip = None

    def flipped(self, w: torch.Tensor) -> torch.Tensor:
        key = (w.data_ptr(), w._version)
        if self._flip_key != key:
            with torch.no_grad():
                self._flip = w.detach().flip(2, 3).transpose(0, 1).contiguous()
            self._flip_key = key
        return self._flip


def _conv_forward(x: torch.Tensor, packed, stride: int, padding: Tuple[int, int], bias) -> torch.Tensor:
    """fp32 NCHW -> engine -> fp32 NCHW.  The result leaves the engine as blocked fp32 (not as a split pair: scaled gradients may
    exceed the split format's 65504) and is re-ordered by one torch copy."""
    xs = S.from_nchw(x)
    _, yf = S.conv(xs, packed, stride=stride, padding=padding, shift=bias, want_split=False, want_f32=True)
    cout = packed[1][0]
    B, cb, rows, _ = yf.shape
    kh, kw = packed[1][2], packed[1][3]
    Ho, Wo = (x.shape[2] + 2 * padding[0] - kh) // stride + 1, (x.shape[3] + 2 * padding[1] - kw) // stride + 1
    out = yf[:, :, :Ho * Wo].permute(0, 1, 3, 2).reshape(B, cb * 32, Ho, Wo)       # a copy (the permuted tensor is not viewable as NCHW)
    # (never hand autograd a VIEW as a Function output: the in-place ReLUs that follow would be refused)
    return out if cout == cb * 32 and rows == Ho * Wo else out[:, :cout].clone()


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, cache):
        x = x.float().contiguous()
        ctx.save_for_backward(x, weight)
        ctx.stride, ctx.padding, ctx.cache, ctx.has_bias = stride, padding, cache, bias is not None
        with torch.no_grad():
            return _conv_forward(x, cache.fwd.get(weight), stride, padding, None if bias is None else bias.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, (ph, pw), cache = ctx.stride, ctx.padding, ctx.cache
        cout, cin, kh, kw = w.shape
        B, _, H, W = x.shape
        dy = dy.float().contiguous()
        _, _, Ho, Wo = dy.shape
        dx = dw = db = None
        with torch.no_grad():
            s = _pow2_scale(dy)
            inv = 1.0 / s
            if ctx.needs_input_grad[0]:
                dys = dy * s
                if stride > 1:                                   # zero-dilated dy: position (yo*stride, xo*stride) of the stride-1 output grid
                    z = torch.zeros((B, cout, H + 2 * ph - kh + 1, W + 2 * pw - kw + 1), dtype=torch.float32, device=dy.device)
                    z[:, :, ::stride, ::stride][:, :, :Ho, :Wo] = dys
                    dys = z
                dx = _conv_forward(dys, cache.bwd.get(cache.flipped(w)), 1, (kh - 1 - ph, kw - 1 - pw), None) * inv
            if ctx.needs_input_grad[1]:
                dw = _weight_grad(x, dy, s, (kh, kw), stride, (ph, pw)) * inv
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = dy.sum(dim=(0, 2, 3))
        return dx, dw, db, None, None, None


def _weight_grad(x: torch.Tensor, dy: torch.Tensor, scale: torch.Tensor, ksize, stride: int, padding) -> torch.Tensor:
    """dw (Cout, Cin, KH, KW) * scale: one engine launch over (tap, k-chunk) "images" + a sum over the k-chunks."""
    B, cin, H, W = x.shape
    _, cout, Ho, Wo = dy.shape
    kh, kw = ksize
    taps = kh * kw
    K = B * Ho * Wo
    kb = (K + 31) // 32
    cout_pad = (cout + 127) // 128 * 128
    # split-K: enough (tap, chunk, tile) workgroups for two per CU, chunks of at least 16 k-blocks
    tiles = ((cin + 127) // 128) * ((cout + 63) // 64) * taps
    G = max(1, min(kb // 16, (512 + tiles - 1) // tiles))
    kbg = (kb + G - 1) // G
    kb_pad = kbg * G
    xp = S.wgrad_pack(x, (Ho, Wo), ksize, stride, padding, rows=cin, k_blocks=kb_pad)              # (2, taps, kb_pad, cin, 32)
    gp = S.wgrad_pack(dy, (Ho, Wo), rows=cout_pad, k_blocks=kb_pad, scale=scale)                    # (2, 1, kb_pad, cout_pad, 32)
    xt = S.SplitTensor(xp.view(2, taps * G, kbg, cin, 32), cin, 1)                                  # image = (tap, chunk): cin "pixels" x kbg*32 "channels"
    packed = (gp.view(2, kb_pad, cout_pad, 32), (cout, kbg * 32, 1, 1, cout_pad))                   # filter set g = k-tiles [g*kbg, (g+1)*kbg)
    _, part = S.conv(xt, packed, want_split=False, want_f32=True, weight_sets=G)                    # (taps*G, cout/32 blocks, cin, 32)
    cb = part.shape[1]
    dw = part.view(taps, G, cb, cin, 32).sum(dim=1)                                                 # (taps, cb, cin, 32)
    return dw.permute(1, 3, 2, 0).reshape(cb * 32, cin, kh, kw)[:cout].contiguous()


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose GPU forward under autograd runs (and differentiates) on the HIP conv engine.  State-dict compatible."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (ENABLED and x.is_cuda and torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad) and self.groups == 1
                and self.dilation == (1, 1) and self.stride[0] == self.stride[1] and self.stride[0] in (1, 2) and self.padding_mode == "zeros"
                and not isinstance(self.padding, str)):
            cache = self.__dict__.setdefault("_hip_pack", _PackCache())
            return _ConvFn.apply(x, self.weight, self.bias, self.stride[0], tuple(self.padding), cache)
        return super().forward(x)


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias, padding, cache: _PackCache, stride: int = 1) -> torch.Tensor:
    """Functional form for derived filters (the merged z|r convolution of the training forward)."""
    if ENABLED and x.is_cuda and torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        return _ConvFn.apply(x, weight, bias, stride, tuple(padding), cache)
    return torch.nn.functional.conv2d(x, weight, bias, stride=stride, padding=padding)
