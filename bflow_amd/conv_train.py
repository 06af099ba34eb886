"""Conv2d forward AND backward on the hand-written split-fp16 conv engine -- the convolutions of the training path.

The reference trains with `torch.nn.Conv2d` under autograd (models/raft_utils/extractor.py:5-125, models/raft_spline/update.py:8-126,
modules/raft_spline.py:63-188).  Here a convolution module routes GPU tensors under autograd through `_ConvFn`:

  forward   y = conv(x, w) + b                      bflow_conv_split (blocked split tensors, fp16 MFMA x 3, fp32 accumulation)
  dgrad     dx = conv_transpose(dy, w)              the SAME engine: a stride-1 convolution of dy (zero-dilated for stride 2) with
                                                    the filter flipped in space and transposed in (cin, cout)
  wgrad     dw[co, ci, r, q] = sum_k dy[k, co] x[k + (r, q), ci]      (the contraction index is the PIXEL)
                                                    stride-1 3x3 / 1x5 / 5x1 / 1x1: bflow_conv_wgrad_halo, a dedicated MFMA kernel on the SAME
                                                    blocked split tensors the forward (x) and the input-gradient pass (dy) staged --
                                                    nothing is re-packed; pixel-contiguous fragments come from the row-major
                                                    [pixel][channel] LDS tiles through ds_read_b64_tr_b16 (hardware transpose read);
                                                    every other shape (stride 2, 7x7): the engine's 1x1 "convolution" as a GEMM:
                                                    bflow_wgrad_pack re-blocks x (one shifted copy per filter tap) and dy so that
                                                    pixels sit in the 32-wide block position; "image" = (tap, k-chunk), "filter" =
                                                    the packed operand of that k-chunk (bflow_conv_desc_t.weight_sets); the k-chunks
                                                    (split-K, to fill the chip) are summed by bflow_wgrad_reduce
  dbias     sum of dy over (batch, y, x)            bflow_grad_stats: the same pass over dy that finds the scale below

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
MIN_CHUNK, TARGET_WGS = 8, 768      # split-K of the weight gradient: k-blocks per chunk at least / workgroups aimed at
HALO_WGRAD = True                   # stride-1 3x3 / 1x5 / 5x1 / 1x1: bflow_conv_wgrad_halo (no re-packing); False = the pack + GEMM path everywhere
_DEBUG_CMP = None                    # tools: a list collects (relative difference halo vs pack-GEMM weight gradient, shape, scale) per call
CAPTURE_EPOCH = 0                   # != 0 while training.GraphedTrainStep captures: every filter is packed ONCE inside that capture (recorded into
                                    # the graph, so that replays re-pack the weights AdamW just wrote), whatever the caches held before
_TARGET = 8192.0                    # max |dy| after scaling: well inside fp16 (65504), 13 bits of head-room for sums of products


class _PackCache:
    """Packed filters of one convolution (forward: the filter itself; backward: flipped in space, transposed in (cin, cout)), keyed on
    the storage and in-place version of the SOURCE parameters (an optimiser step bumps the version).  The tensor a pack was made from
    is kept alive next to it: a derived filter (torch.cat of two gate filters, the flipped copy) is a fresh tensor on every call, and a
    key on ITS address could match a freed predecessor's."""

    def __init__(self):
        self._key = {"fwd": None, "bwd": None}
        self._store = {"fwd": None, "bwd": None}
        self._epoch = {"fwd": 0, "bwd": 0}

    def get(self, which: str, weight: torch.Tensor, srcs):
        key = tuple((t.data_ptr(), hip.tensor_version(t), str(t.device)) for t in srcs)
        if self._key[which] != key or (CAPTURE_EPOCH and self._epoch[which] != CAPTURE_EPOCH):
            with torch.no_grad():
                w = weight.detach().float().contiguous()
                # "bwd": flipped in space, transposed in (cin, cout) -- by the pack kernel itself (bflow_conv_pack_weights_adjoint)
                packed = S.PackedConvWeight().get(w, adjoint=which == "bwd")    # a fresh packer: nothing to mistake for
            self._store[which], self._key[which], self._epoch[which] = (w, packed), key, CAPTURE_EPOCH
        return self._store[which][1]


def _conv_forward(x: torch.Tensor, packed, stride: int, padding: Tuple[int, int], bias, in_scale=None, out_scale=None) -> torch.Tensor:
    """fp32 NCHW -> engine -> fp32 NCHW (bflow_norm_act_split [x in_scale] -> bflow_conv_split -> bflow_blocked_f32_to_nchw [x out_scale]).
    The result leaves the engine as blocked fp32, not as a split pair: scaled gradients may exceed the split format's 65504."""
    xs = S.from_nchw(x, in_scale)
    _, yf = S.conv(xs, packed, stride=stride, padding=padding, shift=bias, want_split=False, want_f32=True)
    cout, _, kh, kw, _ = packed[1]
    Ho, Wo = (x.shape[2] + 2 * padding[0] - kh) // stride + 1, (x.shape[3] + 2 * padding[1] - kw) // stride + 1
    return S.blocked_f32_to_nchw(yf, cout, Ho, Wo, out_scale)


def _halo_wgrad_ok(stride: int, padding, ksize) -> bool:
    """bflow_conv_wgrad_halo: stride 1, "same" padding, 3x3 / 1x5 / 5x1 / 1x1."""
    kh, kw = ksize
    return HALO_WGRAD and stride == 1 and tuple(padding) == (kh // 2, kw // 2) and (kh, kw) in ((3, 3), (1, 5), (5, 1), (1, 1))


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, cache, srcs, relu=False):
        x = x.float().contiguous()
        cout, cin, kh, kw = weight.shape
        ctx.stride, ctx.padding, ctx.cache, ctx.srcs, ctx.has_bias = stride, padding, cache, srcs, bias is not None
        ctx.xshape = tuple(x.shape)
        with torch.no_grad():
            packed = cache.get("fwd", weight, srcs)
            xs = S.from_nchw(x)
            _, yf = S.conv(xs, packed, stride=stride, padding=padding, shift=None if bias is None else bias.detach().float().contiguous(),
                           act=S.ACT_RELU if relu else S.ACT_NONE, want_split=False, want_f32=True)
            Ho, Wo = (x.shape[2] + 2 * padding[0] - kh) // stride + 1, (x.shape[3] + 2 * padding[1] - kw) // stride + 1
            y = S.blocked_f32_to_nchw(yf, cout, Ho, Wo)
        ctx.halo = _halo_wgrad_ok(stride, padding, (kh, kw))
        # the weight gradient reads X: in the engine's own layout (as staged for the forward) where bflow_conv_wgrad_halo takes it
        ctx.relu = bool(relu)
        ctx.save_for_backward(xs.planes if ctx.halo else x, weight, *((y,) if relu else ()))      # ReLU in the conv epilogue: its mask is y > 0
        return y

    @staticmethod
    def backward(ctx, dy):
        xsaved, w = ctx.saved_tensors[:2]
        if ctx.relu:
            dy = torch.ops.aten.threshold_backward(dy.float(), ctx.saved_tensors[2], 0.0)
        stride, (ph, pw), cache = ctx.stride, ctx.padding, ctx.cache
        cout, cin, kh, kw = w.shape
        B, _, H, W = ctx.xshape
        dy = dy.float().contiguous()
        _, _, Ho, Wo = dy.shape
        dx = dw = db = None
        with torch.no_grad():
            sc, dbs = S.grad_stats(dy, _TARGET)                  # {s, 1/s, s per channel} on the device and the bias gradient, one pass over dy
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = dbs
            s, inv, svec = sc[0:1], sc[1:2], sc[2:]
            gs = None
            if stride == 1 and (ctx.needs_input_grad[0] or (ctx.halo and ctx.needs_input_grad[1])):
                gs = S.from_nchw(dy, scale_vec=svec)             # dY in the engine's layout, pre-scaled: shared by dgrad and wgrad
            if ctx.needs_input_grad[0]:
                packed = cache.get("bwd", w, ctx.srcs)
                if stride == 1:
                    g_in = gs
                else:                                            # zero-dilated dy: position (yo*stride, xo*stride) of the stride-1 output grid
                    g = torch.zeros((B, cout, H + 2 * ph - kh + 1, W + 2 * pw - kw + 1), dtype=torch.float32, device=dy.device)
                    g[:, :, ::stride, ::stride][:, :, :Ho, :Wo] = dy
                    g_in = S.from_nchw(g, scale_vec=svec)
                _, xf = S.conv(g_in, packed, stride=1, padding=(kh - 1 - ph, kw - 1 - pw), want_split=False, want_f32=True)
                dx = S.blocked_f32_to_nchw(xf, cin, H, W, inv)
            if ctx.needs_input_grad[1]:
                if ctx.halo:
                    xs = S.SplitTensor(xsaved, H, W, cin)
                    dw = S.conv_wgrad_halo(xs, gs, cout, cin, (kh, kw), inv)
                    if _DEBUG_CMP is not None:
                        xr = xs.to_nchw()[:, :cin]
                        ref = _weight_grad(xr.contiguous(), dy, s, inv, (kh, kw), stride, (ph, pw))
                        e = float((dw - ref).abs().max() / (ref.abs().max() + 1e-30))
                        _DEBUG_CMP.append((e, (B, cin, cout, H, W, kh, kw), float(ref.abs().max())))
                else:
                    dw = _weight_grad(xsaved, dy, s, inv, (kh, kw), stride, (ph, pw))
        return dx, dw, db, None, None, None, None, None


def _pad_ratio(n: int, t: int = 128) -> float:
    return ((n + t - 1) // t * t) / n


def _weight_grad(x: torch.Tensor, dy: torch.Tensor, s: torch.Tensor, inv: torch.Tensor, ksize, stride: int, padding) -> torch.Tensor:
    """dw (Cout, Cin, KH, KW): engine launches over k-chunk "images" (split-K, to fill the chip) + a sum over the chunks.
    Two orientations of the same GEMM  dw[co, (tap, ci)] = sum_k dy[k, co] x_tap[k, ci]:
      A  the GEMM's rows (the engine's 128-row pixel tile) = ci, one image per (tap, chunk), filter = packed dy;
      B  rows = co, one image per chunk, filter = packed x with all taps as its output channels (small cin: the stem, convf1)."""
    B, cin, H, W = x.shape
    _, cout, Ho, Wo = dy.shape
    kh, kw = ksize
    taps = kh * kw
    kb = (B * Ho * Wo + 31) // 32
    use_b = cin < 64 or _pad_ratio(cout) < _pad_ratio(cin)
    if not use_b:                                                                                      # ---- orientation A
        cout_pad = (cout + 127) // 128 * 128
        tiles = ((cin + 127) // 128) * ((cout + 63) // 64) * taps
        G = max(1, min(kb // MIN_CHUNK, (TARGET_WGS + tiles - 1) // tiles))
        kbg = (kb + G - 1) // G
        xp = S.wgrad_pack(x, (Ho, Wo), ksize, stride, padding, rows=cin, k_blocks=kbg * G)                 # (2, taps, kb, cin, 32)
        gp = S.wgrad_pack(dy, (Ho, Wo), rows=cout_pad, k_blocks=kbg * G, scale=s)                          # (2, 1, kb, cout_pad, 32)
        xt = S.SplitTensor(xp.view(2, taps * G, kbg, cin, 32), cin, 1)                                     # image = (tap, chunk)
        packed = (gp.view(2, kbg * G, cout_pad, 32), (cout, kbg * 32, 1, 1, cout_pad))                     # filter set g = k-tiles of chunk g
        _, part = S.conv(xt, packed, want_split=False, want_f32=True, weight_sets=G)                       # (taps*G, cout blocks, cin, 32)
        return S.wgrad_reduce(part, G, cout, cin, ksize, 0, inv)
    else:
        n = taps * cin
        n_pad = (n + 127) // 128 * 128
        tiles = ((cout + 127) // 128) * ((n + 63) // 64)
        G = max(1, min(kb // MIN_CHUNK, (TARGET_WGS + tiles - 1) // tiles))
        kbg = (kb + G - 1) // G
        gp = S.wgrad_pack(dy, (Ho, Wo), rows=cout, k_blocks=kbg * G, scale=s)                              # (2, 1, kb, cout, 32)
        xp = S.wgrad_pack(x, (Ho, Wo), ksize, stride, padding, rows=n_pad, k_blocks=kbg * G, taps_in_rows=True)   # (2, kb, n_pad, 32)
        gt = S.SplitTensor(gp.view(2, G, kbg, cout, 32), cout, 1)                                          # image = chunk, rows = co
        packed = (xp, (n, kbg * 32, 1, 1, n_pad))
        _, part = S.conv(gt, packed, want_split=False, want_f32=True, weight_sets=G)                       # (G, n blocks, cout, 32)
        return S.wgrad_reduce(part, G, cout, cin, ksize, 1, inv)


def _refuse_library(x: torch.Tensor, what: str):
    """A GPU call that does not take the engine path would silently run on the vendor library (MIOpen): refuse it.  Every filter shape of
    the network is on the engine (with or without grad; a frozen encoder too), so only a foreign filter (grouped / dilated / ...) ends here.
    `ENABLED = False` is the explicit A/B switch of tools / tests and keeps torch's path."""
    if ENABLED and x.is_cuda:
        raise hip.BflowHipError(f"{what} on a GPU tensor with a filter the conv engine does not implement (grouped, dilated, non-zero padding mode, "
                                "stride other than 1 / 2): there is no library fall-back")


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose GPU forward under autograd runs (and differentiates) on the HIP conv engine.  State-dict compatible.
    A GPU forward that cannot take that path raises (no MIOpen fall-through); CPU tensors behave like nn.Conv2d."""

    def forward(self, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
        # every GPU call with a filter the engine supports runs on it -- also with grad disabled or a frozen encoder (nothing requires grad:
        # autograd then records nothing and _ConvFn.forward is just the engine's forward)
        if (ENABLED and x.is_cuda and self.groups == 1
                and self.dilation == (1, 1) and self.stride[0] == self.stride[1] and self.stride[0] in (1, 2) and self.padding_mode == "zeros"
                and not isinstance(self.padding, str)):
            cache = self.__dict__.setdefault("_hip_pack", _PackCache())
            return _ConvFn.apply(x, self.weight, self.bias, self.stride[0], tuple(self.padding), cache, (self.weight,), relu)
        _refuse_library(x, "Conv2d.forward")
        y = super().forward(x)
        return torch.relu(y) if relu else y

    def relu(self, x: torch.Tensor) -> torch.Tensor:
        """relu(conv(x)): the activation runs in the convolution's epilogue on the engine path (one launch less, forward)."""
        return self.forward(x, relu=True)


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias, padding, cache: _PackCache, srcs, stride: int = 1) -> torch.Tensor:
    """Functional form for derived filters (the merged z|r convolution of the training forward); `srcs` = the parameters `weight` was
    built from (the cache key)."""
    if ENABLED and x.is_cuda:
        return _ConvFn.apply(x, weight, bias, stride, tuple(padding), cache, tuple(srcs))
    _refuse_library(x, "conv2d")
    return torch.nn.functional.conv2d(x, weight, bias, stride=stride, padding=padding)
