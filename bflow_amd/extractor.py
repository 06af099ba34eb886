"""Feature / context encoder (K4) -- the drop-in for models/raft_utils/extractor.py:5-125.

Same module tree and parameter names as the reference (so its checkpoints load), inference only.  `forward_split` is the product
path: every convolution runs on the hand-written split-fp16 MFMA engine (csrc/conv_split.hip: 7x7/2 stem with im2col in LDS,
halo kernel for the 3x3s, generic kernel for stride 2 / 1x1), bias / folded BatchNorm / ReLU / InstanceNorm statistics live in the
conv epilogues, and one normalise + activate + residual kernel sits between convolutions.  `forward` is the plain nn.Module forward
(torch ops): the differentiable training path (bflow_amd/training.py) runs on it under autograd; inference never does.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from . import split as S
from .conv_train import Conv2d   # nn.Conv2d whose GPU training forward / backward run on the HIP conv engine
from .norm_train import norm_act  # [relu](norm(x)): hand-written forward / backward in GPU training mode

STATS_R = 8
FUSE_NORM_IN = os.environ.get("BFLOW_NO_NORM_IN") is None     # A/B switch (tools/): conv2 of a residual block normalises its input on load
# round 4: the stem's norm + ReLU is never materialised either (layer1.0.conv1 normalises on load, the block-end kernel takes the residual as
# relu(norm_b(b))): 3.581 / 3.597 / 3.614 vs 3.604 / 3.632 / 3.626 ms per frame over three alternating pairs.  BFLOW_NO_NORM_IN_STEM for A/B.
FUSE_NORM_IN_STEM = os.environ.get("BFLOW_NO_NORM_IN_STEM") is None
# round 4: BatchNorm encoders (the context encoder) take relu(x + y) as conv2's epilogue (bflow_conv_desc_t.gate = 3): six block-end launches
# less; 3.581 / 3.569 / 3.593 vs 3.606 / 3.656 / 3.619 ms per frame over three alternating pairs.  BFLOW_NO_FUSE_RESIDUAL for A/B.
FUSE_RESIDUAL = os.environ.get("BFLOW_NO_FUSE_RESIDUAL") is None


def _make_norm(kind: str, channels: int) -> nn.Module:
    if kind == "group":
        return nn.GroupNorm(num_groups=channels // 8, num_channels=channels)
    if kind == "batch":
        return nn.BatchNorm2d(channels)
    if kind == "instance":
        return nn.InstanceNorm2d(channels)
    if kind == "none":
        return nn.Sequential()
    raise NotImplementedError(kind)


class ResidualBlock(nn.Module):
    """extractor.py:5-55: two 3x3 convs (+norm+relu), 1x1 strided shortcut when stride != 1."""

    def __init__(self, in_planes: int, planes: int, norm_fn: str = "group", stride: int = 1):
        super().__init__()
        self.conv1 = Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, padding=1)
        self.norm1 = _make_norm(norm_fn, planes)
        self.norm2 = _make_norm(norm_fn, planes)
        self.downsample = None
        if stride != 1:
            # registered under both names, exactly like the reference (norm3 and downsample.1 share parameters)
            self.norm3 = _make_norm(norm_fn, planes)
            self.downsample = nn.Sequential(Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = norm_act(self.norm1, self.conv1(x), True)        # relu(norm(conv)): csrc/norm_train.hip in GPU training mode, else the modules
        y = norm_act(self.norm2, self.conv2(y), True)
        if self.downsample is not None:
            x = norm_act(self.norm3, self.downsample[0](x), False)
        return F.relu_(x + y)


class BasicEncoder(nn.Module):
    """extractor.py:58-125: 7x7/2 stem, three stages of two residual blocks (64, 96/2, 128/2), 1x1 projection."""

    def __init__(self, input_dim: int = 3, output_dim: int = 128, norm_fn: str = "batch"):
        super().__init__()
        self.norm_fn = norm_fn
        if norm_fn == "group":
            self.norm1 = nn.GroupNorm(num_groups=8, num_channels=64)
        else:
            self.norm1 = _make_norm(norm_fn, 64)
        self.conv1 = Conv2d(input_dim, 64, kernel_size=7, stride=2, padding=3)
        planes = [(64, 1), (96, 2), (128, 2)]
        cin = 64
        for idx, (dim, stride) in enumerate(planes, start=1):
            setattr(self, f"layer{idx}", nn.Sequential(ResidualBlock(cin, dim, norm_fn, stride=stride),
                                                       ResidualBlock(dim, dim, norm_fn, stride=1)))
            cin = dim
        self.conv2 = Conv2d(128, output_dim, kernel_size=1)
        for m in self.modules():   # extractor.py:85-92
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------------------------------------ split-fp16 engine
    def _packed(self, name: str, conv: nn.Conv2d):
        cache: Dict[str, S.PackedConvWeight] = self.__dict__.setdefault("_pack_cache", {})
        if name not in cache:
            cache[name] = S.PackedConvWeight()
        return cache[name].get(conv.weight)

    def _bn_affine(self, norm: nn.BatchNorm2d, conv_bias: Optional[torch.Tensor]):
        """Eval-mode BatchNorm folded with the conv bias: y = conv * scale + shift.  Cached per (norm, bias) until a source
        tensor changes (load_state_dict, .to(device)), so the steady-state forward launches nothing for it."""
        cache = self.__dict__.setdefault("_affine_cache", {})
        if not isinstance(norm, nn.BatchNorm2d):      # norm_fn = 'none' (extractor.py:33-37,69-70: an empty nn.Sequential): y = conv + bias
            key = (conv_bias.data_ptr(), hip.tensor_version(conv_bias))
            hit = cache.get(id(conv_bias))
            if hit is None or hit[0] != key:
                hit = (key, torch.ones_like(conv_bias, dtype=torch.float32), conv_bias.detach().float().contiguous())
                cache[id(conv_bias)] = hit
            return hit[1], hit[2]
        srcs = [norm.weight, norm.bias, norm.running_mean, norm.running_var] + ([conv_bias] if conv_bias is not None else [])
        key = tuple((t.data_ptr(), hip.tensor_version(t)) for t in srcs)
        hit = cache.get(id(norm))
        if hit is None or hit[0] != key:
            with torch.no_grad():
                scale = norm.weight / torch.sqrt(norm.running_var + norm.eps)
                shift = norm.bias - norm.running_mean * scale
                if conv_bias is not None:
                    shift = shift + conv_bias * scale
                hit = (key, scale.float().contiguous(), shift.float().contiguous())
            cache[id(norm)] = hit
        return hit[1], hit[2]

    @staticmethod
    def _group_stats(st: torch.Tensor, norm: nn.GroupNorm, hw: int) -> torch.Tensor:
        """GroupNorm (extractor.py:13-19,63-64: groups of 8 channels, per-channel affine) through the InstanceNorm machinery.  The conv
        epilogues deliver per-(image, channel) sums and sums of squares (fp64, `st`: (R, n, C, 2)); a group's statistics are the sums over
        its channels, and the normalisation of channel c of image b is the affine map x * mul + add with mul = rstd_g * gamma_c,
        add = beta_c - mean_g * mul.  The table handed to the normalisation kernel holds (mul, add) THEMSELVES (its direct mode, eps = -1:
        csrc/conv_engine.h norm_coeffs), so gamma may have any sign or be zero (round 5 rewrote the table to a (mean', var') pair and
        needed gamma > 0).  A few tiny fp64 torch kernels per normalised convolution: this norm_fn is supported for the constructor
        surface of the reference (no shipped experiment uses it), not tuned."""
        s = st.sum(0)                                        # (n, C, 2)
        n, C, _ = s.shape
        G = norm.num_groups
        cpg = C // G
        sg = s.view(n, G, cpg, 2).sum(2)                     # (n, G, 2)
        cnt = float(hw * cpg)
        mean_g = sg[..., 0] / cnt
        var_g = (sg[..., 1] / cnt - mean_g * mean_g).clamp_min(0.0)
        rstd_g = torch.rsqrt(var_g + norm.eps)
        mul = rstd_g.repeat_interleave(cpg, dim=1) * norm.weight.double()
        add = norm.bias.double() - mean_g.repeat_interleave(cpg, dim=1) * mul
        return torch.stack((mul, add), dim=-1).unsqueeze(0).contiguous()   # (1, n, C, 2): direct coefficients

    def forward_split(self, x, out_rows: Optional[int] = None, trunk_only: bool = False, after_layer=None, out: Optional["S.SplitTensor"] = None,
                      out_ready=None, out_gain: Optional[float] = None):
        """The same network on the split-fp16 MFMA engine (csrc/conv_split.hip): channels-last split activations, implicit-GEMM
        convolutions with bias / folded BatchNorm / ReLU / InstanceNorm statistics fused into their epilogues, and one
        normalise+activate+residual kernel between convolutions; the 7x7 stem reads the fp32 NCHW input directly (im2col in LDS).  x: (n, c_in, H, W) fp32 -> SplitTensor (n, H/8, W/8, out_dim); `out_rows` pads the
        pixel rows of the result with zeros (K5 wants a multiple of 128).  after_layer = (i, fn): fn() is called once the launches of
        layer i are enqueued (the caller forks work there that should start behind them, e.g. the context encoder).  out / out_ready: see the
        last convolution."""
        kind = self.norm_fn
        assert kind in ("instance", "batch", "group", "none"), kind
        group = kind == "group"
        if group:
            kind = "instance"          # the statistics path; every table is rewritten by `_group_stats` behind its convolution, nothing is fused on load
        elif kind == "none":
            kind = "batch"             # the affine path with the identity (scale 1, shift = conv bias)
        neps = -1.0 if group else 1e-5  # eps of the normalisation kernel; -1 = the table holds (mul, add) directly (GroupNorm's own eps is inside them)
        fuse_in = FUSE_NORM_IN and not group
        n = x.shape[0]
        dev = x.device
        # InstanceNorm statistics of all 16 normalised convolutions: ONE zero-filled arena per forward.  Every table has STATS_R
        # replicas (workgroup w adds into replica w % R; the normalisation kernel sums them): a few thousand workgroups adding
        # fp64 atomics to the same 128 addresses per image serialise otherwise (+20 % on these launches, measured).
        R = STATS_R
        arena = torch.zeros((16 * R * n * 128 * 2,), dtype=torch.float64, device=dev) if kind == "instance" else None
        gs = (lambda st, norm, hw: self._group_stats(st, norm, hw)) if group else (lambda st, norm, hw: st)
        used = [0]

        def new_stats(c):
            st = arena[used[0]:used[0] + R * n * c * 2].view(R, n, c, 2)
            used[0] += R * n * c * 2
            return st

        # ---- stem: 7x7/2 straight from the fp32 NCHW input (im2col in LDS, csrc/conv_split.hip conv_stem_kernel)
        cache = self.__dict__.setdefault("_pack_cache", {})
        if "stem" not in cache:
            cache["stem"] = S.PackedStemWeight()
        pk0 = cache["stem"].get(self.conv1.weight)
        c0 = self.conv1.out_channels
        h0, w0 = (x.shape[2] - 1) // 2 + 1, (x.shape[3] - 1) // 2 + 1
        xin = x if isinstance(x, (S.ChannelWindows, S.StemInput)) else x.contiguous()
        raw0 = None                # (f0, st0): the stem's pre-normalisation output when relu(norm1(conv1)) is never materialised
        if kind == "instance":     # the conv bias cancels under InstanceNorm; statistics come out of the epilogue
            st0 = new_stats(c0)
            # (the conv bias cancels under InstanceNorm and is never added; under GroupNorm it does not: channels of a group differ in it)
            _, f0 = S.conv_stem(xin, pk0, stats=st0, want_split=False, want_f32=True, shift=self.conv1.bias if group else None)
            st0 = gs(st0, self.norm1, h0 * w0)
            if fuse_in and FUSE_NORM_IN_STEM and c0 <= 128 and c0 % 32 == 0 and self.layer1[0].conv1.stride[0] == 1:
                # extractor.py:113 `x = relu(norm1(conv1(x)))` has two consumers: layer1.0.conv1 normalises it on load (x_raw), and the
                # block's residual `x + y` (extractor.py:55) takes it as relu(norm_b(b)) inside the block-end kernel: one launch and one
                # read + write of the half-resolution map less
                raw0, cur = (f0, st0), None
            else:
                cur, _ = S.norm_act(f0, (n, h0, w0, c0), stats_a=st0, act_a=S.ACT_RELU, eps=neps)
        else:                      # folded BatchNorm + ReLU in the epilogue: the stem is ONE launch
            sc, sh = self._bn_affine(self.norm1, self.conv1.bias)
            cur, _ = S.conv_stem(xin, pk0, scale=sc, shift=sh, act=S.ACT_RELU)

        def conv_norm(name, conv, norm, src, stride, relu):
            """conv (+ bias) -> norm -> optional relu.  instance: returns (fp32 NHWC, stats) for the fused norm kernel;
            batch: affine + relu folded into the conv epilogue, returns (fp32 NHWC, None)."""
            pad = conv.padding
            pk = self._packed(name, conv)
            if kind == "instance":
                st = new_stats(conv.out_channels)
                _, f = S.conv(src, pk, stride=stride, padding=pad, want_split=False, want_f32=True, stats=st, shift=conv.bias if group else None)
                ho_, wo_ = out_hw(src, conv, stride)
                return f, gs(st, norm, ho_ * wo_)
            sc, sh = self._bn_affine(norm, conv.bias)
            _, f = S.conv(src, pk, stride=stride, padding=pad, scale=sc, shift=sh, act=S.ACT_RELU if relu else S.ACT_NONE,
                          want_split=False, want_f32=True)
            return f, None

        def out_hw(src, conv, stride):
            _, H_, W_, _ = src.shape
            return ((H_ + 2 * conv.padding[0] - conv.kernel_size[0]) // stride + 1, (W_ + 2 * conv.padding[1] - conv.kernel_size[1]) // stride + 1)

        def conv_norm_relu_split(name, conv, norm, src, stride):
            pk = self._packed(name, conv)
            ho_wo = out_hw(src, conv, stride)
            if kind == "instance":
                f, st = conv_norm(name, conv, norm, src, stride, True)
                out, _ = S.norm_act(f, (n, ho_wo[0], ho_wo[1], conv.out_channels), stats_a=st, act_a=S.ACT_RELU, eps=neps)
                return out
            sc, sh = self._bn_affine(norm, conv.bias)
            out, _ = S.conv(src, pk, stride=stride, padding=conv.padding, scale=sc, shift=sh, act=S.ACT_RELU)
            return out

        for li in (1, 2, 3):
            layer = getattr(self, f"layer{li}")
            for bi, blk in enumerate(layer):
                stride = blk.conv1.stride[0]
                pre = f"layer{li}.{bi}"
                if kind == "instance" and fuse_in and blk.conv1.out_channels <= 128 and blk.conv1.out_channels % 32 == 0:
                    # relu(norm1(conv1(x))) is never materialised: conv2 normalises conv1's fp32 output while it stages its halo
                    # (bflow_conv_desc_t.x_raw): the same arithmetic, one read + write of the activation and one launch less
                    if raw0 is not None and li == 1 and bi == 0:
                        st1 = new_stats(blk.conv1.out_channels)
                        f1 = S.conv_norm_in(raw0[0], (n, h0, w0, c0), raw0[1], self._packed(pre + ".conv1", blk.conv1), stats=st1, eps=self.norm1.eps)
                        ho, wo = h0, w0
                    else:
                        f1, st1 = conv_norm(pre + ".conv1", blk.conv1, blk.norm1, cur, stride, True)
                        ho, wo = out_hw(cur, blk.conv1, stride)
                    st2 = new_stats(blk.conv2.out_channels)
                    c2 = S.conv_norm_in(f1, (n, ho, wo, blk.conv1.out_channels), st1, self._packed(pre + ".conv2", blk.conv2), stats=st2,
                                        eps=blk.norm1.eps)
                    shape = (n, ho, wo, blk.conv2.out_channels)
                elif kind == "batch" and FUSE_RESIDUAL and blk.conv2.out_channels % 32 == 0:
                    # folded BatchNorm (the context encoder): relu(x + relu(bn2(conv2(y)))) (extractor.py:49-55) is conv2's epilogue -- the
                    # residual x (or the folded-BatchNorm down-sampling branch, written as a split tensor) enters as `gate_h` (GATE_RES):
                    # no block-end launch, no fp32 round trip of conv2's output
                    a1 = conv_norm_relu_split(pre + ".conv1", blk.conv1, blk.norm1, cur, stride)
                    resid = cur
                    if blk.downsample is not None:
                        scd, shd = self._bn_affine(blk.norm3, blk.downsample[0].bias)
                        resid, _ = S.conv(cur, self._packed(pre + ".downsample.0", blk.downsample[0]), stride=stride, padding=blk.downsample[0].padding,
                                          scale=scd, shift=shd)
                    sc2, sh2 = self._bn_affine(blk.norm2, blk.conv2.bias)
                    out_new = S.SplitTensor.empty(n, a1.H, a1.W, blk.conv2.out_channels, dev)
                    S.conv(a1, self._packed(pre + ".conv2", blk.conv2), padding=blk.conv2.padding, scale=sc2, shift=sh2, act=S.ACT_RELU,
                           gate=S.GATE_RES, gate_h=resid, out_split=out_new)
                    cur = out_new
                    continue
                else:
                    a1 = conv_norm_relu_split(pre + ".conv1", blk.conv1, blk.norm1, cur, stride)
                    c2, st2 = conv_norm(pre + ".conv2", blk.conv2, blk.norm2, a1, 1, True)
                    shape = (n, a1.H, a1.W, blk.conv2.out_channels)
                if blk.downsample is None and raw0 is not None and li == 1 and bi == 0:
                    # relu(x + relu(norm2(conv2))) with x = relu(norm1(stem)) taken from the stem's raw output
                    cur, _ = S.norm_act(c2, shape, stats_a=st2, act_a=S.ACT_RELU, b=raw0[0], stats_b=raw0[1], act_b=S.ACT_RELU, act_out=S.ACT_RELU, eps=neps)
                elif blk.downsample is None:
                    # relu(x + relu(norm2(conv2)))          (extractor.py:50-55)
                    cur, _ = S.norm_act(c2, shape, stats_a=st2, act_a=S.ACT_RELU if kind == "instance" else S.ACT_NONE, res=cur,
                                        act_out=S.ACT_RELU, eps=neps)
                else:
                    d, std = conv_norm(pre + ".downsample.0", blk.downsample[0], blk.norm3, cur, stride, False)
                    cur, _ = S.norm_act(c2, shape, stats_a=st2, act_a=S.ACT_RELU if kind == "instance" else S.ACT_NONE, b=d, stats_b=std,
                                        act_out=S.ACT_RELU, eps=neps)
            if after_layer is not None and after_layer[0] == li:
                after_layer[1]()
        if trunk_only:      # the caller applies the 1x1 projection itself (e.g. split into tanh / relu halves, raft.py:145-147)
            return cur
        pk = self._packed("conv2", self.conv2)
        bias = self.conv2.bias
        if out is not None:
            # a caller-owned output whose pad rows were zeroed elsewhere (`out_ready`: the event behind that fill) -- on the frame's context
            # branch, instead of a 7.5-us strided fill on the feature encoder's chain right in front of its last convolution
            if out_ready is not None:
                torch.cuda.current_stream().wait_event(out_ready)
            if out_gain is not None:      # the projection times a constant (see RAFTSpline._encode: feature dims the correlation kernel pads)
                gain = torch.full((self.conv2.out_channels,), out_gain, dtype=torch.float32, device=bias.device)
                out, _ = S.conv(cur, pk, scale=gain, shift=bias.detach().float() * out_gain, out_split=out, out_rows=out.rows)
            else:
                out, _ = S.conv(cur, pk, shift=bias, out_split=out, out_rows=out.rows)
        else:
            out, _ = S.conv(cur, pk, shift=bias, out_rows=out_rows)
        return out

    def forward(self, x: Union[torch.Tensor, Sequence[torch.Tensor]], project: bool = True):
        """A list input is stacked along the batch axis and split again (extractor.py:106-110,122-123).
        project=False returns the 128-channel trunk output (the caller applies conv2 itself)."""
        as_list = isinstance(x, (list, tuple))
        if as_list:
            nb, count = x[0].shape[0], len(x)
            x = torch.cat(list(x), dim=0)
        x = norm_act(self.norm1, self.conv1(x), True)
        x = self.layer3(self.layer2(self.layer1(x)))
        if project:
            x = self.conv2(x)
        if as_list:
            return list(torch.split(x, [nb] * count, dim=0))
        return x
