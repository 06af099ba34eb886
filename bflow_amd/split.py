"""Split-fp16 tensors and the convolution engine's host side (csrc/conv_split.hip, include/bflow_hip.h).

A `SplitTensor` is a channels-last activation held as two fp16 planes: `planes[0]` = hi, `planes[1]` = lo, each
(B, H, W, C); value = hi + lo * 2^-11 (fp32-class accuracy on the fp16 matrix cores).  This module wraps the C-ABI calls
that produce and consume them: weight packing, the implicit-GEMM convolution with its fused epilogues, InstanceNorm
statistics / normalisation + activation + residual, and conversion from / to ordinary NCHW fp32 tensors."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

from . import hip

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
GATE_NONE, GATE_ZR, GATE_BLEND, GATE_RES = 0, 1, 2, 3     # GATE_RES: split(relu(act(conv) + gate_h)), the residual block's relu(x + y)
LO_INV = 1.0 / 2048.0


def pick_tile(cout: int, m_tiles: int) -> int:
    """Output-channel tile (64 / 96 / 128).  Large problems take the widest tile with the least padding; small ones (too few
    128-pixel tiles to fill 256 CUs, e.g. the batch-1 update block) take 64 to get more workgroups."""
    cands = []
    for t in (128, 96, 64):
        ny = (cout + t - 1) // t
        cands.append((ny * t - cout, -t, t, ny * m_tiles))
    if max(c[3] for c in cands) >= 512:
        ok = [c for c in cands if c[3] >= 512]
        return min(ok)[2]
    return 64


class SplitTensor:
    """planes: (2, B, C/32, P, 32) fp16 (hi, lo); H x W image of P >= H*W pixel rows; C = logical channel count."""

    def __init__(self, planes: torch.Tensor, H: int, W: int, C: Optional[int] = None):
        assert planes.dtype == torch.float16 and planes.dim() == 5 and planes.shape[0] == 2 and planes.shape[4] == 32
        assert planes.is_contiguous() and planes.shape[3] >= H * W
        self.planes, self.H, self.W = planes, H, W
        self.C = planes.shape[2] * 32 if C is None else C

    @classmethod
    def empty(cls, B: int, H: int, W: int, C: int, device, rows: Optional[int] = None, zero: bool = False, zero_tail: bool = False) -> "SplitTensor":
        """zero: every element; zero_tail: only the pad rows [H*W, rows) -- what a producer that writes all H*W pixel rows of every channel
        block leaves undefined (one small strided fill instead of a pass over the whole tensor)."""
        rows = H * W if rows is None else rows
        alloc = torch.zeros if zero else torch.empty
        t = alloc((2, B, (C + 31) // 32, rows, 32), dtype=torch.float16, device=device)
        if zero_tail and not zero and rows > H * W:
            t[:, :, :, H * W:].zero_()
        return cls(t, H, W, C)

    @property
    def shape(self) -> Tuple[int, int, int, int]:
        return (self.planes.shape[1], self.H, self.W, self.C)   # logical (B, H, W, C)

    @property
    def rows(self) -> int:
        return self.planes.shape[3]

    @property
    def channels_padded(self) -> int:
        return self.planes.shape[2] * 32

    @property
    def hi(self) -> torch.Tensor:
        return self.planes[0]

    @property
    def lo(self) -> torch.Tensor:
        return self.planes[1]

    def to_nchw(self, c_first: int = 0, c_count: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, H, W, C = self.shape
        assert self.rows == H * W
        c_count = C - c_first if c_count is None else c_count
        if out is None:
            out = torch.empty((B, c_count, H, W), dtype=torch.float32, device=self.planes.device)
        assert out.shape[0] == B and out.shape[1] == c_count and out[0].is_contiguous()
        bs = out.stride(0) if B > 1 else c_count * H * W
        hip._check(hip.lib().bflow_split_to_nchw(self.hi.data_ptr(), self.lo.data_ptr(), out.data_ptr(), B, H * W, self.channels_padded,
                                                 c_first, c_count, bs, hip._stream()), "bflow_split_to_nchw")
        return out

    def float_nhwc(self) -> torch.Tensor:
        """(B, H, W, C) fp32 value (debug / tests)."""
        v = self.hi.float() + self.lo.float() * LO_INV            # (B, CB, P, 32)
        B, CB, P, _ = v.shape
        return v[:, :, :self.H * self.W].permute(0, 2, 1, 3).reshape(B, self.H, self.W, CB * 32)[..., :self.C]


def blocked_f32_to_nhwc(t: torch.Tensor, H: int, W: int, C: int) -> torch.Tensor:
    """(B, C/32, P, 32) blocked fp32 -> (B, H, W, C) (debug / tests)."""
    B, CB, P, _ = t.shape
    return t[:, :, :H * W].permute(0, 2, 1, 3).reshape(B, H, W, CB * 32)[..., :C]


class PackedConvWeight:
    """Conv weights pre-split for the engine: (cout_pad, KH*KW, cin_pad) fp16 hi/lo planes; rebuilt when the source changes."""

    def __init__(self):
        self._key = None
        self.planes = None

    def get(self, weight: torch.Tensor, cin_pad: Optional[int] = None, adjoint: bool = False):
        """adjoint: pack the filter of the input-gradient convolution (flipped in space, transposed in channels) straight from the forward
        weight (bflow_conv_pack_weights_adjoint): its "cout" is the forward's cin."""
        cout, cin, kh, kw = weight.shape
        if adjoint:
            cout, cin = cin, cout
        cin_pad = (cin + 31) // 32 * 32 if cin_pad is None else cin_pad
        cout_pad = (cout + 127) // 128 * 128   # any channel tile (64/96/128) may read up to 128 rows from its first row
        key = (weight.data_ptr(), hip.tensor_version(weight), cin_pad, str(weight.device), adjoint)
        if self._key != key:
            w = weight.detach().float().contiguous()
            planes = torch.empty((2, kh * kw * (cin_pad // 32), cout_pad, 32), dtype=torch.float16, device=w.device)
            fn = hip.lib().bflow_conv_pack_weights_adjoint if adjoint else hip.lib().bflow_conv_pack_weights
            hip._check(fn(hip._dev(w, name="weight"), planes[0].data_ptr(), planes[1].data_ptr(), cout, cin,
                          kh, kw, cout_pad, cin_pad, hip._stream()), "bflow_conv_pack_weights")
            self._key, self.planes, self.meta = key, planes, (cout, cin_pad, kh, kw, cout_pad)
        return self.planes, self.meta


THIN_MFMA = os.environ.get("BFLOW_NO_THIN_MFMA") is None      # A/B switch (tools/): the matrix-core form of the thin head (bflow_conv_thin_mfma_acc)


class ThinConvWeight:
    """Weights of a thin-output convolution: fp32 tap-major (KH*KW, Cout, Cin) for bflow_conv_thin_acc, and -- for a 3 x 3 filter with
    Cout <= 28 -- the derived 1 x 1 filter W'[tap * Cout + co][c] packed as a split tensor for bflow_conv_thin_mfma_acc."""

    def __init__(self):
        self._key = None
        self.w = None
        self.mfma = None

    def get(self, weight: torch.Tensor):
        key = (weight.data_ptr(), hip.tensor_version(weight), str(weight.device))
        if self._key != key:
            cout, cin, kh, kw = weight.shape
            with torch.no_grad():
                w = weight.detach().float()
                self.w = w.permute(2, 3, 0, 1).reshape(kh * kw, cout, cin).contiguous()
                self.mfma = None
                if (kh, kw) == (3, 3) and cout <= 28 and cin % 32 == 0 and cin <= 256 and weight.is_cuda:
                    taps = self.w.reshape(9 * cout, cin, 1, 1).contiguous()          # row = tap * Cout + co
                    self.mfma = PackedConvWeight().get(taps)
            self._key, self.meta = key, (cout, cin, kh, kw)
        return self.w, self.meta, self.mfma


def conv_thin_acc(x: SplitTensor, packed, bias: Optional[torch.Tensor], acc_nchw: torch.Tensor, out_split: Optional[SplitTensor] = None,
                  channel_offset: int = 0, mfma: Optional[bool] = None):
    """acc_nchw (B, cout, H, W) fp32 += conv(x, w) + bias ("same" zero padding, stride 1, cout <= 32); out_split's 32-channel block at
    `channel_offset` receives the updated values.  3 x 3 with cout <= 28 (the Bezier head up to degree 14): on the matrix cores
    (bflow_conv_thin_mfma_acc, taps as output channels); otherwise on the vector ALU in fp32 (bflow_conv_thin_acc).  mfma=False forces the latter."""
    w, (cout, cin, kh, kw) = packed[0], packed[1]
    mf = packed[2] if len(packed) > 2 else None
    B, H, W, _ = x.shape
    assert cin == x.channels_padded, f"input has {x.channels_padded} (padded) channels, weight expects {cin}"
    assert acc_nchw.dtype == torch.float32 and acc_nchw.is_contiguous() and tuple(acc_nchw.shape) == (B, cout, H, W)
    assert channel_offset % 32 == 0 or channel_offset % 32 + cout <= 32      # inside a block: the other channels of it are left alone
    oh = ol = None
    cbo = rows_o = 0
    if out_split is not None:
        assert out_split.planes.shape[1] == B and out_split.H == H and out_split.W == W and channel_offset < out_split.channels_padded
        oh, ol, cbo, rows_o = out_split.hi.data_ptr(), out_split.lo.data_ptr(), out_split.planes.shape[2], out_split.rows
    use_mfma = (THIN_MFMA if mfma is None else mfma) and mf is not None
    if use_mfma:
        planes, (_, cin_pad, _, _, cout_pad) = mf
        assert cin_pad == cin
        hip._check(hip.lib().bflow_conv_thin_mfma_acc(x.hi.data_ptr(), x.lo.data_ptr(), planes[0].data_ptr(), planes[1].data_ptr(), cout_pad,
                                                      None if bias is None else hip._dev(bias, name="bias"), acc_nchw.data_ptr(), oh, ol, B, H, W, cin,
                                                      x.rows, cout, cbo, channel_offset // 32, rows_o, channel_offset % 32, hip._stream()),
                   "bflow_conv_thin_mfma_acc")
        return
    hip._check(hip.lib().bflow_conv_thin_acc(x.hi.data_ptr(), x.lo.data_ptr(), hip._dev(w, name="weight"),
                                             None if bias is None else hip._dev(bias, name="bias"), acc_nchw.data_ptr(), oh, ol, B, H, W, cin,
                                             x.rows, cout, kh, kw, cbo, channel_offset // 32, rows_o, channel_offset % 32, hip._stream()), "bflow_conv_thin_acc")


def _stats_replicas(stats: Optional[torch.Tensor], B: int, C: int) -> int:
    """Statistics tables are (B, C, 2) or (R, B, C, 2): R replicas that spread the epilogue's fp64 atomics (summed by norm_act)."""
    if stats is None:
        return 0
    assert stats.is_contiguous() and tuple(stats.shape[-3:]) == (B, C, 2) and stats.dim() in (3, 4), tuple(stats.shape)
    return stats.shape[0] if stats.dim() == 4 else 1


STEM_LAYOUT = 0 if os.environ.get("BFLOW_STEM_IM2COL") else 1     # bflow_stem_desc_t.layout: 1 = row windows (round 3), 0 = im2col tiles (A/B)


class PackedStemWeight:
    """The 7x7 stem filter (Cout, Cin, 7, 7) laid out for bflow_conv_stem as a 1x1 filter over K columns in chunks of min(Cin, 8)
    channels, packed like any conv weight.  layout 0: K = (chunk, c_local, r, q), every chunk zero padded to a multiple of 32;
    layout 1 (row windows): K = (chunk, r, q * chunk + c_local), every filter row zero padded to a multiple of 16 and every chunk's seven
    rows to a multiple of 32."""

    def __init__(self, layout: Optional[int] = None):
        self._key = None
        self.layout = STEM_LAYOUT if layout is None else layout

    def get(self, weight: torch.Tensor):
        cout, cin, kh, kw = weight.shape
        assert kh == kw == 7
        key = (weight.data_ptr(), hip.tensor_version(weight), str(weight.device))
        if self._key != key:
            chunk = min(cin, 8)
            kpc = (chunk * kh * kw + 31) // 32 * 32
            cols = []
            w = weight.detach().float()
            for c0 in range(0, cin, chunk):
                if self.layout >= 1:
                    blk = torch.nn.functional.pad(w[:, c0:c0 + chunk], (0, 0, 0, 0, 0, chunk - w[:, c0:c0 + chunk].shape[1]))   # missing channels: zero
                    rows = blk.permute(0, 2, 3, 1).reshape(cout, kh, kw * chunk)               # (r, q * chunk + c_local)
                    rows = torch.nn.functional.pad(rows, (0, (kw * chunk + 15) // 16 * 16 - kw * chunk)).reshape(cout, -1)
                    cols.append(torch.nn.functional.pad(rows, (0, (rows.shape[1] + 31) // 32 * 32 - rows.shape[1])))
                    continue
                blk = w[:, c0:c0 + chunk].reshape(cout, -1)                      # (c_local, r, q), c_local slowest
                cols.append(torch.nn.functional.pad(blk, (0, kpc - blk.shape[1])))
            wm = torch.cat(cols, dim=1).contiguous().view(cout, -1, 1, 1)
            self._inner = PackedConvWeight()
            planes, meta = self._inner.get(wm)
            self._key, self.planes, self.meta = key, planes, (cout, cin, planes.shape[1], meta[4])   # (cout, cin, k_blocks, cout_pad)
        return self.planes, self.meta, self.layout


class ChannelWindows:
    """A stacked batch of channel windows of one source tensor, never materialised: image n = source image n % B_src, channels
    [starts[n // B_src], + width) -- what RAFTSpline.gen_voxel_grids + torch.cat(dim=0) build (raft.py:88-99,121)."""

    def __init__(self, source: torch.Tensor, starts, width: int):
        assert source.dim() == 4 and source.dtype == torch.float32 and source.is_contiguous()
        assert 1 <= len(starts) <= 8 and all(0 <= s_ and s_ + width <= source.shape[1] for s_ in starts)
        self.source, self.starts, self.width = source, [int(s_) for s_ in starts], width
        self.shape = (len(starts) * source.shape[0], width, source.shape[2], source.shape[3])
        self.device = source.device

    def materialize(self) -> torch.Tensor:
        return torch.cat([self.source[:, s_:s_ + self.width] for s_ in self.starts], dim=0)


class StemInput:
    """The stem convolution's input assembled IN the kernel's load (bflow_stem_desc_t.window_bases / x2 / *_dtype / *_image_norm) instead of by
    torch launches: image n = window group n // B, source image n % B; its channels are [window channels | extra channels]:
      windows : list of (tensor (B, C_src, H, W), first channel) -- the groups, stacked along the batch axis (extractor.py:106-110's torch.cat);
      width   : channels a window supplies;
      extra   : optional tensor (B, C_x, H, W) whose channels follow the window's (raft.py:137-140: cat((context_grid, img0)));
      norm / extra_norm : 2 * (v / 255) - 1 on that part (raft.py:134).
    Tensors are fp32 or uint8, contiguous, never copied or converted."""

    def __init__(self, windows, width: int, extra: Optional[torch.Tensor] = None, norm: bool = False, extra_norm: bool = False):
        assert 1 <= len(windows) <= 8
        t0 = windows[0][0]
        B, C_src, H, W = t0.shape
        for t, st in windows:
            assert t.dim() == 4 and tuple(t.shape) == (B, C_src, H, W) and t.dtype == t0.dtype and t.is_contiguous() and t.device == t0.device
            assert 0 <= st and st + width <= C_src
        assert t0.dtype in (torch.float32, torch.uint8)
        if extra is not None:
            assert extra.dim() == 4 and extra.shape[0] == B and tuple(extra.shape[2:]) == (H, W) and extra.is_contiguous()
            assert extra.dtype in (torch.float32, torch.uint8) and extra.device == t0.device
        self.windows, self.width, self.extra, self.norm, self.extra_norm = [(t, int(st)) for t, st in windows], int(width), extra, bool(norm), bool(extra_norm)
        self.shape = (len(windows) * B, width + (0 if extra is None else extra.shape[1]), H, W)
        self.device = t0.device

    def materialize(self) -> torch.Tensor:
        """The tensor the reference builds with torch ops (tests).  The normalisation runs on the CPU like the reference path does: there
        `x / 255` is a true fp32 division (torch's GPU kernel multiplies by the reciprocal of a scalar divisor, one ulp off for some values)."""
        def nrm(t, on):
            t = t.float()
            return (2 * (t.cpu() / 255) - 1).to(t.device) if on else t
        parts = []
        for t, st in self.windows:
            w = nrm(t[:, st:st + self.width], self.norm)
            parts.append(w if self.extra is None else torch.cat((w, nrm(self.extra, self.extra_norm)), dim=1))
        return torch.cat(parts, dim=0)


def conv_stem(x, packed, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, act: int = ACT_NONE,
              stats: Optional[torch.Tensor] = None, want_split: bool = True, want_f32: bool = False, layout: Optional[int] = None):
    """7x7 / stride 2 / pad 3 convolution of a few-channel fp32 NCHW tensor or ChannelWindows (BasicEncoder.conv1) -> (split_out or
    None, blocked fp32 or None), epilogue as `conv`.  `packed` = PackedStemWeight.get(weight).  `layout` 2 / 3 forces the persistent /
    the per-patch form of the row-window kernel (tests, A/B; same packed filter as layout 1)."""
    planes, (cout, cin, k_blocks, cout_pad) = packed[0], packed[1]
    windows = x if isinstance(x, ChannelWindows) else None
    general = x if isinstance(x, StemInput) else None
    B, C, H, W = x.shape
    if windows is not None:
        x = windows.source
    if general is not None:
        x = general.windows[0][0]
        assert C == cin
    else:
        assert C == cin and x.dtype == torch.float32 and x.is_contiguous()
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    dev = x.device
    out_split = SplitTensor.empty(B, Ho, Wo, cout, dev) if want_split else None
    out_f32 = torch.empty((B, (cout + 31) // 32, Ho * Wo, 32), dtype=torch.float32, device=dev) if want_f32 else None
    d = hip.StemDesc()
    d.x, d.w_hi, d.w_lo = (hip._dev(x, name="x") if general is None else x.data_ptr()), planes[0].data_ptr(), planes[1].data_ptr()
    d.B, d.Cin, d.H, d.W, d.Cout, d.cout_pad, d.k_blocks = B, C, H, W, cout, cout_pad, k_blocks
    d.ksize, d.stride, d.pad = 7, 2, 3
    d.layout = STEM_LAYOUT if len(packed) < 3 else packed[2]
    if layout is not None:
        assert d.layout >= 1 and layout in (1, 2, 3), "forcing a form of the row-window kernel needs row-window weights"
        d.layout = layout
    d.out_f32 = None if out_f32 is None else out_f32.data_ptr()
    d.out_hi = None if out_split is None else out_split.hi.data_ptr()
    d.out_lo = None if out_split is None else out_split.lo.data_ptr()
    d.scale = None if scale is None else hip._dev(scale, name="scale")
    d.shift = None if shift is None else hip._dev(shift, name="shift")
    d.act = act
    d.stats = None if stats is None else hip._dev(stats, torch.float64, "stats")
    d.stats_replicas = _stats_replicas(stats, B, cout)
    if windows is not None:
        d.n_windows, d.src_channels = len(windows.starts), x.shape[1]
        starts = (ctypes.c_int * len(windows.starts))(*windows.starts)
        d.window_starts = starts
    if general is not None:
        nw = len(general.windows)
        d.n_windows, d.src_channels = nw, x.shape[1]
        starts = (ctypes.c_int * nw)(*[st for _, st in general.windows])
        bases = (ctypes.c_void_p * nw)(*[t.data_ptr() for t, _ in general.windows])
        d.window_starts, d.window_bases = starts, bases
        d.x_dtype, d.x_image_norm = int(x.dtype == torch.uint8), int(general.norm)
        if general.extra is not None:
            d.x2, d.x2_channels = general.extra.data_ptr(), general.extra.shape[1]
            d.x2_dtype, d.x2_image_norm = int(general.extra.dtype == torch.uint8), int(general.extra_norm)
    hip._check(hip.lib().bflow_conv_stem(ctypes.byref(d), hip._stream()), "bflow_conv_stem")
    return out_split, out_f32


def conv(x: SplitTensor, packed, stride: int = 1, padding=(0, 0), scale: Optional[torch.Tensor] = None,
         shift: Optional[torch.Tensor] = None, act: int = ACT_NONE, out_split: Optional[SplitTensor] = None,
         out_f32: Optional[torch.Tensor] = None, channel_offset: int = 0, stats: Optional[torch.Tensor] = None,
         want_split: bool = True, want_f32: bool = False, out_rows: Optional[int] = None, zero_rows: bool = False,
         x2: Optional[SplitTensor] = None, addend: Optional[torch.Tensor] = None, tile: Optional[int] = None,
         gate: int = GATE_NONE, gate_h: Optional[SplitTensor] = None, gate_z: Optional[torch.Tensor] = None,
         acc_nchw: Optional[torch.Tensor] = None, weight_sets: int = 1, keep_pad: bool = False, _desc_only: bool = False):
    """Implicit-GEMM convolution.  `packed` = PackedConvWeight.get(weight).  Returns (split_out or None, f32_out or None);
    f32_out is blocked fp32 (B, Cs/32, P_out, 32).  When out_* buffers are given the result is written at channel
    `channel_offset` (a multiple of 32) of their channel dimension (free concatenation).  out_rows > Ho*Wo allocates
    zero-filled tail rows (K5's 128-row operand padding).
    acc_nchw (B, cout, Ho, Wo) fp32: accumulated in place (+= result), outputs receive the updated value.
    weight_sets S > 1: `packed` holds S filters back to back, image b uses filter b % S (weight-gradient GEMMs, conv_train.py).
    gate = GATE_ZR / GATE_BLEND fuses the SepConvGRU element-wise stage (update.py:38-47) into the epilogue: see bflow_conv_desc_t."""
    planes, (cout, cin_pad, kh, kw, cout_pad) = packed
    B, H, W, _ = x.shape
    c_in = x.channels_padded + (0 if x2 is None else x2.channels_padded)
    assert c_in == cin_pad, f"input has {c_in} (padded) channels, packed weight expects {cin_pad}"
    assert x2 is None or (x2.rows == x.rows and x2.H == H and x2.W == W)
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    Ho, Wo = (H + 2 * ph - kh) // stride + 1, (W + 2 * pw - kw) // stride + 1
    dev = x.planes.device
    rows = Ho * Wo if out_rows is None else out_rows
    if tile is None:
        tile = pick_tile(cout, B * ((Ho * Wo + 127) // 128))
    if out_split is None and want_split:
        out_split = SplitTensor.empty(B, Ho, Wo, cout, dev, rows=rows, zero_tail=zero_rows or rows != Ho * Wo)     # the kernel writes every pixel row (pad channels as zeros)
    if out_f32 is None and want_f32:
        out_f32 = torch.empty((B, (cout + 31) // 32, rows, 32), dtype=torch.float32, device=dev)
    cs = None
    for o in (out_split.planes[0] if out_split is not None else None, out_f32):
        if o is not None:
            assert o.shape[0] == B and o.shape[2] == rows and o.shape[3] == 32 and o.is_contiguous()
            assert cs is None or cs == o.shape[1] * 32
            cs = o.shape[1] * 32
    assert cs is not None and channel_offset % 32 == 0
    if gate == GATE_NONE:
        assert channel_offset + cout <= cs
    else:
        ch = cout // 2 if gate == GATE_ZR else cout
        assert cs == ch and channel_offset == 0 and out_split is not None and gate_h is not None and gate_h.planes.shape == out_split.planes.shape
        assert gate != GATE_ZR or (out_f32 is not None and tuple(out_f32.shape) == (B, ch // 32, rows, 32))
        assert gate != GATE_BLEND or (gate_z is not None and tuple(gate_z.shape) == (B, ch // 32, rows, 32) and gate_z.dtype == torch.float32)
    d = hip.ConvDesc()
    d.x_hi, d.x_lo, d.w_hi, d.w_lo = x.hi.data_ptr(), x.lo.data_ptr(), planes[0].data_ptr(), planes[1].data_ptr()
    d.B, d.H, d.W, d.C, d.Cout, d.cout_pad = B, H, W, cin_pad, cout, cout_pad
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.tile_n = kh, kw, stride, ph, pw, tile
    d.out_f32 = None if out_f32 is None else out_f32.data_ptr()
    d.out_hi = None if out_split is None else out_split.hi.data_ptr()
    d.out_lo = None if out_split is None else out_split.lo.data_ptr()
    d.out_channel_stride, d.out_channel_offset, d.out_rows_per_image, d.in_rows_per_image = cs, channel_offset, rows, x.rows
    if x2 is not None:
        d.x2_hi, d.x2_lo, d.x_split_channels = x2.hi.data_ptr(), x2.lo.data_ptr(), x.channels_padded
    if addend is not None:
        assert addend.dtype == torch.float32 and addend.is_contiguous() and tuple(addend.shape) == (B, (cout + 31) // 32, rows, 32)
        d.addend = addend.data_ptr()
    d.scale = None if scale is None else hip._dev(scale, name="scale")
    d.shift = None if shift is None else hip._dev(shift, name="shift")
    d.act = act
    d.stats = None if stats is None else hip._dev(stats, torch.float64, "stats")
    d.stats_replicas = _stats_replicas(stats, B, cout)
    if acc_nchw is not None:
        assert acc_nchw.dtype == torch.float32 and acc_nchw.is_contiguous() and tuple(acc_nchw.shape) == (B, cout, Ho, Wo) and stats is None
        d.acc_nchw = acc_nchw.data_ptr()
    if gate != GATE_NONE:
        d.gate, d.gate_h_hi, d.gate_h_lo = gate, gate_h.hi.data_ptr(), gate_h.lo.data_ptr()
        d.gate_z = None if gate_z is None else gate_z.data_ptr()
    d.weight_sets = weight_sets
    d.keep_pad_channels = int(keep_pad)     # the pad channels of the last output block belong to another producer: do not zero them
    if _desc_only:                          # conv_pair: the resolved descriptor instead of a launch
        return d, out_split, out_f32
    hip._check(hip.lib().bflow_conv_split(ctypes.byref(d), hip._stream()), "bflow_conv_split")
    return out_split, out_f32


def conv_pair(first: dict, second: dict):
    """Two INDEPENDENT convolutions (keyword arguments of `conv` each; neither reads the other's output) through bflow_conv_split_pair: ONE
    launch when both resolve to the same small-grid kernel (the batch-1 motion encoder's convc1 | convf1 and convc2 | convf2), else two.
    Returns ((split, f32) of the first, (split, f32) of the second, fused: bool); results bit-identical to two `conv` calls."""
    d0, s0, f0 = conv(**first, _desc_only=True)
    d1, s1, f1 = conv(**second, _desc_only=True)
    fused = ctypes.c_int(0)
    hip._check(hip.lib().bflow_conv_split_pair(ctypes.byref(d0), ctypes.byref(d1), ctypes.byref(fused), hip._stream()), "bflow_conv_split_pair")
    return (s0, f0), (s1, f1), bool(fused.value)


def conv_norm_in(raw: torch.Tensor, shape_bhwc, x_stats: torch.Tensor, packed, stats: Optional[torch.Tensor] = None, eps: float = 1e-5,
                 out_f32: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 stride-1 "same" convolution of relu(instance_norm(raw)) where `raw` is the previous convolution's blocked fp32 output
    (B, C/32, H*W, 32) and `x_stats` (R, B, C, 2) the statistics its epilogue accumulated (extractor.py:47-48: relu(norm1(conv1(x)))
    feeding conv2): the normalisation is applied while the halo is staged (bflow_conv_desc_t.x_raw), bit-identical to
    norm_act(...) followed by conv(...), one pass over the activation less.  Returns the blocked fp32 output (+= statistics into `stats`)."""
    planes, (cout, cin_pad, kh, kw, cout_pad) = packed
    B, H, W, C = shape_bhwc
    assert (kh, kw) == (3, 3) and C == cin_pad and C % 32 == 0 and C <= 128 and tuple(raw.shape) == (B, C // 32, H * W, 32) and raw.dtype == torch.float32
    if out_f32 is None:
        out_f32 = torch.empty((B, (cout + 31) // 32, H * W, 32), dtype=torch.float32, device=raw.device)
    d = hip.ConvDesc()
    d.x_raw, d.x_stats, d.x_eps = hip._dev(raw, name="raw"), hip._dev(x_stats, torch.float64, "x_stats"), eps
    d.x_stats_replicas = _stats_replicas(x_stats, B, C)
    d.w_hi, d.w_lo = planes[0].data_ptr(), planes[1].data_ptr()
    d.B, d.H, d.W, d.C, d.Cout, d.cout_pad = B, H, W, C, cout, cout_pad
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.tile_n = 3, 3, 1, 1, 1, 64
    d.out_f32 = out_f32.data_ptr()
    d.out_channel_stride, d.out_channel_offset, d.out_rows_per_image, d.in_rows_per_image = (cout + 31) // 32 * 32, 0, H * W, H * W
    d.stats = None if stats is None else hip._dev(stats, torch.float64, "stats")
    d.stats_replicas = _stats_replicas(stats, B, cout)
    hip._check(hip.lib().bflow_conv_split(ctypes.byref(d), hip._stream()), "bflow_conv_split")
    return out_f32


def wgrad_pack(src: torch.Tensor, out_hw, ksize=(1, 1), stride: int = 1, padding=(0, 0), rows: Optional[int] = None,
               k_blocks: Optional[int] = None, scale: Optional[torch.Tensor] = None, taps_in_rows: bool = False) -> torch.Tensor:
    """NCHW fp32 -> split planes with the pixel index k = (b*Ho + yo)*Wo + xo in the block position (bflow_wgrad_pack):
    (2, KH*KW, k_blocks, rows, 32), or with taps_in_rows (2, k_blocks, rows, 32) where row n = tap*C + c."""
    B, C, H, W = src.shape
    Ho, Wo = out_hw
    kh, kw = ksize
    ph, pw = padding
    rows = (kh * kw * C if taps_in_rows else C) if rows is None else rows
    kb = (B * Ho * Wo + 31) // 32 if k_blocks is None else k_blocks
    shape = (2, kb, rows, 32) if taps_in_rows else (2, kh * kw, kb, rows, 32)
    out = torch.empty(shape, dtype=torch.float16, device=src.device)
    hip._check(hip.lib().bflow_wgrad_pack(hip._dev(src, name="src"), out[0].data_ptr(), out[1].data_ptr(), B, C, H, W, Ho, Wo, kh, kw, stride, ph, pw,
                                          rows, kb, int(taps_in_rows), None if scale is None else scale.data_ptr(), hip._stream()), "bflow_wgrad_pack")
    return out


def blocked_f32_to_nchw(x: torch.Tensor, C: int, H: int, W: int, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Blocked fp32 engine output (B, C/32 blocks, rows >= H*W, 32) -> (B, C, H, W) fp32, optionally times a device scalar."""
    B, cb, rows, _ = x.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    hip._check(hip.lib().bflow_blocked_f32_to_nchw(hip._dev(x, name="x"), out.data_ptr(), B, H * W, C, cb, rows,
                                                   None if scale is None else scale.data_ptr(), hip._stream()), "bflow_blocked_f32_to_nchw")
    return out


def wgrad_reduce(part: torch.Tensor, G: int, cout: int, cin: int, ksize, orientation: int, inv: Optional[torch.Tensor]) -> torch.Tensor:
    """Sum of the k-chunk partial results -> (cout, cin, kh, kw) fp32 times the device scalar `inv` (bflow_wgrad_reduce)."""
    kh, kw = ksize
    blocks, rows = part.shape[-3], part.shape[-2]
    dw = torch.empty((cout, cin, kh, kw), dtype=torch.float32, device=part.device)
    hip._check(hip.lib().bflow_wgrad_reduce(hip._dev(part, name="part"), dw.data_ptr(), G, cout, cin, kh * kw, blocks, rows, orientation,
                                            None if inv is None else inv.data_ptr(), hip._stream()), "bflow_wgrad_reduce")
    return dw


_wgrad_acc = {}


def conv_wgrad_halo(xs: "SplitTensor", gs: "SplitTensor", cout: int, cin: int, ksize, inv_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dw (cout, cin, KH, KW) fp32 = inv_scale * sum over pixels of dY (gs, blocked split, pre-scaled) x shifted X (xs): bflow_conv_wgrad_halo
    into a persistent fp32 accumulator (one per shape, device and stream; zero between calls) + bflow_conv_wgrad_finish (scale, re-order
    to the filter layout, re-zero the accumulator)."""
    kh, kw = ksize
    B, H, W, _ = xs.shape
    assert gs.shape[:3] == (B, H, W) and xs.rows == gs.rows
    dev = xs.planes.device
    key = (kh * kw, (cout + 63) // 64 * 64, xs.channels_padded, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    acc = _wgrad_acc.pop(key, None)              # popped while in use: an exception in between does not leave a dirty accumulator behind
    if acc is None:
        acc = torch.zeros(key[:3], dtype=torch.float32, device=dev)
    dw = torch.empty((cout, cin, kh, kw), dtype=torch.float32, device=dev)
    hip._check(hip.lib().bflow_conv_wgrad_halo(xs.hi.data_ptr(), xs.lo.data_ptr(), gs.hi.data_ptr(), gs.lo.data_ptr(), acc.data_ptr(), B, H, W,
                                               xs.channels_padded, cout, xs.rows, kh, kw, hip._stream()), "bflow_conv_wgrad_halo")
    hip._check(hip.lib().bflow_conv_wgrad_finish(acc.data_ptr(), dw.data_ptr(), kh * kw, cout, cin, xs.channels_padded,
                                                 None if inv_scale is None else inv_scale.data_ptr(), hip._stream()), "bflow_conv_wgrad_finish")
    _wgrad_acc[key] = acc
    return dw


_pow2_work = {}


def pow2_scale(x: torch.Tensor, target: float) -> torch.Tensor:
    """(2,) fp32 device tensor {s, 1/s}, s = 2^floor(log2(target / max|x|)) -- computed on the device (bflow_pow2_scale)."""
    key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
    if key not in _pow2_work:
        _pow2_work[key] = torch.zeros(2, dtype=torch.int32, device=x.device)
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    hip._check(hip.lib().bflow_pow2_scale(hip._dev(x, name="x"), x.numel(), float(target), out.data_ptr(), _pow2_work[key].data_ptr(), hip._stream()),
               "bflow_pow2_scale")
    return out


def grad_stats(dy: torch.Tensor, target: float):
    """({s, 1/s, s x C}, dbias): the power-of-two scale of `pow2_scale` (and C copies of it: `from_nchw(dy, scale_vec=out[2:])`) and the
    per-channel sum of an NCHW gradient in one pass (bflow_grad_stats)."""
    B, C, H, W = dy.shape
    out = torch.empty(2 + C, dtype=torch.float32, device=dy.device)
    db = torch.empty(C, dtype=torch.float32, device=dy.device)
    partial = torch.empty(B * C * ((H * W + 1023) // 1024) + 1024, dtype=torch.float32, device=dy.device)
    hip._check(hip.lib().bflow_grad_stats(hip._dev(dy, name="dy"), B, C, H * W, float(target), out.data_ptr(), partial.data_ptr(), db.data_ptr(),
                                          hip._stream()), "bflow_grad_stats")
    return out, db


def plane_stats(x_nchw: torch.Tensor) -> torch.Tensor:
    """(B, C, 2) fp64 (sum, sum of squares) per plane of an NCHW fp32 tensor."""
    B, C, H, W = x_nchw.shape
    stats = torch.empty((B, C, 2), dtype=torch.float64, device=x_nchw.device)
    hip._check(hip.lib().bflow_plane_stats(hip._dev(x_nchw, name="x"), stats.data_ptr(), B * C, H * W, hip._stream()), "bflow_plane_stats")
    return stats


def norm_act(a: torch.Tensor, shape_bhwc, a_is_nchw: bool = False, stats_a: Optional[torch.Tensor] = None,
             scale_a: Optional[torch.Tensor] = None, shift_a: Optional[torch.Tensor] = None, act_a: int = ACT_NONE,
             b: Optional[torch.Tensor] = None, stats_b: Optional[torch.Tensor] = None, res: Optional[SplitTensor] = None,
             act_out: int = ACT_NONE, eps: float = 1e-5, out: Optional[SplitTensor] = None, out_f32: Optional[torch.Tensor] = None,
             want_split: bool = True, act_b: int = ACT_NONE) -> Tuple[Optional[SplitTensor], Optional[torch.Tensor]]:
    """out = act_out(res + act_a(norm_a(a)) + act_b(norm_b(b)))  (see bflow_norm_act_split).  a / b: blocked fp32 (B, C/32, H*W, 32), or a: NCHW."""
    B, H, W, C = shape_bhwc
    if out is None and want_split:
        out = SplitTensor.empty(B, H, W, C, a.device)
    d = hip.NormDesc()
    d.a = hip._dev(a, name="a")
    d.stats_a = None if stats_a is None else hip._dev(stats_a, torch.float64, "stats_a")
    d.scale_a = None if scale_a is None else hip._dev(scale_a, name="scale_a")
    d.shift_a = None if shift_a is None else hip._dev(shift_a, name="shift_a")
    d.a_is_nchw, d.act_a = int(a_is_nchw), act_a
    d.b = None if b is None else hip._dev(b, name="b")
    d.stats_b = None if stats_b is None else hip._dev(stats_b, torch.float64, "stats_b")
    d.res_hi = None if res is None else res.hi.data_ptr()
    d.res_lo = None if res is None else res.lo.data_ptr()
    assert res is None or res.rows == H * W
    d.act_out = act_out
    d.out_hi = None if out is None else out.hi.data_ptr()
    d.out_lo = None if out is None else out.lo.data_ptr()
    assert out is None or out.rows == H * W
    d.out_f32 = None if out_f32 is None else hip._dev(out_f32, name="out_f32")
    d.B, d.HW, d.C, d.eps, d.rows_per_image = B, H * W, C, eps, H * W
    ra, rb = _stats_replicas(stats_a, B, C), _stats_replicas(stats_b, B, C)
    assert not (ra and rb) or ra == rb, "both statistics tables must use the same number of replicas"
    d.stats_replicas = max(ra, rb)
    d.act_b = act_b
    hip._check(hip.lib().bflow_norm_act_split(ctypes.byref(d), hip._stream()), "bflow_norm_act_split")
    return out, out_f32


_zeros_c = {}


def from_nchw(x: torch.Tensor, scale: Optional[torch.Tensor] = None, scale_vec: Optional[torch.Tensor] = None) -> SplitTensor:
    """NCHW fp32 -> blocked split (identity transform, or times a 1-element device tensor `scale`, or times the (C,) device vector
    `scale_vec`); channels are zero-padded to the next multiple of 32."""
    B, C, H, W = x.shape
    if scale is None and scale_vec is None:
        out, _ = norm_act(x.float().contiguous(), (B, H, W, C), a_is_nchw=True)
    else:
        zkey = (C, x.device.index)
        if zkey not in _zeros_c:
            _zeros_c[zkey] = torch.zeros(C, dtype=torch.float32, device=x.device)
        if scale_vec is None:
            scale_vec = scale.reshape(1).expand(C).contiguous()
        assert scale_vec.shape == (C,) and scale_vec.is_contiguous()
        out, _ = norm_act(x.float().contiguous(), (B, H, W, C), a_is_nchw=True, scale_a=scale_vec, shift_a=_zeros_c[zkey])
    return out


def from_rows(x: torch.Tensor, scale: Optional[torch.Tensor] = None) -> SplitTensor:
    """(R, P, C) fp32 row-major [pixel][channel] -> SplitTensor of R images with P pixel rows (H = 1, W = P) and C channels, times the
    1-element device tensor `scale` (bflow_rows_to_split)."""
    R, P, C = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = SplitTensor.empty(R, 1, P, C, x.device)
    hip._check(hip.lib().bflow_rows_to_split(hip._dev(x, name="x"), out.hi.data_ptr(), out.lo.data_ptr(), R, P, C,
                                             None if scale is None else scale.data_ptr(), hip._stream()), "bflow_rows_to_split")
    return out


def bezier_update(params: torch.Tensor, delta: Optional[torch.Tensor], dst: SplitTensor, dst_block: int,
                  dst2: Optional[SplitTensor] = None, dst2_block: int = 0, channel_in_block: int = 0):
    """params (B, 2deg, h, w) fp32 += delta (blocked fp32, first 2deg channels); re-emit as split channel block(s), or (channel_in_block
    > 0) as channels [channel_in_block, + 2deg) of block `dst_block`, leaving the rest of that block alone."""
    B, C2 = params.shape[:2]
    P = params.shape[2] * params.shape[3]
    assert dst.rows == P and (dst2 is None or dst2.rows == P)
    hip._check(hip.lib().bflow_bezier_update(hip._dev(params, name="params"), None if delta is None else hip._dev(delta, name="delta"), C2,
                                             dst.hi.data_ptr(), dst.lo.data_ptr(), dst.planes.shape[2], dst_block,
                                             None if dst2 is None else dst2.hi.data_ptr(), None if dst2 is None else dst2.lo.data_ptr(),
                                             0 if dst2 is None else dst2.planes.shape[2], dst2_block, B, P, channel_in_block, hip._stream()), "bflow_bezier_update")


def im2col_small(x: torch.Tensor, kh: int, kw: int, padding, out: Optional[SplitTensor] = None) -> SplitTensor:
    """x (B, C, H, W) fp32 -> blocked split (B, H, W, kh*kw*C) with K index = tap*C + c."""
    B, C, H, W = x.shape
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    if out is None:
        out = SplitTensor.empty(B, H, W, kh * kw * C, x.device)
    hip._check(hip.lib().bflow_im2col_small(hip._dev(x, name="x"), out.hi.data_ptr(), out.lo.data_ptr(), B, C, H, W, kh, kw, ph, pw, out.rows,
                                            hip._stream()), "bflow_im2col_small")
    return out
