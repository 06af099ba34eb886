"""Correlation volume, pyramid and look-up on the HIP kernels -- drop-in for models/raft_utils/corr.py:127-351.

`CorrComputation` validates/collects the feature maps exactly like the reference (corr.py:128-185,223-227);
`CorrBlockParallelMultiTarget` builds the (T, B, N, N) volume with the fp32-MFMA kernel (K5), the per-target pyramid
with the pooling kernel (K6), and answers look-ups with the LDS-staged gather kernel (K7).  Layouts:

    level 0 : one tensor (T, B*N, h, w)            == reference CorrData.corr (T, B*N, 1, h, w) without the unit dim
    level L : one tensor (T_L, B*N, h>>L, w>>L)    for the targets with num_levels > L (floor on odd sizes)

Everything is allocated with torch (caller-owned memory); the kernels only see pointers.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import hip

# Correlation arithmetic (all hand-written HIP; override the default with BFLOW_CORR_PRECISION or per block / per model):
#   "split" : split-fp16 MFMA engine, fp32 volume -- fp32-class accuracy (~2^-22 per product); the default on row-major planes
#   "f32"   : exact fp32 MFMA, fp32 volume (row-major planes only)
#   "f16"   : plain fp16 operands, fp16 tiled volume (BASELINE configs[4]: "fp16 MFMA correlation"): half the volume bytes, a third of the
#             matrix-core work, fp16 accuracy (2^-11 per operand and stored value) -- measured EPE vs the fp32 oracle: tests/, DESIGN.md
#   "split8": (the default on tiled planes with feature dim 128 / 256 -- `default_precision`) hi*hi on the fp16 rate + BOTH cross terms (hi*lo + lo*hi) of a 32-channel block in one fp8 (e4m3) K = 64 MFMA, fp32 tiled volume:
#             two matrix-pipe units per product instead of three and a higher sustained clock (the matrix cores are power-limited on real
#             data); the cross terms carry 2^-11 of a product, so their 2^-4 operand rounding leaves ~2^-16 per product
#   "split/h", "split8/h", "f16/w": the same arithmetics with the other storage type (fp16 / fp16 / fp32 volume) -- the decomposition of the
#             fp16 variant's error into operand and storage rounding (tests/test_hip_parity.py), not product defaults
# name -> (bflow_corr_build_tiled arithmetic, fp16 volume)
PRECISIONS = {"split": (hip.ARITH_SPLIT, False), "f32": (None, False), "f16": (hip.ARITH_F16, True), "split8": (hip.ARITH_SPLIT8, False),
              "split/h": (hip.ARITH_SPLIT, True), "split8/h": (hip.ARITH_SPLIT8, True), "f16/w": (hip.ARITH_F16, False)}
TILED_ONLY = frozenset(("f16", "split8", "split/h", "split8/h", "f16/w"))      # written by bflow_corr_build_tiled only


def default_precision(dim: int, tiled: bool = True) -> str:
    """THE resolver of the correlation arithmetic when none is named (model, CorrBlockParallelMultiTarget and CorrComputation all ask here):
    BFLOW_CORR_PRECISION when set (read at call time, validated against PRECISIONS up front), else "split8" where it exists -- tiled planes,
    feature dim 128 / 256 -- and "split" otherwise (row-major planes: the reference-shaped API and the training path)."""
    env = os.environ.get("BFLOW_CORR_PRECISION")
    if env is not None:
        check_precision(env)
        return env
    return "split8" if tiled and dim in (128, 256) else "split"


def check_precision(name: str) -> str:
    if name not in PRECISIONS:
        raise ValueError(f"correlation precision {name!r}: expected one of {sorted(PRECISIONS)}")
    return name

# Level 1 of the pyramid written by the K5 launch itself (bflow_corr_build_tiled pool_out).  Built, bit-identical to the separate pooling
# pass (tests) and NEUTRAL in frames/s (271.0 / 273.3 / 274.6 vs 271.0 / 273.4 / 276.0 over three A/B pairs): the fused launch takes 122 us
# against 92 us + 24.7 us for the pooling pass -- the two DPP exchanges, the selects and 16 more (partial-line) stores per chunk cost the
# panels of the pooled target ~40 %, and those panels sit on two XCDs, which then set the kernel's time.  Off by default, so that the
# product's K5 launch is the plain streaming kernel; BFLOW_FUSED_POOL=1 (or corr.FUSE_POOL1 = True) enables it.
FUSE_POOL1 = os.environ.get("BFLOW_FUSED_POOL") is not None


def _x8_planes(p1: torch.Tensor, p2: torch.Tensor):
    """x8 planes of the two packed operands of one reference group; one launch when they are neighbouring slices of one tensor."""
    n1 = p1[0].numel()
    if (p1[0].data_ptr() + 2 * n1 == p2[0].data_ptr() and p1[1].data_ptr() + 2 * n1 == p2[1].data_ptr()
            and p1._base is not None and p1._base is p2._base and p1.shape[2:] == p2.shape[2:]):
        both = torch.empty((p1.shape[1] + p2.shape[1],) + tuple(p1.shape[2:-1]) + (64,), dtype=torch.uint8, device=p1.device)
        hip._check(hip.lib().bflow_split_to_x8(p1[0].data_ptr(), p1[1].data_ptr(), both.data_ptr(), (n1 + p2[0].numel()) // 32, hip._stream()),
                   "bflow_split_to_x8")
        return both[:p1.shape[1]], both[p1.shape[1]:]
    return hip.split_to_x8(p1), hip.split_to_x8(p2)

_LIST_TYPES: Tuple[type, ...] = (list, tuple)
try:  # omegaconf is optional (the reference passes ListConfig objects when driven by Hydra, corr.py:8,147)
    from omegaconf import ListConfig as _ListConfig  # type: ignore
    _LIST_TYPES = (list, tuple, _ListConfig)
except Exception:  # pragma: no cover
    pass


class _PackedShape:
    """Stand-in for an fmap2 tensor of a packed reference group: only its target count is ever asked for."""

    def __init__(self, targets: int):
        self.shape = (targets,)


class CorrComputation:
    def __init__(self,
                 fmap1: Union[torch.Tensor, List[torch.Tensor]],
                 fmap2: Union[torch.Tensor, List[torch.Tensor]],
                 num_levels_per_target: Union[int, List[int], List[torch.Tensor]]):
        """fmap1: (B,D,h,w) or list of those; fmap2: (B,D,h,w) | (T,B,D,h,w) or list of (T,B,D,h,w);
        num_levels_per_target: int | list[int] (len T) | list of int64 tensors (one per reference)."""
        single = isinstance(fmap1, torch.Tensor)
        if isinstance(num_levels_per_target, int):
            num_levels_per_target = [num_levels_per_target]
        else:
            assert isinstance(num_levels_per_target, _LIST_TYPES)
        if single:
            assert fmap1.ndim == 4 and isinstance(fmap2, torch.Tensor)
            if fmap2.ndim == 4:
                fmap2 = fmap2.unsqueeze(0)
            assert fmap2.ndim == 5 and fmap1.shape == fmap2.shape[1:]
            assert len(num_levels_per_target) == fmap2.shape[0]
            levels = [[int(v) for v in num_levels_per_target]]
            fmap1, fmap2 = [fmap1], [fmap2]
        else:
            assert isinstance(fmap1, list) and isinstance(fmap2, list)
            assert len(fmap1) == len(fmap2) == len(num_levels_per_target)
            levels = []
            for f1, f2, lv in zip(fmap1, fmap2, num_levels_per_target):
                assert f1.ndim == 4 and f2.ndim == 5 and f1.shape == f2.shape[1:]
                lv = [int(v) for v in (lv.tolist() if isinstance(lv, torch.Tensor) else lv)]
                assert len(lv) == f2.shape[0]
                levels.append(lv)
        self._has_single_reference = single
        self._fmap1, self._fmap2, self._levels = fmap1, fmap2, levels
        self._bdhw = tuple(fmap1[0].shape)
        self._packed = [None] * len(fmap1)   # per reference group: (p1, p2) split operands when they already exist

    @classmethod
    def from_packed(cls, p1: torch.Tensor, p2: torch.Tensor, batch: int, dim: int, height: int, width: int,
                    num_levels_per_target: Union[int, Sequence[int]]) -> "CorrComputation":
        """Feature maps that already live in the engine's operand format (the encoder's last convolution writes them):
        p1 (2, B, D/32, Np, 32), p2 (2, T*B, D/32, Np, 32) fp16 hi/lo planes (k-blocked), rows >= h*w zero."""
        levels = [int(num_levels_per_target)] if isinstance(num_levels_per_target, int) else [int(v) for v in num_levels_per_target]
        T = p2.shape[1] // batch
        assert p1.shape[1] == batch and p2.shape[1] == T * batch and len(levels) == T and p1.shape[2] * 32 == p2.shape[2] * 32 == dim
        self = cls.__new__(cls)
        self._has_single_reference = True
        self._fmap1, self._fmap2, self._levels = [None], [_PackedShape(T)], [levels]
        self._bdhw = (batch, dim, height, width)
        self._packed = [(p1, p2)]
        return self

    batch = property(lambda self: self._bdhw[0])
    dim = property(lambda self: self._bdhw[1])
    height = property(lambda self: self._bdhw[2])
    width = property(lambda self: self._bdhw[3])
    num_references = property(lambda self: len(self._fmap1))
    num_targets_per_reference = property(lambda self: [f.shape[0] for f in self._fmap2])
    num_targets_overall = property(lambda self: sum(f.shape[0] for f in self._fmap2))

    @property
    def num_levels_per_target(self) -> List[torch.Tensor]:
        return [torch.tensor(lv) for lv in self._levels]

    @property
    def num_levels_per_target_merged(self) -> torch.Tensor:
        return torch.tensor(sum(self._levels, []))

    def levels_flat(self) -> List[int]:
        return sum(self._levels, [])

    def __add__(self, other: "CorrComputation") -> "CorrComputation":
        if any(p is not None for p in self._packed + other._packed):
            assert self._bdhw == other._bdhw
            out = CorrComputation.__new__(CorrComputation)
            out._has_single_reference = False
            out._fmap1, out._fmap2 = self._fmap1 + other._fmap1, self._fmap2 + other._fmap2
            out._levels, out._bdhw, out._packed = self._levels + other._levels, self._bdhw, self._packed + other._packed
            return out
        return CorrComputation(fmap1=self._fmap1 + other._fmap1, fmap2=self._fmap2 + other._fmap2,
                               num_levels_per_target=[torch.tensor(lv) for lv in self._levels + other._levels])

    def tiled_supported(self, precision: Optional[str] = None) -> bool:
        """The tiled-plane volume is written by the streaming K5 kernel only: split operands with D in {64, 128, 256}, fp16 with D in
        {128, 256}."""
        precision = default_precision(self.dim, True) if precision is None else check_precision(precision)
        return (precision == "split" and self.dim in (64, 128, 256)) or (precision in TILED_ONLY and self.dim in (128, 256))

    def pool1_fusable(self, precision: Optional[str] = None) -> bool:
        """True when the K5 launch of this precision can write the level-1 planes itself (get_correlation_volume(pool1=...))."""
        precision = default_precision(self.dim, True) if precision is None else check_precision(precision)
        if PRECISIONS[precision][0] is None or not self.tiled_supported(precision):
            return False
        ar, st16 = PRECISIONS[precision]
        return hip.pool_fusable(ar, st16, self.dim, max(f2.shape[0] for f2 in self._fmap2))

    def get_correlation_volume(self, tiled: bool = False, precision: Optional[str] = None, out: Optional[torch.Tensor] = None,
                               pool1=None) -> torch.Tensor:
        """(T, B*N, 1, h, w) fp32 -- corr.py:229-272.  One K5 launch per reference group, written straight into its
        slice of the volume (the reference expands fmap1 per target and concatenates, corr.py:254-259).
        tiled=True: (T, B, N, tiled_plane_size(h, w)) with every plane stored as 4 x 8 tiles (the look-up's layout).
        pool1 = (level1 (T1, B, N, tiled_plane_size(h//2, w//2)), keep): the level-1 planes of the targets `keep` (overall target indices,
        corr.py:297-305) are written by the K5 launches themselves (tiled, fusable precisions: `pool1_fusable`)."""
        B, D, h, w = self._bdhw
        N = h * w
        T = self.num_targets_overall
        device = self._packed[0][0].device if self._packed[0] is not None else self._fmap1[0].device
        precision = default_precision(D, tiled) if precision is None else check_precision(precision)
        if (tiled or precision in TILED_ONLY) and not (tiled and self.tiled_supported(precision)):
            raise hip.BflowHipError(f"correlation volume (tiled={tiled}, precision={precision!r}, D={D}): the tiled layout needs 'split' with D in "
                                    f"(64, 128, 256) or one of {sorted(TILED_ONLY)} with D in (128, 256), which exist in the tiled layout only")
        arithmetic, st16 = PRECISIONS[precision]
        vshape, vdtype = (T, B, N, hip.tiled_plane_size(h, w) if tiled else N), torch.float16 if st16 else torch.float32
        if out is not None:        # a caller-owned volume (double-buffered frames, bflow_amd/pipeline.py)
            assert tuple(out.shape) == vshape and out.dtype == vdtype and out.is_contiguous() and out.device == device
            vol = out
        else:
            vol = torch.empty(vshape, dtype=vdtype, device=device)
        thw = (h, w) if tiled else None
        split = precision != "f32" and D % 64 == 0

        def build(p1, p2, dst, tg, t_first):
            if precision in TILED_ONLY or pool1 is not None:
                x8 = None
                if arithmetic == hip.ARITH_SPLIT8:   # ONE conversion launch when the two operands are neighbours in one tensor (the encoder's output)
                    x8 = _x8_planes(p1, p2)
                pool = None
                if pool1 is not None:                # rows of this group's targets in the level-1 tensor (-1: the target has a single level)
                    index = [pool1[1].index(t) if t in pool1[1] else -1 for t in range(t_first, t_first + tg)]
                    pool = (pool1[0], index) if any(k >= 0 for k in index) else None
                hip.corr_build_tiled(p1, p2, dst, tg, B, N, shared_f1=True, tiled_hw=thw, arithmetic=arithmetic, x8=x8, pool=pool)
            else:
                hip.corr_build_split(p1, p2, dst, tg, B, N, shared_f1=True, tiled_hw=thw)

        if pool1 is not None:
            assert tiled and self.pool1_fusable(precision)
        t0 = 0
        for f1, f2, packed in zip(self._fmap1, self._fmap2, self._packed):
            tg = f2.shape[0]
            if packed is not None:
                build(packed[0], packed[1], vol[t0:t0 + tg], tg, t0)
                t0 += tg
                continue
            f1 = f1.float().contiguous().view(B, D, N)
            f2 = f2.float().contiguous().view(tg * B, D, N)
            if split:   # split-fp16 MFMA engine
                build(hip.split_pack(f1), hip.split_pack(f2), vol[t0:t0 + tg], tg, t0)
            else:       # exact-fp32 MFMA
                hip.corr_build_f32(f1, f2.view(tg, B, D, N), vol[t0:t0 + tg])
            t0 += tg
        return vol if tiled else vol.view(T, B * N, 1, h, w)


class CorrBlockParallelMultiTarget:
    def __init__(self,
                 corr_computation_events: Optional[CorrComputation] = None,
                 corr_computation_frames: Optional[CorrComputation] = None,
                 radius: int = 4,
                 layout: str = "rows",
                 precision: Optional[str] = None,
                 volume_out: Optional[torch.Tensor] = None):
        """layout = "rows": the reference's (T, B*N, h_L, w_L) planes (any K5 variant; every look-up entry point).
        layout = "tiled": planes stored as 4 x 8 tiles -- the inference product path (lookup_bezier_split); the reference-shaped
        accessors untile on demand.  precision: None = `default_precision(dim, tiled)`; "f16" needs layout = "tiled"."""
        assert corr_computation_events is not None or corr_computation_frames is not None
        assert radius == hip.LOOKUP_RADIUS, "the look-up radius is 4 everywhere in the reference (raft.py:40, corr.py:279)"
        assert layout in ("rows", "tiled")
        if corr_computation_frames is None:
            cc = corr_computation_events
        elif corr_computation_events is None:
            cc = corr_computation_frames
        else:
            cc = corr_computation_events + corr_computation_frames
        levels = cc.levels_flat()
        precision = default_precision(cc.dim, layout == "tiled") if precision is None else check_precision(precision)   # resolved ONCE, up front
        self.precision = precision
        self._num_targets_base = len(levels)
        self._radius = radius
        self._batch = cc.batch
        self._hw = (cc.height, cc.width)
        self._tiled = layout == "tiled"
        self._rows_cache = None
        B, h, w = cc.batch, cc.height, cc.width
        N = h * w

        fused1 = None    # (level-1 tensor, targets): written by the K5 launches themselves (fused K6, level 0 -> 1)
        if self._tiled:
            vout = None if volume_out is None else volume_out.view(len(levels), B, N, hip.tiled_plane_size(h, w))   # a caller-owned level 0
            keep1 = [t for t, lv in enumerate(levels) if lv >= 2]
            if keep1 and FUSE_POOL1 and h >= 2 and w >= 2 and cc.pool1_fusable(precision):
                fused1 = (torch.empty((len(keep1), B, N, hip.tiled_plane_size(h // 2, w // 2)), dtype=torch.float16 if PRECISIONS[precision][1] else torch.float32,
                                      device=cc._packed[0][0].device if cc._packed[0] is not None else cc._fmap1[0].device), keep1)
            base = cc.get_correlation_volume(tiled=True, precision=precision, out=vout, pool1=fused1).view(len(levels), B * N, hip.tiled_plane_size(h, w))
        else:
            assert volume_out is None, "volume_out: tiled layout only"
            base = cc.get_correlation_volume(precision=precision).view(len(levels), B * N, h, w)
        self._f16 = base.dtype == torch.float16
        # pyramid: corr.py:297-305 -- level L keeps the targets whose num_levels > L
        self._pyramid: List[Tuple[torch.Tensor, List[int]]] = [(base, list(range(len(levels))))]
        self._level_hw: List[Tuple[int, int]] = [(h, w)]
        for num_levels in range(2, max(levels) + 1):
            prev, prev_idx = self._pyramid[-1]
            keep = [t for t, lv in enumerate(levels) if lv >= num_levels]
            ph, pw = self._level_hw[-1]
            if self._tiled and num_levels == 2 and fused1 is not None:
                assert fused1[1] == keep
                cur = fused1[0].view(len(keep), B * N, hip.tiled_plane_size(ph // 2, pw // 2))
            elif self._tiled:
                cur = torch.empty((len(keep), B * N, hip.tiled_plane_size(ph // 2, pw // 2)), dtype=base.dtype, device=base.device)
                for k, t in enumerate(keep):
                    hip.corr_pool2x2_tiled(prev[prev_idx.index(t)], cur[k], ph, pw)
            else:
                cur = torch.empty((len(keep), B * N, ph // 2, pw // 2), dtype=torch.float32, device=base.device)
                for k, t in enumerate(keep):
                    hip.corr_pool2x2(prev[prev_idx.index(t)], cur[k])
            self._pyramid.append((cur, keep))
            self._level_hw.append((ph // 2, pw // 2))
        planes = []
        for lvl, (tensor, tidx) in enumerate(self._pyramid):
            for k, t in enumerate(tidx):
                planes.append(dict(tensor=tensor[k], level=lvl, target=t, hw=self._level_hw[lvl] if self._tiled else None))
        self._planes = planes
        self._table = hip.make_plane_table(planes)

    def _rows(self) -> "CorrBlockParallelMultiTarget":
        """Row-major twin of a tiled block (reference-shaped accessors and the fp32 NCHW look-ups; built once, on demand)."""
        if not self._tiled:
            return self
        if self._rows_cache is None:
            twin = CorrBlockParallelMultiTarget.__new__(CorrBlockParallelMultiTarget)
            twin.__dict__.update(self.__dict__)
            twin._tiled, twin._rows_cache, twin._f16 = False, None, False
            twin._pyramid = [(hip.untile_planes(t.float(), *hw), idx) for (t, idx), hw in zip(self._pyramid, self._level_hw)]
            twin._planes = [dict(tensor=twin._pyramid[p["level"]][0][twin._pyramid[p["level"]][1].index(p["target"])], level=p["level"],
                                 target=p["target"], hw=None) for p in self._planes]
            twin._table = hip.make_plane_table(twin._planes)
            self._rows_cache = twin
        return self._rows_cache

    @property
    def num_planes(self) -> int:
        return len(self._planes)

    def pyramid_level(self, level: int) -> Tuple[torch.Tensor, List[int]]:
        """(T_L, B*N, 1, h_L, w_L) tensor (reference CorrData.corr layout) and its base-target indices."""
        t, idx = self._rows()._pyramid[level]
        return t.unsqueeze(2), list(idx)

    def new_output(self) -> torch.Tensor:
        h, w = self._hw
        return torch.empty((self._batch, self.num_planes * 81, h, w), dtype=torch.float32, device=self._pyramid[0][0].device)

    def __call__(self, coords: Union[torch.Tensor, Sequence[torch.Tensor]], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """coords: (T, B, 2, h, w) tensor or list of T (B, 2, h, w) tensors -> (B, P*81, h, w)   (corr.py:307-351)."""
        if isinstance(coords, (list, tuple)):
            coords = torch.stack(list(coords), dim=0)
        assert coords.ndim == 5 and coords.shape[0] == self._num_targets_base
        out = self.new_output() if out is None else out
        hip.corr_lookup(self._rows()._table, coords.float().contiguous(), out)
        return out

    def lookup_bezier(self, params: torch.Tensor, coef: np.ndarray, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Fused get_flow_from_reference + coords0 + look-up (raft.py:180-184): coords are never materialised."""
        assert coef.shape[0] == self._num_targets_base
        out = self.new_output() if out is None else out
        hip.corr_lookup_bezier(self._rows()._table, params, coef, out)
        return out

    def new_output_split(self):
        """Zero-initialised blocked split tensor for lookup_bezier_split (pad channels of the last block stay zero)."""
        from .split import SplitTensor
        h, w = self._hw
        # the tiled look-up writes every channel of the last block itself (pads as zeros): only the row-major kernel needs a zeroed buffer
        return SplitTensor.empty(self._batch, h, w, self.num_planes * 81, self._pyramid[0][0].device, zero=not self._tiled)

    def lookup_bezier_split(self, params: torch.Tensor, coef: np.ndarray, out, im2col=None):
        """lookup_bezier writing the conv engine's blocked split layout directly (no NCHW intermediate).
        im2col = (SplitTensor, kh, kw, (ph, pw)) (tiled pyramids): the same launch also writes the kh x kw filter windows of `params` -- the
        input of the motion encoder's 7x7 `convf1` (update.py:91) -- into that tensor (bflow_corr_lookup_im2col)."""
        assert coef.shape[0] == self._num_targets_base
        rider = None
        if im2col is not None:
            col, kh, kw, pad = im2col
            ph, pw = (pad, pad) if isinstance(pad, int) else pad
            rider = (col.planes, kh, kw, ph, pw)
        hip.corr_lookup_bezier_split(self._table, params, coef, out.planes, tiled=self._tiled, f16_planes=self._f16, im2col=rider)
        return out

    @property
    def im2col_rider(self) -> bool:
        """True when lookup_bezier_split can carry the im2col of the Bezier parameters in its launch (the tiled look-up kernel)."""
        return self._tiled
