"""Training path: `test_mode=False` forward with gradients, the L1 sequence losses and the training step.

What runs where:
    K5 correlation volume, K6 pyramid, K7 look-up, K13 convex up-sampling, masked L1   forward AND backward on the HIP kernels
                                                                                       (csrc/backward.hip: hand-written adjoints)
    K5 backward (two GEMMs per target, K = N)                                           the conv engine as a batch of 1x1 convolutions with one
                                                                                       filter per image (_volume_adjoint_engine)
    convolutions (forward, input and weight gradient)                                   the split-fp16 conv engine + bflow_conv_wgrad_halo
                                                                                       (conv_train.py: custom autograd Function)
    norms / activations / GRU gate arithmetic                                           torch element-wise ops under autograd
    the whole step (forward + loss + backward + AdamW)                                  optionally ONE hipGraph (GraphedTrainStep)

Reference: RAFTSpline.forward with test_mode=False (models/raft_spline/raft.py:101-200), utils/losses.py:6-62,
RAFTSplineModule.training_step / configure_optimizers (modules/raft_spline.py:64-188,321-356).

The 4-D volume gradient is the one large object of the backward pass (368.6 MB per sample at DSEC size); every look-up of every
GRU iteration adds its sparse patch gradients INTO ONE buffer owned by the correlation block (instead of autograd materialising
and summing one dense volume gradient per iteration), and a token tensor threads the autograd dependency so that the volume's own
backward (pooling adjoint, then the two GEMMs) runs after the last look-up has contributed.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import conv_train, hip
from . import split as S
from .bezier import BezierCurves
from .corr import CorrBlockParallelMultiTarget, CorrComputation
from .validation import DataLoading, DataSetType, _get


# ----------------------------------------------------------------------------------------------- correlation block with gradients
class TrainCorrBlock:
    """groups: [(fmap1 (B,D,h,w), fmap2 (T,B,D,h,w), levels list)] -- events first, then frames (corr.py:276-305)."""

    def __init__(self, groups: Sequence[Tuple[torch.Tensor, torch.Tensor, Sequence[int]]]):
        self._levels = [list(int(v) for v in g[2]) for g in groups]
        self._group_sizes = [int(g[1].shape[0]) for g in groups]
        self._inner: Optional[CorrBlockParallelMultiTarget] = None
        self._grads: Optional[List[torch.Tensor]] = None
        self._grad_table = None
        flat: List[torch.Tensor] = []
        for f1, f2, _ in groups:
            assert f1.ndim == 4 and f2.ndim == 5 and f1.shape == f2.shape[1:]
            flat += [f1, f2]
        self.token = _CorrBlockFn.apply(self, *flat)

    # -- forward (inside _CorrBlockFn.forward: no graph is recorded)
    def _build(self, fmaps: Sequence[torch.Tensor]):
        ccs = []
        for g, lv in enumerate(self._levels):
            ccs.append(CorrComputation(fmaps[2 * g].float().contiguous(), fmaps[2 * g + 1].float().contiguous(), lv))
        cc = ccs[0]
        for other in ccs[1:]:
            cc = cc + other
        self._inner = CorrBlockParallelMultiTarget(corr_computation_events=cc)
        self._plane_targets = [p["target"] for p in self._inner._planes]

    @property
    def num_planes(self) -> int:
        return self._inner.num_planes

    def _ensure_grads(self):
        if self._grads is None:
            self._grads = [torch.zeros_like(t) for t, _ in self._inner._pyramid]
            per_plane = []
            for lvl, (_, tidx) in enumerate(self._inner._pyramid):
                for k in range(len(tidx)):
                    per_plane.append(self._grads[lvl][k])
            self._grad_table = hip.make_grad_table(per_plane)
        return self._grad_table

    # -- backward of the volume: pooling adjoint coarse -> fine, then dC -> d(fmaps)
    def _backward_volume(self, fmaps: Sequence[torch.Tensor]) -> List[Optional[torch.Tensor]]:
        if self._grads is None:
            return [None] * len(fmaps)
        pyr = self._inner._pyramid
        for lvl in range(len(pyr) - 1, 0, -1):
            _, idx = pyr[lvl]
            _, prev_idx = pyr[lvl - 1]
            for k, t in enumerate(idx):
                hip.corr_pool2x2_bwd(self._grads[lvl][k], self._grads[lvl - 1][prev_idx.index(t)])
        g0 = self._grads[0]                                    # (T, B*N, h, w)
        T = g0.shape[0]
        out: List[Optional[torch.Tensor]] = []
        t0 = 0
        for g, tg in enumerate(self._group_sizes):
            f1, f2 = fmaps[2 * g], fmaps[2 * g + 1]
            B, D, h, w = f1.shape
            N = h * w
            dC = g0[t0:t0 + tg].view(tg, B, N, N)               # dC[t,b,n,m]: n = reference pixel, m = target pixel
            a = f1.float().reshape(B, D, N)
            b = f2.float().reshape(tg, B, D, N)
            s = 1.0 / math.sqrt(D)
            if NATIVE_VOLUME_ADJOINT and dC.is_cuda and D % 32 == 0:
                gf1, gf2 = _volume_adjoint_engine(dC, a, b, s)
            else:
                gf1 = torch.matmul(b, dC.transpose(-1, -2)).sum(dim=0).mul_(s)      # (B, D, N): sum_t f2[t] @ dC[t]^T
                gf2 = torch.matmul(a.unsqueeze(0), dC).mul_(s)                     # (tg, B, D, N): f1 @ dC[t]
            out += [gf1.view(B, D, h, w).to(f1.dtype), gf2.view(tg, B, D, h, w).to(f2.dtype)]
            t0 += tg
        assert t0 == T
        self._grads = None                                     # one backward per forward
        self._grad_table = None
        return out

    def lookup_bezier(self, params: torch.Tensor, coef: np.ndarray) -> torch.Tensor:
        """(B, 2*deg, h, w) Bezier parameters -> (B, P*81, h, w) correlation features, differentiable in params and the volume."""
        return _LookupBezierFn.apply(self, coef, params, self.token)


NATIVE_VOLUME_ADJOINT = True        # tools / A-B: False = the two GEMMs per target through torch.matmul (rocBLAS)


def _stacked_filters(w: torch.Tensor, n_pad: int):
    """w (S, D, N): S filters of a 1x1 "convolution" with N input channels -> packed for bflow_conv_desc_t.weight_sets = S (filter set s =
    k-tiles [s * n_pad / 32, (s + 1) * n_pad / 32))."""
    S_, D, N = w.shape
    if N == n_pad:
        flat = w.permute(1, 0, 2).reshape(D, S_ * N, 1, 1)
    else:
        flat = torch.zeros((D, S_, n_pad), dtype=torch.float32, device=w.device)
        flat[:, :, :N] = w.permute(1, 0, 2)
        flat = flat.reshape(D, S_ * n_pad, 1, 1)
    planes, (cout, _, _, _, cout_pad) = S.PackedConvWeight().get(flat.contiguous())
    return planes, (cout, n_pad, 1, 1, cout_pad)


def _volume_adjoint_engine(dC: torch.Tensor, f1: torch.Tensor, f2: torch.Tensor, s: float):
    """The adjoint of CorrComputation._corr_dot_prod_util (corr.py:264-272) on the split-fp16 conv engine instead of rocBLAS:
        d f1[b, d, n] = s * sum_t sum_m f2[t, b, d, m] dC[t, b, n, m]        d f2[t, b, d, m] = s * sum_n f1[b, d, n] dC[t, b, n, m]
    Both are 1x1 "convolutions" over N pixels with N input channels and one filter per image (weight_sets): for d f2 the pixel index is m and
    dC[t, b] (n-major) IS the NCHW operand; for d f1 the pixel index is n and the row-major dC is staged by bflow_rows_to_split.  dC (values
    of 1e-6 ... 1e-9) is pre-scaled by a device-side power of two, as in conv_train."""
    tg, B, N, _ = dC.shape
    D = f1.shape[1]
    n_pad = (N + 31) // 32 * 32
    dCf = dC.reshape(tg * B, N, N).contiguous()
    sc = S.pow2_scale(dCf, conv_train._TARGET)
    sc_s = sc * s                                                   # {s_pow2 * s (unused), s / s_pow2}: the un-scaling also applies 1 / sqrt(D)
    inv = sc_s[1:2]
    # d f2: image (t, b), pixel m, channel n; filter of image (t, b) = f1[b]  (image index % B == b)
    x2 = S.from_nchw(dCf.view(tg * B, N, 1, N), sc[0:1])
    _, o2 = S.conv(x2, _stacked_filters(f1, n_pad), want_split=False, want_f32=True, weight_sets=B, tile=128)
    gf2 = S.blocked_f32_to_nchw(o2, D, 1, N, inv).view(tg, B, D, N)
    # d f1: image (t, b), pixel n, channel m; filter of image (t, b) = f2[t, b]; summed over t
    x1 = S.from_rows(dCf, sc[0:1])
    _, o1 = S.conv(x1, _stacked_filters(f2.reshape(tg * B, D, N), n_pad), want_split=False, want_f32=True, weight_sets=tg * B, tile=128)
    gf1 = S.blocked_f32_to_nchw(o1, D, 1, N, inv).view(tg, B, D, N).sum(dim=0)
    return gf1, gf2


class _CorrBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, block: TrainCorrBlock, *fmaps):
        block._build([f.detach() for f in fmaps])
        ctx.block = block
        ctx.save_for_backward(*fmaps)
        return torch.zeros(1, dtype=torch.float32, device=fmaps[0].device)

    @staticmethod
    def backward(ctx, _grad_token):
        grads = ctx.block._backward_volume(ctx.saved_tensors)
        return (None, *grads)


class _LookupBezierFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, block: TrainCorrBlock, coef: np.ndarray, params: torch.Tensor, token: torch.Tensor):
        p = params.detach().float().contiguous()
        out = block._inner.lookup_bezier(p, coef)
        ctx.block, ctx.coef = block, coef
        ctx.save_for_backward(p)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (p,) = ctx.saved_tensors
        block, coef = ctx.block, ctx.coef
        inner = block._inner
        gc = hip.corr_lookup_bezier_bwd(inner._table, block._ensure_grads(), p, coef, grad_out.float().contiguous())
        T, deg = coef.shape
        B, _, h, w = p.shape
        gt = torch.zeros((T, B, 2, h, w), dtype=torch.float32, device=p.device)
        for k, t in enumerate(block._plane_targets):          # planes of one target (its pyramid levels), fixed order
            gt[t] += gc[k]
        cf = hip.const_tensor(coef, p.device)
        # sum_t gt[t, b, d] * cf[t, p] as a broadcast product (T, deg <= 16: an einsum would go through a library GEMM)
        gp = (gt.unsqueeze(3) * cf.view(T, 1, 1, deg, 1, 1).to(gt.dtype)).sum(dim=0).reshape(B, 2 * deg, h, w)
        return None, None, gp, torch.zeros(1, dtype=torch.float32, device=p.device)


class _CvxUpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data: torch.Tensor, mask: torch.Tensor):
        d, m = data.detach().float().contiguous(), mask.detach().float().contiguous()
        ctx.save_for_backward(d, m)
        return hip.cvx_upsample(d, m)

    @staticmethod
    def backward(ctx, grad_up):
        d, m = ctx.saved_tensors
        gd, gm = hip.cvx_upsample_bwd(grad_up.float().contiguous(), d, m)
        return gd, gm


def cvx_upsample(data: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Differentiable K13 (raft_utils/utils.py:33-48)."""
    return _CvxUpsampleFn.apply(data, mask)


# ----------------------------------------------------------------------------------------------- losses (utils/losses.py)
class _L1MaskedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, source: torch.Tensor, target: torch.Tensor, valid: Optional[torch.Tensor]):
        s, t = source.detach().float().contiguous(), target.detach().float().contiguous()
        v = None if valid is None else valid.contiguous()
        acc = torch.zeros(2, dtype=torch.float64, device=s.device)
        hip.l1_masked_accumulate(s, t, v, acc)
        ctx.save_for_backward(s, t, acc) if v is None else ctx.save_for_backward(s, t, acc, v)
        ctx.masked = v is not None
        return (acc[0] / acc[1]).float()

    @staticmethod
    def backward(ctx, upstream):
        saved = ctx.saved_tensors
        s, t, acc = saved[:3]
        v = saved[3] if ctx.masked else None
        g = hip.l1_masked_grad(s, t, v, acc, upstream.float().contiguous().view(1), 1.0)
        return g, None, None


def l1_loss_channel_masked(source: torch.Tensor, target: torch.Tensor, valid_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """losses.py:6-22: mean over (valid) positions of the channel-summed absolute difference."""
    assert source.ndim > 2 and source.shape == target.shape
    if valid_mask is not None:
        assert valid_mask.shape[0] == target.shape[0] and valid_mask.ndim == target.ndim - 1 and valid_mask.dtype == torch.bool
        assert tuple(valid_mask.shape) == (source.shape[0],) + tuple(source.shape[2:])
    return _L1MaskedFn.apply(source, target, valid_mask)


def l1_seq_loss_channel_masked(source_list: Sequence[torch.Tensor], target: torch.Tensor, valid_mask: Optional[torch.Tensor] = None,
                               gamma: float = 0.8) -> torch.Tensor:
    """losses.py:24-40: sum_i gamma^(I-1-i) * l1(source_i)."""
    n = len(source_list)
    loss = 0
    for i, src in enumerate(source_list):
        loss = loss + gamma ** (n - i - 1) * l1_loss_channel_masked(src, target, valid_mask)
    return loss


def l1_multi_seq_loss_channel_masked(src_list_list: Sequence[Sequence[torch.Tensor]], target_list: Sequence[torch.Tensor],
                                     valid_mask_list: Optional[Sequence[torch.Tensor]] = None, gamma: float = 0.8) -> torch.Tensor:
    """losses.py:42-62: per iteration the mean over the M supervision targets, weighted like the sequence loss."""
    loss = 0
    n_iters = len(src_list_list)
    for it, sources in enumerate(src_list_list):
        assert len(sources) > 0 and len(sources) == len(target_list)
        it_loss = 0
        for k, src in enumerate(sources):
            it_loss = it_loss + l1_loss_channel_masked(src, target_list[k], None if valid_mask_list is None else valid_mask_list[k])
        loss = loss + gamma ** (n_iters - it - 1) * (it_loss / len(sources))
    return loss


# ----------------------------------------------------------------------------------------------- the training forward
class _GruZRFn(torch.autograd.Function):
    """(z, r*h) = (sigmoid(zr[:, :C]), sigmoid(zr[:, C:]) * h): bflow_gru_zr_fwd / bflow_gru_zr_bwd (update.py:38-44 under autograd)."""

    @staticmethod
    def forward(ctx, zr_pre, h):
        z, r, rh = hip.gru_zr_fwd(zr_pre, h)
        hs = h.detach().float().contiguous()
        ctx.save_for_backward(z, r, hs)
        return z, rh

    @staticmethod
    def backward(ctx, dz, drh):
        z, r, h = ctx.saved_tensors
        if drh is None:
            drh = torch.zeros_like(h)
        dzr, dh = hip.gru_zr_bwd(dz, drh, z, r, h)
        return dzr, dh


class _GruBlendFn(torch.autograd.Function):
    """h' = (1 - z) h + z tanh(q_pre): bflow_gru_blend_fwd / bflow_gru_blend_bwd (update.py:45-47 under autograd)."""

    @staticmethod
    def forward(ctx, q_pre, z, h):
        q, hn = hip.gru_blend_fwd(q_pre, z, h)
        ctx.save_for_backward(q, z.detach().float().contiguous(), h.detach().float().contiguous())
        return hn

    @staticmethod
    def backward(ctx, dhn):
        q, z, h = ctx.saved_tensors
        return hip.gru_blend_bwd(dhn, q, z, h)


FUSED_GATES = True                  # tools / A-B: False = the reference's chain of element-wise torch ops under autograd
HOIST_INP = True                    # the context features' share of the GRU gate convolutions once per forward instead of once per iteration


def _update_block_train(ub, net, inp, corr, bezier, merged=None):
    """BasicUpdateBlock.forward (update.py:116-126) under autograd; the convolutions (modules = conv_train.Conv2d, the merged z|r filter
    through conv_train.conv2d) run forward and backward on the HIP conv engine; z|r share one launch per GRU half; the gate arithmetic of
    a half is two fused launches (csrc/gru_gates.hip).  `merged`: dict the caller keeps for ONE forward -- the z|r filters / biases are
    concatenated once per forward, not once per iteration."""
    enc, gru = ub.encoder, ub.gru
    cor = enc.convc2.relu(enc.convc1.relu(corr))                 # relu(conv(.)): the activation is the convolution's epilogue
    bez = enc.convf2.relu(enc.convf1.relu(bezier))
    motion = torch.cat([enc.conv.relu(torch.cat([cor, bez], dim=1)), bezier], dim=1)
    hd = ub.hidden_dim
    cd = inp.shape[1]
    merged = {} if merged is None else merged
    hoist = HOIST_INP and inp.is_cuda
    x = None if hoist else torch.cat([inp, motion], dim=1)
    for sfx in ("1", "2"):
        cz, cr, cq = (getattr(gru, f"conv{g}{sfx}") for g in "zrq")
        pad = cz.padding
        if sfx not in merged:
            wzr, bzr = torch.cat([cz.weight, cr.weight], dim=0), torch.cat([cz.bias, cr.bias], dim=0)
            if hoist:
                # the GRU input is cat(h, inp, motion) and `inp` does not change over the iterations (raft.py:145-147): its share of every
                # gate convolution is computed ONCE per forward (as the inference path does, update.py prepare_inp) and added per iteration
                pc = lambda tag: gru.__dict__.setdefault("_pack_" + tag + sfx, conv_train._PackCache())
                srcs_zr, srcs_q = (cz.weight, cr.weight), (cq.weight,)
                t_zr = conv_train.conv2d(inp, wzr[:, hd:hd + cd].contiguous(), bzr, pad, pc("zr_inp"), srcs_zr)
                t_q = conv_train.conv2d(inp, cq.weight[:, hd:hd + cd].contiguous(), cq.bias, pad, pc("q_inp"), srcs_q)
                w_zr = torch.cat([wzr[:, :hd], wzr[:, hd + cd:]], dim=1)
                w_q = torch.cat([cq.weight[:, :hd], cq.weight[:, hd + cd:]], dim=1)
                merged[sfx] = (w_zr, t_zr, w_q, t_q, pc("zr_hm"), pc("q_hm"), srcs_zr, srcs_q)
            else:
                merged[sfx] = (wzr, bzr)
        if hoist:
            w_zr, t_zr, w_q, t_q, pk_zr, pk_q, srcs_zr, srcs_q = merged[sfx]
            zr = conv_train.conv2d(torch.cat([net, motion], dim=1), w_zr, None, pad, pk_zr, srcs_zr) + t_zr
        else:
            wzr, bzr = merged[sfx]
            zr = conv_train.conv2d(torch.cat([net, x], dim=1), wzr, bzr, pad,
                                   gru.__dict__.setdefault("_zr_pack" + sfx, conv_train._PackCache()), (cz.weight, cr.weight))
        fused = FUSED_GATES and zr.is_cuda and (hd * net.shape[2] * net.shape[3]) % 4 == 0
        if fused:
            z, rh = _GruZRFn.apply(zr, net)
        else:
            z, r = torch.sigmoid(zr[:, :hd]), torch.sigmoid(zr[:, hd:])
            rh = r * net
        q_pre = (conv_train.conv2d(torch.cat([rh, motion], dim=1), w_q, None, pad, pk_q, srcs_q) + t_q) if hoist else cq(torch.cat([rh, x], dim=1))
        net = _GruBlendFn.apply(q_pre, z, net) if fused else (1 - z) * net + z * torch.tanh(q_pre)
    delta = ub.bezier_head.conv2(ub.bezier_head.conv1.relu(net))
    mask = 0.25 * ub.mask[2](ub.mask[0].relu(net))               # nn.Sequential(conv, ReLU, conv) (update.py:111-114)
    return net, mask, delta


def forward_train(model, voxel_grid: Optional[torch.Tensor], images: Optional[List[torch.Tensor]], iters: int,
                  flow_init: Optional[torch.Tensor]) -> Tuple[torch.Tensor, List[BezierCurves]]:
    """raft.py:101-200 under autograd: (low-resolution parameters, one up-sampled BezierCurves per GRU iteration)."""
    # What the differentiable path does not run is refused HERE, before the first launch (the eval engine's check_engine_support covers more:
    # GroupNorm encoders and zero-padded feature dims exist on the inference engine only).
    bad = []
    for name in ("fnet_ev", "fnet_img", "cnet"):
        net_ = getattr(model, name)
        if net_ is None:
            continue
        if net_.norm_fn == "group":
            bad.append(f"{name}.norm_fn = 'group' (GroupNorm has no training kernels: norm_train.py)")
        if name != "cnet" and net_.conv2.out_channels not in (64, 128, 256):
            bad.append(f"{name} output dim {net_.conv2.out_channels} (the training correlation contracts 64, 128 or 256 feature channels; other dims are "
                       "zero-padded on the inference engine only)")
    if bad and (voxel_grid if voxel_grid is not None else images[0]).is_cuda:
        raise hip.BflowHipError("RAFTSpline training forward on the GPU: not supported: " + "; ".join(bad))
    hdim, cdim = model.hidden_dim, model.context_dim
    groups = []
    context_input = None
    if model.fnet_ev is not None:
        assert voxel_grid is not None
        voxel_grid = voxel_grid.contiguous().float()
        grids, context_input = model.gen_voxel_grids(voxel_grid)
        fm = [f.float() for f in model.fnet_ev(list(grids))]
        groups.append((fm[0], torch.stack(fm[1:], dim=0), list(model.ev_corr_levels)))
    if model.fnet_img is not None:
        assert images is not None and len(images) == 2
        images = [2 * (x.float().contiguous() / 255) - 1 for x in images]
        fi = [f.float() for f in model.fnet_img(list(images))]
        groups.append((fi[0], fi[1].unsqueeze(0), [int(model.img_corr_params["levels"])]))
        context_input = images[0] if context_input is None else torch.cat((context_input, images[0]), dim=-3)
    assert context_input is not None
    B, _, H, W = context_input.shape
    assert H % 8 == 0 and W % 8 == 0
    h, w = H // 8, W // 8

    cnet = model.cnet(context_input.contiguous())
    net, inp = torch.split(cnet, [hdim, cdim], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)

    block = TrainCorrBlock(groups)
    params = torch.zeros((B, 2 * model.bezier_degree, h, w), dtype=torch.float32, device=context_input.device)
    if flow_init is not None:
        params = params + flow_init
    coef = model._coefficients()
    ups: List[BezierCurves] = []
    merged: Dict[str, Any] = {}                                  # z|r filters of the two GRU halves, concatenated once per forward
    for _ in range(iters):
        if model.detach_bezier:                                  # raft.py:167-168
            params = params.detach()
        corr = block.lookup_bezier(params, coef)
        net, mask, delta = _update_block_train(model.update_block, net, inp, corr, params, merged)
        params = params + delta                                  # bezier.py:137-139
        ups.append(BezierCurves(cvx_upsample(params, mask)))
    return params, ups


# ----------------------------------------------------------------------------------------------- the training step
def configure_optimizers(model: torch.nn.Module, train_params: Dict[str, Any], capturable: bool = False):
    """modules/raft_spline.py:321-356: AdamW (+ linear OneCycleLR stepped per optimiser step).  capturable: the step counters and the
    learning rate live on the device, so that `GraphedTrainStep` can record the optimiser step into its hipGraph."""
    lr = train_params["learning_rate"]
    if capturable:
        lr = torch.tensor(float(lr), dtype=torch.float32, device=next(model.parameters()).device)
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=train_params["weight_decay"], capturable=capturable)
    sch_p = train_params["lr_scheduler"]
    if not sch_p["use"]:
        return opt, None
    total = sch_p["total_steps"]
    assert total is not None and total > 0
    sch = torch.optim.lr_scheduler.OneCycleLR(optimizer=opt, max_lr=train_params["learning_rate"], total_steps=total + 100,
                                              pct_start=sch_p["pct_start"], cycle_momentum=False, anneal_strategy="linear")
    return opt, sch


class TrainStep:
    """RAFTSplineModule.training_step (modules/raft_spline.py:64-188) without the Lightning harness: batch dict -> outputs with
    'loss' (call .backward() on it).  DSEC: L1 sequence loss of flow(1.0) over the iterations; MultiFlow: the multi-target sequence
    loss at the GT timestamps, or -- with multi_loss off -- the reference's sequence loss over the LAST iteration's per-timestamp
    predictions against the last GT (:148-153)."""

    def __init__(self, model, num_iter_train: int, use_events: bool = True, use_images: bool = False, multi_loss: bool = True):
        self.model, self.iters = model, int(num_iter_train)
        self.use_events, self.use_images, self.multi_loss = use_events, use_images, multi_loss

    @staticmethod
    def timestamps(batch: Dict[Any, Any]) -> Optional[List[float]]:
        """MultiFlow: the GT timestamps as host floats (one per GT, equal along the batch: modules/raft_spline.py:135-139) -- a device
        synchronisation; DSEC: None."""
        tss = _get(batch, DataLoading.FLOW_TIMESTAMPS)
        if tss is None:
            return None
        times = []
        for ts in tss:
            if ts.numel() > 1:
                assert 0 <= (ts[1:] - ts[:-1]).abs().mean().item() < 0.001
            times.append(float(ts[0].item()))
        return times

    def __call__(self, batch: Dict[Any, Any], times: Optional[List[float]] = None) -> Dict[str, Any]:
        """times: the result of `timestamps(batch)` when the caller already has it (GraphedTrainStep reads it outside its capture)."""
        gt = _get(batch, DataLoading.FLOW)
        valid = _get(batch, DataLoading.FLOW_VALID)
        ev = _get(batch, DataLoading.EV_REPR)
        images = _get(batch, DataLoading.IMG) if self.use_images else None
        ref = ev if ev is not None else images[0]
        if ref.shape[-2] % 8 or ref.shape[-1] % 8:
            raise NotImplementedError("training inputs must be multiples of 8 (modules/raft_spline.py:80-82)")
        ds = _get(batch, DataLoading.DATASET_TYPE)[0]
        ds = getattr(ds, "name", ds)
        preds: List[BezierCurves] = self.model(voxel_grid=ev if self.use_events else None, images=images, iters=self.iters,
                                               test_mode=False)
        out: Dict[str, Any] = {"bezier_prediction": preds[-1].detach()}
        if ds == DataSetType.DSEC.name:
            flows = [p.get_flow_from_reference(1.0) for p in preds]
            out.update(loss=l1_seq_loss_channel_masked(flows, gt, valid), pred=flows[-1], gt=gt, gt_valid=valid)
            return out
        if ds == DataSetType.MULTIFLOW2D.name:
            times = self.timestamps(batch) if times is None else times
            flows = [[p.get_flow_from_reference(t) for t in times] for p in preds]
            loss = l1_multi_seq_loss_channel_masked(flows, gt) if self.multi_loss else l1_seq_loss_channel_masked(flows[-1], gt[-1])
            out.update(loss=loss, pred=flows[-1][-1], gt=gt)
            return out
        raise NotImplementedError(ds)


class GraphedTrainStep:
    """forward + loss + backward + AdamW step of one batch signature as ONE hipGraph (torch.cuda.CUDAGraph == hipGraph on ROCm).

    A training step at the reference's DSEC shape is ~6000 small launches; eagerly the HOST is the bound (tools/train_probe.py: the time
    to enqueue a step equals the step time, on the library convolutions as well as on the engine).  Captured once and replayed, the step
    runs at the speed of its kernels.  The optimiser must be built with `configure_optimizers(..., capturable=True)`; the scheduler (if
    any) is stepped on the host after every replay -- it only fills the learning-rate tensor the recorded AdamW kernels read.
    The batch tensors are copied into the graph's static inputs; the returned loss / prediction are the graph's static outputs
    (overwritten by the next call)."""

    _epochs = 0

    def __init__(self, step: TrainStep, optimizer: torch.optim.Optimizer, scheduler=None, warmup: int = 2, grad_sync=None, max_graphs: int = 1):
        """grad_sync: a `dist.GradientBuckets` of the model (data-parallel training, one process per GPU): its bucketed RCCL all-reduces are
        launched by the gradient hooks during the backward pass and finished before the optimiser step -- recorded into the graph like
        every other node (stream-ordered collectives are capturable).  On one rank it is a no-op."""
        self.step, self.opt, self.sch, self.warmup, self.grad_sync = step, optimizer, scheduler, int(warmup), grad_sync
        self._sig = None
        self._times = None
        self._graph = None
        # the most recently used signatures keep their captured graph (a smaller last batch per epoch, alternating shapes or timestamp sets
        # would otherwise pay two warm-up steps + a capture at every change): signature -> (graph, static inputs, static outputs, times)
        self._lru: "OrderedDict[Tuple, Tuple]" = OrderedDict()
        # max_graphs = captured graphs alive at a time (the current one + parked ones).  Every graph owns a private memory pool with ALL
        # activations of a training step, so the default keeps ONE (what fitted before the LRU existed still fits); a run whose batch shape
        # alternates (a smaller last batch per epoch) can buy back the re-captures with max_graphs = 2 or 3 at 2-3x the activation memory.
        self.max_graphs = max(1, int(max_graphs))

    @staticmethod
    def _sig_of(v):
        if torch.is_tensor(v):
            return (tuple(v.shape), str(v.dtype), str(v.device))
        if isinstance(v, (list, tuple)):
            return tuple(GraphedTrainStep._sig_of(x) for x in v)
        return repr(v)

    @staticmethod
    def _clone(v):
        if torch.is_tensor(v):
            return v.clone()
        if isinstance(v, (list, tuple)):
            return type(v)(GraphedTrainStep._clone(x) for x in v)
        return v

    @staticmethod
    def _copy_into(dst, src):
        if torch.is_tensor(dst):
            dst.copy_(src)
        elif isinstance(dst, (list, tuple)):
            for d, s_ in zip(dst, src):
                GraphedTrainStep._copy_into(d, s_)

    def _signature(self, batch, times) -> Tuple:
        return tuple((str(k), self._sig_of(batch[k])) for k in sorted(batch, key=str)) + (("times", tuple(times) if times else None),)

    def _run(self, batch):
        out = self.step(batch, times=self._times)
        out["loss"].backward()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.opt.step()
        return out

    def _capture(self, batch):
        import gc
        self._static = {k: self._clone(v) for k, v in batch.items()}
        # warm-up on a side stream (lazy state: AdamW moments -- created INSIDE a capture they would be re-zeroed by every replay --,
        # device constants, allocator growth), then undone: parameters, buffers and optimiser state are put back, so neither the
        # warm-up nor the capture counts as a training step
        model = self.step.model
        params = [p for g in self.opt.param_groups for p in g["params"]]
        snap_p = [p.detach().clone() for p in params]
        snap_b = [b.detach().clone() for b in model.buffers()]
        snap_s = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in self.opt.state[p].items()} for p in params if p in self.opt.state}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, self.warmup)):
                self.opt.zero_grad(set_to_none=True)
                self._run(self._static)
            with torch.no_grad():
                for p, q in zip(params, snap_p):
                    p.copy_(q)
                for b, q in zip(model.buffers(), snap_b):
                    b.copy_(q)
                for p in params:
                    for k, v in self.opt.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            old = snap_s.get(id(p), {}).get(k)
                            v.copy_(old) if old is not None else v.zero_()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.opt.zero_grad(set_to_none=True)                # .grad tensors are allocated from the graph's pool and overwritten by every replay
        self._mutated = list(params) + list(model.buffers())
        self._graph = torch.cuda.CUDAGraph()
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()                                        # see graph.py: no finaliser of an older graph inside the capture
        GraphedTrainStep._epochs += 1
        conv_train.CAPTURE_EPOCH = GraphedTrainStep._epochs  # filter packs are RE-BUILT (once each) inside the graph: a warm cache would freeze stale packs into it
        try:
            with torch.cuda.graph(self._graph):
                self._out = self._run(self._static)
        finally:
            conv_train.CAPTURE_EPOCH = 0
            if gc_was_on:
                gc.enable()

    def __call__(self, batch: Dict[Any, Any]) -> Dict[str, Any]:
        times = self.step.timestamps(batch)                 # MultiFlow: host floats frozen into the graph (Bezier coefficient tables) -> part of the signature
        sig = self._signature(batch, times)
        if self._sig != sig:
            if self._graph is not None:                     # park the current graph under its signature
                self._lru[self._sig] = (self._graph, self._static, self._out, self._times)
                self._lru.move_to_end(self._sig)
            hit = self._lru.pop(sig, None)
            if hit is not None:
                self._graph, self._static, self._out, self._times = hit
                self._sig = sig
            else:
                if self._sig is not None and not getattr(self, "_recapture_noted", False):
                    self._recapture_noted = True            # once: a run whose batch signature alternates pays two warm-up steps + a capture per change
                    import warnings
                    warnings.warn(f"GraphedTrainStep: a new batch signature forces a re-capture (max_graphs = {self.max_graphs}; the graph of the previous "
                                  "signature is " + ("destroyed to make room" if len(self._lru) >= self.max_graphs else "parked") +
                                  "); max_graphs = 2 or 3 keeps alternating shapes captured at 2-3x the activation memory")
                # every graph owns all activations of a step: the oldest parked graph has to go BEFORE the new capture allocates its pool (with
                # max_graphs = 1 that is the graph that was current a moment ago: a capture that then fails leaves no graph at all, see below)
                while len(self._lru) >= self.max_graphs:    # destroyed here, outside any capture
                    torch.cuda.synchronize()
                    self._lru.popitem(last=False)
                # a capture can fail part-way (a constant missing from the cache, out of memory): the step then has NO current graph --
                # signature, times and graph are only adopted after a successful capture, so the next call captures again instead of
                # replaying a half-built graph on stale static buffers
                self._graph, self._sig = None, None
                prev_times, self._times = self._times, times      # _capture reads self._times (frozen into the graph)
                try:
                    self._capture(batch)                    # warm-up steps + the capture itself do NOT count as training steps of `batch` ...
                except Exception:
                    self._graph, self._times = None, prev_times
                    raise
                self._sig = sig
        for k, v in batch.items():
            self._copy_into(self._static[k], v)
        self._graph.replay()                                # ... this replay does
        # the replay rewrote parameters, BatchNorm buffers and optimiser state behind autograd's back: bump their version counters (no
        # launch), every cache keyed on them -- packed filters of the inference engine, captured inference graphs -- then sees the change
        torch.autograd.graph.increment_version(self._mutated)
        if self.sch is not None:
            self.sch.step()
        return self._out
