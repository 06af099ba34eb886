"""CPU ORACLE for the RAFT-spline inference hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a functional, torch-CPU (fp32) restatement of the reference algorithm
(uzh-rpg/bflow).  It exists to check the HIP path; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The product
package (`bflow_amd`) never imports anything from `oracle/`.

Pinning: the reference ships no tests / golden vectors for this path, so the
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, run in the build container:
  * live   : tests/test_oracle_vs_reference.py (runs whenever /root/reference is present)
  * frozen : tests/golden/*.npz, produced by tests/golden/make_golden.py from the reference.

Every function cites the reference file:line it restates (paths relative to the reference root).
The model is expressed over a flat state dict {reference parameter name: tensor} instead of
nn.Modules, so there is exactly one code path per op and no hidden module state.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]

LOOKUP_RADIUS = 4  # hard-coded in the reference: models/raft_spline/raft.py:40, raft_utils/corr.py:279


# --------------------------------------------------------------------------------------
# Parameter inventory + deterministic weights
# --------------------------------------------------------------------------------------
def _encoder_shapes(prefix: str, c_in: int, c_out: int, norm: str, out: Dict[str, Tuple[int, ...]]):
    """Parameter names/shapes of BasicEncoder (models/raft_utils/extractor.py:58-100)."""

    def conv(name, co, ci, kh, kw):
        out[f"{prefix}.{name}.weight"] = (co, ci, kh, kw)
        out[f"{prefix}.{name}.bias"] = (co,)

    def bn(name, c):
        if norm not in ("batch", "group"):
            return  # InstanceNorm2d has no parameters (affine=False), extractor.py:27-31; 'none' is an empty nn.Sequential (:33-37)
        out[f"{prefix}.{name}.weight"] = (c,)
        out[f"{prefix}.{name}.bias"] = (c,)
        if norm == "group":
            return  # nn.GroupNorm: affine parameters only (extractor.py:15-19)
        out[f"{prefix}.{name}.running_mean"] = (c,)
        out[f"{prefix}.{name}.running_var"] = (c,)
        out[f"{prefix}.{name}.num_batches_tracked"] = ()

    bn("norm1", 64)
    conv("conv1", 64, c_in, 7, 7)
    cin = 64
    for li, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 2)), start=1):
        for bi in range(2):
            p = f"layer{li}.{bi}"
            s = stride if bi == 0 else 1
            conv(f"{p}.conv1", dim, cin, 3, 3)
            conv(f"{p}.conv2", dim, dim, 3, 3)
            bn(f"{p}.norm1", dim)
            bn(f"{p}.norm2", dim)
            if s != 1:
                bn(f"{p}.norm3", dim)  # registered twice: as .norm3 and as .downsample.1 (extractor.py:43-44)
                conv(f"{p}.downsample.0", dim, cin, 1, 1)
                bn(f"{p}.downsample.1", dim)
            cin = dim
    conv("conv2", c_out, 128, 1, 1)


def num_corr_planes(cfg: Dict[str, Any]) -> int:
    """models/raft_spline/update.py:69-86 (_num_cor_planes)."""
    corr = cfg["correlation"]
    n = 0
    if cfg["use_events"]:
        for lvl, rad in zip(corr["ev"]["levels"], corr["ev"]["radius"]):
            n += lvl * (2 * rad + 1) ** 2
    if cfg["use_boundary_images"]:
        n += corr["img"]["levels"] * (2 * corr["img"]["radius"] + 1) ** 2
    return n


def param_shapes(cfg: Dict[str, Any]) -> Dict[str, Tuple[int, ...]]:
    """All state-dict entries of RAFTSpline (models/raft_spline/raft.py:15-73), in module order."""
    out: Dict[str, Tuple[int, ...]] = {}
    hdim, cdim = cfg["hidden"]["dim"], cfg["context"]["dim"]
    fdim = cfg["feature"]["dim"]
    ctx_in = 0
    if cfg["use_boundary_images"]:
        _encoder_shapes("fnet_img", 3, fdim, cfg["feature"]["norm"], out)
        ctx_in += 3
    if cfg["use_events"]:
        _encoder_shapes("fnet_ev", cfg["num_bins"]["correlation"], fdim, cfg["feature"]["norm"], out)
        ctx_in += cfg["num_bins"]["context"]
    _encoder_shapes("cnet", ctx_in, hdim + cdim, cfg["context"]["norm"], out)
    deg2 = 2 * cfg["bezier_degree"]
    mdim = cfg["motion"]["dim"]

    def conv(name, co, ci, kh, kw):
        out[f"update_block.{name}.weight"] = (co, ci, kh, kw)
        out[f"update_block.{name}.bias"] = (co,)

    conv("encoder.convc1", 256, num_corr_planes(cfg), 1, 1)   # update.py:56
    conv("encoder.convc2", 192, 256, 3, 3)                    # update.py:58
    conv("encoder.convf1", 128, deg2, 7, 7)                   # update.py:62
    conv("encoder.convf2", 64, 128, 3, 3)                     # update.py:64
    conv("encoder.conv", mdim - deg2, 64 + 192, 3, 3)         # update.py:67
    gin = hdim + cdim + mdim
    conv("gru.convz1", hdim, gin, 1, 5)
    conv("gru.convr1", hdim, gin, 1, 5)
    conv("gru.convq1", hdim, gin, 1, 5)
    conv("gru.convz2", hdim, gin, 5, 1)
    conv("gru.convr2", hdim, gin, 5, 1)
    conv("gru.convq2", hdim, gin, 5, 1)
    conv("bezier_head.conv1", 256, hdim, 3, 3)
    conv("bezier_head.conv2", deg2, 256, 3, 3)
    conv("mask.0", 256, hdim, 3, 3)
    conv("mask.2", 64 * 9, 256, 1, 1)
    return out


# Per-layer gains that keep the recurrent loop in a numerically meaningful regime with random weights
# (un-saturated GRU state, ~0.3 px/iteration updates at 1/8 resolution so look-ups stay inside the volume).
_LAYER_GAINS = (("update_block.bezier_head.conv2.", 0.1), ("update_block.gru.", 0.2),
                ("update_block.encoder.convc1.", 0.1), ("cnet.conv2.", 0.1))


def make_state_dict(cfg: Dict[str, Any], seed: int = 0, gain: float = 1.0) -> StateDict:
    """Deterministic weights, independent of the torch RNG: numpy RandomState(seed), tensors filled in
    SORTED name order.  Conv weights ~ N(0, sqrt(2/fan_out)) (the encoder's kaiming fan_out init,
    extractor.py:85-87, applied everywhere), biases ~ U(-0.05, 0.05), BN affine near identity, BN running
    stats non-trivial.  `.norm3` and `.downsample.1` of a BatchNorm residual block are the same module in the
    reference (extractor.py:43-44) and therefore get identical values."""
    shapes = param_shapes(cfg)
    rs = np.random.RandomState(seed)
    sd: StateDict = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(100, dtype=torch.int64)
            continue
        if ".downsample.1." in name:
            continue  # aliased below
        leaf = name.rsplit(".", 1)[1]
        if len(shp) == 4:
            fan_out = shp[0] * shp[2] * shp[3]
            g = gain
            for pfx, lg in _LAYER_GAINS:
                if name.startswith(pfx):
                    g = g * lg
            arr = rs.standard_normal(shp) * math.sqrt(2.0 / fan_out) * g
        elif leaf == "running_var":
            arr = rs.uniform(0.5, 1.5, shp)
        elif leaf == "running_mean":
            arr = rs.uniform(-0.1, 0.1, shp)
        elif leaf == "weight":  # BN gamma
            arr = rs.uniform(0.9, 1.1, shp)
        else:  # conv bias / BN beta
            arr = rs.uniform(-0.05, 0.05, shp)
        sd[name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
    for name in shapes:
        if ".downsample.1." in name:
            sd[name] = sd[name.replace(".downsample.1.", ".norm3.")]
    return {k: sd[k] for k in shapes}


# --------------------------------------------------------------------------------------
# K4: feature / context encoder
# --------------------------------------------------------------------------------------
def _norm(sd: StateDict, name: str, x: Tensor, kind: str, training: bool = False) -> Tensor:
    if kind == "instance":  # nn.InstanceNorm2d defaults: eps=1e-5, no affine, no running stats
        return F.instance_norm(x, eps=1e-5)
    if kind == "batch":     # eval: running statistics (Lightning validate -> eval).  train: batch statistics + running-stat update
        #                     (momentum 0.1) -- freeze_bn (raft.py:75-78) is defined but never called by the reference's training
        return F.batch_norm(x, sd[f"{name}.running_mean"], sd[f"{name}.running_var"],
                            sd[f"{name}.weight"], sd[f"{name}.bias"], training=training, momentum=0.1, eps=1e-5)
    if kind == "group":     # extractor.py:13-19 (num_groups = planes // 8) and :63-64 (8 groups of the stem's 64 channels): C // 8 both times
        return F.group_norm(x, x.shape[1] // 8, sd[f"{name}.weight"], sd[f"{name}.bias"], eps=1e-5)
    if kind == "none":
        return x
    raise NotImplementedError(kind)


def _conv(sd: StateDict, name: str, x: Tensor, stride=1, padding=0) -> Tensor:
    return F.conv2d(x, sd[f"{name}.weight"], sd[f"{name}.bias"], stride=stride, padding=padding)


def _residual_block(sd: StateDict, p: str, x: Tensor, kind: str, stride: int, training: bool = False) -> Tensor:
    """models/raft_utils/extractor.py:47-55."""
    y = torch.relu(_norm(sd, f"{p}.norm1", _conv(sd, f"{p}.conv1", x, stride=stride, padding=1), kind, training))
    y = torch.relu(_norm(sd, f"{p}.norm2", _conv(sd, f"{p}.conv2", y, padding=1), kind, training))
    if stride != 1:
        x = _norm(sd, f"{p}.norm3", _conv(sd, f"{p}.downsample.0", x, stride=stride), kind, training)
    return torch.relu(x + y)


def encoder(sd: StateDict, prefix: str, x: Union[Tensor, Sequence[Tensor]], kind: str, training: bool = False):
    """BasicEncoder.forward, models/raft_utils/extractor.py:103-125.  A list input is concatenated along the
    batch axis (:106-110) and split again (:122-123)."""
    is_list = isinstance(x, (list, tuple))
    if is_list:
        nb, length = x[0].shape[0], len(x)
        x = torch.cat(list(x), dim=0)
    x = torch.relu(_norm(sd, f"{prefix}.norm1", _conv(sd, f"{prefix}.conv1", x, stride=2, padding=3), kind, training))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _residual_block(sd, f"{prefix}.layer{li}.0", x, kind, stride, training)
        x = _residual_block(sd, f"{prefix}.layer{li}.1", x, kind, 1, training)
    x = _conv(sd, f"{prefix}.conv2", x)
    if is_list:
        return list(torch.split(x, [nb] * length, dim=0))
    return x


# --------------------------------------------------------------------------------------
# K5 / K6: all-pairs correlation volume + pyramid
# --------------------------------------------------------------------------------------
def corr_volume(fmap1: Tensor, fmap2: Tensor) -> Tensor:
    """CorrComputation._corr_dot_prod_util, models/raft_utils/corr.py:264-272.
    fmap1: (B,D,h,w) [1-to-N, :237-246] or (T,B,D,h,w) [M-to-N, :248-262];  fmap2: (T,B,D,h,w).
    Returns (T, B*h*w, 1, h, w)."""
    T, B, D, h, w = fmap2.shape
    f2 = fmap2.reshape(T, B, D, h * w)
    if fmap1.ndim == 4:
        f1 = fmap1.reshape(B, D, h * w)
    else:
        f1 = fmap1.reshape(T, B, D, h * w)
    corr = f1.transpose(-1, -2) @ f2
    corr = corr / torch.sqrt(torch.tensor(D).float())
    return corr.reshape(T, B * h * w, 1, h, w)


def corr_pyramid(volume: Tensor, levels_per_target: Sequence[int]) -> List[Tuple[Tensor, List[int]]]:
    """CorrBlockParallelMultiTarget.__init__, corr.py:297-305 + CorrData.get_downsampled :108-125.
    Returns [(corr_L (T_L, B*N, 1, h_L, w_L), base-target indices living at level L)] for L = 0..max-1."""
    levels_per_target = [int(v) for v in levels_per_target]
    pyr = [(volume, list(range(volume.shape[0])))]
    for num_levels in range(2, max(levels_per_target) + 1):
        prev, prev_idx = pyr[-1]
        keep = [t for t, lv in enumerate(levels_per_target) if lv >= num_levels]
        sel = prev[[prev_idx.index(t) for t in keep]]
        tn, bhw, _, ht, wd = sel.shape
        down = F.avg_pool2d(sel.reshape(-1, 1, ht, wd), 2, stride=2)   # floor on odd sizes (corr.py:119)
        pyr.append((down.reshape(tn, bhw, 1, down.shape[-2], down.shape[-1]), keep))
    return pyr


def bilinear_sampler(img: Tensor, coords: Tensor) -> Tensor:
    """models/raft_utils/utils.py:5-21: pixel coords -> [-1,1] -> grid_sample(bilinear, zeros, align_corners=True)."""
    H, W = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    xg = 2 * xg / (W - 1) - 1
    yg = 2 * yg / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xg, yg], dim=-1), align_corners=True)


def corr_lookup(pyramid: List[Tuple[Tensor, List[int]]], coords: Union[Tensor, Sequence[Tensor]],
                radius: int = LOOKUP_RADIUS) -> Tensor:
    """CorrBlockParallelMultiTarget.__call__, corr.py:307-351.  coords: (T,B,2,h,w) -> (B, P*(2r+1)^2, h, w).
    Window channel order: x (dx) varies fastest (corr.py:328-331)."""
    if isinstance(coords, (list, tuple)):
        coords = torch.stack(list(coords), dim=0)
    coords = coords.permute(0, 1, 3, 4, 2)
    T, B, h1, w1, _ = coords.shape
    r = radius
    d = torch.linspace(-r, r, 2 * r + 1)
    dyy, dxx = torch.meshgrid(d, d, indexing="ij")
    delta = torch.stack([dxx, dyy], dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)  # [...,0]=x, [...,1]=y
    outs = []
    for lvl, (corr, tidx) in enumerate(pyramid):
        sel = coords[tidx]
        nt = len(tidx)
        centroid = sel.reshape(nt * B * h1 * w1, 1, 1, 2) / 2 ** lvl
        feat = bilinear_sampler(corr.reshape(-1, 1, corr.shape[-2], corr.shape[-1]), centroid + delta)
        outs.append(feat.view(nt, B, h1, w1, -1))
    out = torch.cat(outs, dim=0).permute(1, 0, 4, 2, 3)
    return out.reshape(B, -1, h1, w1).float()


# --------------------------------------------------------------------------------------
# K8 / K12 / K14: Bezier curves
# --------------------------------------------------------------------------------------
def bezier_coeffs(times: Sequence[float], degree: int) -> np.ndarray:
    """(len(times), degree) float64 matrix  C(deg,i) (1-t)^(deg-i) t^i, i=1..deg.
    models/raft_spline/bezier.py:141-163,175-178 (scipy.special.binom == math.comb for these integers)."""
    ts = np.asarray(times, dtype="float64")
    assert ts.size > 0 and ts.min() >= 0 and ts.max() <= 1
    out = np.zeros((ts.size, degree))
    for ti in range(ts.size):
        for di in range(degree):
            i = di + 1
            out[ti, di] = float(math.comb(degree, i)) * ((1 - ts[ti]) ** (degree - i) * ts[ti] ** i)
    return out


def bezier_flow(params: Tensor, time: Union[float, int, Sequence[float]]) -> Tensor:
    """BezierCurves.get_flow_from_reference, bezier.py:188-216 (+ :165-186).
    params (B, 2*deg, h, w), channel = dim*deg + (i-1) (:134-135).  Scalar time -> (B,2,h,w); list -> (T,B,2,h,w)."""
    B, C, h, w = params.shape
    deg = C // 2
    pv = params.view(B, 2, deg, h, w)
    scalar = isinstance(time, (int, float))
    if scalar:
        assert 0.0 <= time <= 1.0
        if time == 1:
            return pv[:, :, -1]
        if time == 0:
            return torch.zeros((B, 2, h, w), dtype=params.dtype)
        time = [time]
    coeffs = torch.from_numpy(bezier_coeffs(time, deg)).float()
    flow = torch.einsum("bdphw,tp->tbdhw", pv, coeffs)
    return flow[0] if scalar else flow


def coords_grid(batch: int, ht: int, wd: int) -> Tensor:
    """models/raft_utils/utils.py:24-30: channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


def cvx_upsample(data: Tensor, mask: Tensor) -> Tensor:
    """models/raft_utils/utils.py:33-48: softmax over the 9 taps, 3x3 unfold of 8*data, pixel shuffle x8."""
    N, dim, H, W = data.shape
    m = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
    up = F.unfold(8 * data, [3, 3], padding=1).view(N, dim, 9, 1, 1, H, W)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, dim, 8 * H, 8 * W)


# --------------------------------------------------------------------------------------
# K9-K11: update block
# --------------------------------------------------------------------------------------
def motion_encoder(sd: StateDict, bezier: Tensor, corr: Tensor) -> Tensor:
    """BasicMotionEncoder.forward, models/raft_spline/update.py:88-97."""
    p = "update_block.encoder"
    cor = torch.relu(_conv(sd, f"{p}.convc1", corr))
    cor = torch.relu(_conv(sd, f"{p}.convc2", cor, padding=1))
    bez = torch.relu(_conv(sd, f"{p}.convf1", bezier, padding=3))
    bez = torch.relu(_conv(sd, f"{p}.convf2", bez, padding=1))
    out = torch.relu(_conv(sd, f"{p}.conv", torch.cat([cor, bez], dim=1), padding=1))
    return torch.cat([out, bezier], dim=1)


def sep_conv_gru(sd: StateDict, h: Tensor, x: Tensor) -> Tensor:
    """SepConvGRU.forward, update.py:33-48: horizontal (1x5) then vertical (5x1) GRU step."""
    p = "update_block.gru"
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], dim=1)
        z = torch.sigmoid(_conv(sd, f"{p}.convz{sfx}", hx, padding=pad))
        r = torch.sigmoid(_conv(sd, f"{p}.convr{sfx}", hx, padding=pad))
        q = torch.tanh(_conv(sd, f"{p}.convq{sfx}", torch.cat([r * h, x], dim=1), padding=pad))
        h = (1 - z) * h + z * q
    return h


def update_block(sd: StateDict, net: Tensor, inp: Tensor, corr: Tensor, bezier: Tensor):
    """BasicUpdateBlock.forward, update.py:116-126 -> (net, mask, delta_bezier)."""
    mf = motion_encoder(sd, bezier, corr)
    net = sep_conv_gru(sd, net, torch.cat([inp, mf], dim=1))
    p = "update_block.bezier_head"
    delta = _conv(sd, f"{p}.conv2", torch.relu(_conv(sd, f"{p}.conv1", net, padding=1)), padding=1)
    m = _conv(sd, "update_block.mask.2", torch.relu(_conv(sd, "update_block.mask.0", net, padding=1)))
    return net, 0.25 * m, delta


# --------------------------------------------------------------------------------------
# RAFTSpline.forward
# --------------------------------------------------------------------------------------
def lookup_times(cfg: Dict[str, Any]) -> List[float]:
    """models/raft_spline/raft.py:156,170-177."""
    ts: List[float] = []
    if cfg["use_events"]:
        dt = 1 / (cfg["num_bins"]["context"] - 1)
        ts += [dt * t for t in cfg["correlation"]["ev"]["target_indices"]]
    if cfg["use_boundary_images"]:
        ts.append(1)
    return ts


def forward(sd: StateDict, cfg: Dict[str, Any], voxel_grid: Optional[Tensor] = None,
            images: Optional[List[Tensor]] = None, iters: int = 12, flow_init: Optional[Tensor] = None,
            test_mode: bool = False, return_intermediates: bool = False, training: bool = False, stage_hook=None):
    """RAFTSpline.forward, models/raft_spline/raft.py:101-200.
    stage_hook(name, begin: bool), optional: called at the boundaries of the reference's CudaTimer stages (raft.py:116-186, same names) --
    the per-stage CPU timing of bench.py's cpu_baseline; it does not touch the arithmetic.
    Returns (bezier_low_params, bezier_up_params) if test_mode else [bezier_up_params per iteration]
    (the reference wraps these tensors in BezierCurves).  training=True: BatchNorm on batch statistics (module.train()); the
    function is plain differentiable torch, so autograd over it is the gradient oracle of the training path."""
    assert voxel_grid is not None or images is not None
    assert iters > 0
    hdim, cdim = cfg["hidden"]["dim"], cfg["context"]["dim"]
    nctx, ncorr = cfg["num_bins"]["context"], cfg["num_bins"]["correlation"]
    fnorm, cnorm = cfg["feature"]["norm"], cfg["context"]["norm"]
    groups = []   # [(fmap1 (B,D,h,w), fmap2 (T,B,D,h,w), levels)]
    context_input = None
    if cfg["use_events"]:
        assert voxel_grid is not None
        voxel_grid = voxel_grid.contiguous()
        assert nctx + ncorr - 1 == voxel_grid.shape[-3]                      # raft.py:90
        idxs = [0] + list(cfg["correlation"]["ev"]["target_indices"])        # raft.py:93-94
        grids = [voxel_grid[:, i:i + ncorr] for i in idxs]
        context_input = voxel_grid[:, -nctx:]
        if stage_hook: stage_hook("fnet_ev", True)
        fm = [x.float() for x in encoder(sd, "fnet_ev", grids, fnorm, training)]
        if stage_hook: stage_hook("fnet_ev", False)
        groups.append((fm[0], torch.stack(fm[1:], dim=0), list(cfg["correlation"]["ev"]["levels"])))
    if cfg["use_boundary_images"]:
        assert len(images) == 2
        images = [2 * (x.float().contiguous() / 255) - 1 for x in images]    # raft.py:134
        if stage_hook: stage_hook("fnet_img", True)
        fi = encoder(sd, "fnet_img", images, fnorm, training)
        if stage_hook: stage_hook("fnet_img", False)
        groups.append((fi[0], fi[1].unsqueeze(0), [int(cfg["correlation"]["img"]["levels"])]))
        context_input = images[0] if context_input is None else torch.cat((context_input, images[0]), dim=-3)
    if stage_hook: stage_hook("cnet", True)
    cnet = encoder(sd, "cnet", context_input, cnorm, training)
    net, inp = torch.split(cnet, [hdim, cdim], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)
    if stage_hook: stage_hook("cnet", False)

    B, _, H, W = context_input.shape
    assert H % 8 == 0 and W % 8 == 0                                         # bezier.py:67-68
    h, w = H // 8, W // 8
    coords0 = coords_grid(B, h, w)
    bezier = torch.zeros(B, 2 * cfg["bezier_degree"], h, w)
    if flow_init is not None:
        bezier = bezier + flow_init

    # corr.py:223-262: 1-to-N for a single reference, M-to-N (fmap1 expanded per target) otherwise
    if stage_hook: stage_hook("corr computation", True)
    if len(groups) == 1:
        volume = corr_volume(groups[0][0], groups[0][1])
    else:
        f1 = torch.cat([g[0].unsqueeze(0).expand(g[1].shape[0], -1, -1, -1, -1) for g in groups], dim=0)
        volume = corr_volume(f1, torch.cat([g[1] for g in groups], dim=0))
    levels = sum((g[2] for g in groups), [])
    pyramid = corr_pyramid(volume, levels)
    if stage_hook: stage_hook("corr computation", False)
    times = lookup_times(cfg)

    ups = []
    inter = []
    bezier_up = None
    if stage_hook: stage_hook("all iters", True)
    for itr in range(iters):
        if stage_hook: stage_hook("1 iter", True)
        if cfg.get("detach_bezier", False):                                  # raft.py:167-168
            bezier = bezier.detach()
        if stage_hook: stage_hook("get_flow (per iter)", True)
        flows = bezier_flow(bezier, times)
        coords1 = coords0 + flows
        if stage_hook: stage_hook("get_flow (per iter)", False); stage_hook("corr lookup (per iter)", True)
        corr_feat = corr_lookup(pyramid, coords1)
        if stage_hook: stage_hook("corr lookup (per iter)", False); stage_hook("update (per iter)", True)
        net, up_mask, delta = update_block(sd, net, inp, corr_feat, bezier)
        bezier = bezier + delta                                              # bezier.py:137-139
        if stage_hook: stage_hook("update (per iter)", False); stage_hook("1 iter", False)
        if return_intermediates:
            inter.append(dict(corr=corr_feat, net=net, delta=delta))
        if (not test_mode) or itr == iters - 1:
            bezier_up = cvx_upsample(bezier, up_mask)                        # bezier.py:81-84
            ups.append(bezier_up)
    if stage_hook: stage_hook("all iters", False)
    if return_intermediates:
        return bezier, bezier_up, dict(volume=volume, pyramid=pyramid, iters=inter, net0=torch.tanh(cnet[:, :hdim]))
    if test_mode:
        return bezier, bezier_up
    return ups


# --------------------------------------------------------------------------------------
# K1 / K2: event voxel grid
# --------------------------------------------------------------------------------------
def voxel_grid_convert(x: Tensor, y: Tensor, pol: Tensor, time: Tensor, channels: int, height: int, width: int,
                       t0_center: Optional[int] = None, t1_center: Optional[int] = None) -> Tensor:
    """VoxelGrid.convert, data/utils/representations.py:64-111.  Float x/y: trilinear over 8 neighbours
    (:96-109); integer x/y: linear over the 2 temporal neighbours (:85-94).  Sequential accumulation."""
    assert x.shape == y.shape == pol.shape == time.shape and x.ndim == 1
    assert not torch.is_floating_point(time)
    is_int_xy = not torch.is_floating_point(x)
    grid = torch.zeros((channels, height, width), dtype=torch.float)
    ch, ht, wd = channels, height, width
    t0c = t0_center if t0_center is not None else time[0]
    t1c = t1_center if t1_center is not None else time[-1]
    t_norm = (time - t0c) / (t1c - t0c) * (ch - 1)                            # :58
    t0 = t_norm.floor().int()
    value = 2 * pol.float() - 1
    if is_int_xy:
        for tl in (t0, t0 + 1):
            m = (tl >= 0) & (tl < ch)
            wgt = value * (1 - (tl - t_norm).abs())
            idx = ht * wd * tl.long() + wd * y.long() + x.long()
            grid.put_(idx[m], wgt[m], accumulate=True)
    else:
        x0, y0 = x.floor().int(), y.floor().int()
        for xl in (x0, x0 + 1):
            for yl in (y0, y0 + 1):
                for tl in (t0, t0 + 1):
                    m = (xl < wd) & (xl >= 0) & (yl < ht) & (yl >= 0) & (tl >= 0) & (tl < ch)
                    wgt = value * (1 - (xl - x).abs()) * (1 - (yl - y).abs()) * (1 - (tl - t_norm).abs())
                    idx = ht * wd * tl.long() + wd * yl.long() + xl.long()
                    grid.put_(idx[m], wgt[m], accumulate=True)
    return grid


def extended_time_window(t0_center: int, t1_center: int, channels: int) -> Tuple[int, int]:
    """VoxelGrid.get_extended_time_window, representations.py:35-39."""
    dt = (t1_center - t0_center) / (channels - 1)
    return math.floor(t0_center - dt), math.ceil(t1_center + dt)


def norm_voxel_grid(grid: Tensor) -> Tensor:
    """representations.py:9-18: zero-mean / unit (unbiased) std over the non-zero entries, in place."""
    mask = torch.nonzero(grid, as_tuple=True)
    if mask[0].size()[0] > 0:
        mean = grid[mask].mean()
        std = grid[mask].std()
        if std > 0:
            grid[mask] = (grid[mask] - mean) / std
        else:
            grid[mask] = grid[mask] - mean
    return grid


# --------------------------------------------------------------------------------------
# K15: end-point error
# --------------------------------------------------------------------------------------
def epe_masked(source: Tensor, target: Tensor, valid_mask: Optional[Tensor] = None) -> Optional[Tensor]:
    """utils/metrics.py:196-213."""
    assert source.ndim > 2 and source.shape == target.shape
    epe = torch.sqrt(torch.square(source - target).sum(1))
    if valid_mask is not None:
        assert valid_mask.dtype == torch.bool and epe.shape == valid_mask.shape
        den = valid_mask.sum()
        if den == 0:
            return None
        return epe[valid_mask].sum() / den
    return torch.mean(epe)


def epe_masked_multi(source_lst, target_lst, valid_mask_lst=None) -> Optional[Tensor]:
    """utils/metrics.py:216-239: mean over the predictions whose mask is non-empty of the per-prediction masked EPE."""
    num_preds = len(source_lst)
    assert num_preds > 0 and len(target_lst) == num_preds
    if valid_mask_lst is not None:
        assert len(valid_mask_lst) == num_preds
    else:
        valid_mask_lst = [None] * num_preds
    epe_sum, den = 0, 0
    for src, tgt, vm in zip(source_lst, target_lst, valid_mask_lst):
        e = epe_masked(src, tgt, vm)
        if e is not None:
            epe_sum = epe_sum + e
            den += 1
    if den == 0:
        return None
    return epe_sum / den


def ae_masked(source: Tensor, target: Tensor, valid_mask: Optional[Tensor] = None, degrees: bool = True) -> Tensor:
    """utils/metrics.py:259-296: angle between (u, v, 1) vectors, clamped cosine, masked mean (no empty-mask guard there)."""
    assert source.ndim > 2 and source.shape == target.shape
    ext_shape = list(source.shape)
    ext_shape[1] = 1
    ext = torch.ones(ext_shape, device=source.device)
    s_ext, t_ext = torch.cat((source, ext), dim=1), torch.cat((target, ext), dim=1)
    nom = torch.sum(s_ext * t_ext, dim=1)
    den = torch.linalg.norm(s_ext, dim=1) * torch.linalg.norm(t_ext, dim=1)
    tmp = torch.div(nom, den)
    tmp[tmp > 1.0] = 1.0
    tmp[tmp < -1.0] = -1.0
    ae = torch.acos(tmp)
    if degrees:
        ae = ae / math.pi * 180
    if valid_mask is not None:
        assert valid_mask.dtype == torch.bool and ae.shape == valid_mask.shape
        return ae[valid_mask].sum() / valid_mask.sum()
    return torch.mean(ae)


def ae_masked_multi(source_lst, target_lst, valid_mask_lst=None, degrees: bool = True) -> Tensor:
    """utils/metrics.py:241-256: plain mean over the predictions."""
    num_preds = len(source_lst)
    assert num_preds > 0 and len(target_lst) == num_preds
    if valid_mask_lst is None:
        valid_mask_lst = [None] * num_preds
    total = 0
    for src, tgt, vm in zip(source_lst, target_lst, valid_mask_lst):
        total = total + ae_masked(src, tgt, vm, degrees)
    return total / num_preds


def n_pixel_error_masked(source: Tensor, target: Tensor, valid_mask: Optional[Tensor], n_pixels: float) -> Tensor:
    """utils/metrics.py:160-193: percentage of (valid) pixels with error > n_pixels AND relative error >= 5 %."""
    assert source.ndim > 2 and source.shape == target.shape
    if valid_mask is not None:
        assert valid_mask.dtype == torch.bool
        num_valid = torch.sum(valid_mask)
        assert num_valid > 0
    gt_magn = torch.linalg.norm(target, dim=1)
    err_magn = torch.linalg.norm(source - target, dim=1)
    if valid_mask is not None:
        rel = torch.zeros_like(err_magn)
        rel[valid_mask] = err_magn[valid_mask] / torch.clip(gt_magn[valid_mask], min=1e-6)
    else:
        rel = err_magn / torch.clip(gt_magn, min=1e-6)
    emap = (err_magn > n_pixels) & (rel >= 0.05)
    if valid_mask is not None:
        err = emap[valid_mask].sum() / num_valid
    else:
        err = torch.mean(emap.float())
    return err * 100


def compute_traj_len(target_lst) -> Tensor:
    """EPE_MULTI.compute_traj_len, utils/metrics.py:60-64: summed length of the ground-truth polyline per pixel, (N, *)."""
    st = torch.stack(list(target_lst), dim=0)
    diff = st[1:] - st[:-1]
    return diff.square().sum(dim=2).sqrt().sum(dim=0)


def predictions_from_lin_assumption(source: Tensor, target_timestamps) -> list:
    """utils/metrics.py:298-305."""
    assert max(target_timestamps) <= 1 and 0 <= min(target_timestamps)
    return [ts * source for ts in target_timestamps]


def input_pad_amounts(ht: int, wd: int, min_size: int = 8, no_top_padding: bool = False):
    """InputPadder.pad's padding list [left, right, top, bottom] (modules/utils.py:63-73)."""
    pad_ht = (((ht // min_size) + 1) * min_size - ht) % min_size
    pad_wd = (((wd // min_size) + 1) * min_size - wd) % min_size
    if no_top_padding:
        return [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]
    return [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]


def input_pad(x: Tensor, pad) -> Tensor:
    """InputPadder.pad (modules/utils.py:78): replicate padding of the last two dims."""
    return F.pad(x, list(pad), mode="replicate")


def input_unpad(x: Tensor, pad) -> Tensor:
    """InputPadder.unpad (modules/utils.py:80-83)."""
    ht, wd = x.shape[-2:]
    return x[..., pad[2]:ht - pad[3], pad[0]:wd - pad[1]]


# --------------------------------------------------------------------------------------
# DSEC two-step sample assembly (data/dsec/subsequence/{base,twostep}.py, data/dsec/eventslicer.py)
# --------------------------------------------------------------------------------------
def rectify_events(rectify_map: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """BaseSubSequence._rectify_events, base.py:137-143: (n, 2) rectified (x', y') of raw integer sensor coordinates."""
    H, W = rectify_map.shape[:2]
    assert rectify_map.shape == (H, W, 2)
    assert x.max() < W and y.max() < H
    return rectify_map[y, x]


def event_window_indices(time_array: np.ndarray, time_start_us: int, time_end_us: int) -> Tuple[int, int]:
    """EventSlicer.get_time_indices_offsets, eventslicer.py:99-158: [i0, i1) with time_start <= t[i0:i1] < time_end on a sorted
    array (the reference scans linearly; same result as two left-sided binary searches)."""
    assert time_array.ndim == 1
    if time_array[-1] < time_start_us:
        return time_array.size, time_array.size
    return int(np.searchsorted(time_array, time_start_us, side="left")), int(np.searchsorted(time_array, time_end_us, side="left"))


def twostep_windows(forward_flow_timestamps: np.ndarray, index: int) -> List[Tuple[int, int]]:
    """TwoStepSubSequence.__getitem__, twostep.py:49-66: (ts_from, ts_to) of the CURRENT flow interval and of the one before it
    (extrapolated backwards by the same duration when `index` is the first sample)."""
    out = []
    ts_from = ts_to = None
    for idx in (index, index - 1):
        if 0 <= idx < len(forward_flow_timestamps):
            ts_from, ts_to = int(forward_flow_timestamps[idx][0]), int(forward_flow_timestamps[idx][1])
        else:
            assert idx == index - 1 and ts_from is not None and ts_to is not None
            dt = ts_to - ts_from
            ts_to = ts_from
            ts_from = ts_from - dt
        out.append((ts_from, ts_to))
    return out


def construct_voxel_grid(events: Dict[str, np.ndarray], rectify_map: np.ndarray, num_bins: int, H: int, W: int, ts_from: int, ts_to: int) -> Tensor:
    """BaseSubSequence._construct_voxel_grid (version 1 = extended window) + _get_events + _events_to_voxel_grid, base.py:121-204.
    events: raw stream {'x','y' uint16, 'p' uint8, 't' int64 sorted}; stream start / final time = t[0] / t[-1]."""
    dt = (ts_to - ts_from) / (num_bins - 1)
    t_start, t_end = math.floor(ts_from - dt), math.ceil(ts_to + dt)            # representations.py:35-39
    assert (ts_from - t_start) < 50000 and (t_end - ts_to) < 50000
    t_all = events["t"]
    start_us, final_us = int(t_all[0]), int(t_all[-1])
    assert t_start > start_us - 50000 and t_end < final_us + 50000               # base.py:168-169
    t_start, t_end = max(t_start, start_us), min(t_end, final_us)                # base.py:170-175
    assert t_start < t_end
    i0, i1 = event_window_indices(t_all, t_start, t_end)
    x, y, p, t = events["x"][i0:i1], events["y"][i0:i1], events["p"][i0:i1], t_all[i0:i1]
    xy = rectify_events(rectify_map, x, y)
    return voxel_grid_convert(torch.from_numpy(xy[:, 0].astype("float32")), torch.from_numpy(xy[:, 1].astype("float32")),
                              torch.from_numpy(p.astype("float32")), torch.from_numpy(t.astype("int64")), num_bins, H, W, ts_from, ts_to)


def twostep_merge(ev_prev: Tensor, ev_cur: Tensor, normalize: bool = True, merge: bool = True) -> Tensor:
    """twostep.py:79-92: drop the temporal slice the two grids share (after checking that they agree), normalise the merged grid."""
    if merge:
        assert (ev_prev[-1] - ev_cur[0]).flatten().abs().max() < 0.5
        out = torch.cat((ev_prev, ev_cur[1:, ...]), dim=0)
        return norm_voxel_grid(out) if normalize else out
    grids = [norm_voxel_grid(g) for g in (ev_prev, ev_cur)] if normalize else [ev_prev, ev_cur]
    return torch.stack(grids)


def dsec_twostep_sample(events: Dict[str, np.ndarray], rectify_map: np.ndarray, forward_flow_timestamps: np.ndarray, index: int,
                        num_bins: int, H: int, W: int, normalize: bool = True, merge: bool = True) -> Tensor:
    """The EV_REPR entry of TwoStepSubSequence.__getitem__ (twostep.py:44-100) built from the raw event stream."""
    (cf, ct), (pf, pt) = twostep_windows(forward_flow_timestamps, index)
    cur = construct_voxel_grid(events, rectify_map, num_bins, H, W, cf, ct)
    prev = construct_voxel_grid(events, rectify_map, num_bins, H, W, pf, pt)
    return twostep_merge(prev, cur, normalize, merge)


# --------------------------------------------------------------------------------------
# Training losses: utils/losses.py
# --------------------------------------------------------------------------------------
def l1_loss_channel_masked(source: Tensor, target: Tensor, valid_mask: Optional[Tensor] = None) -> Tensor:
    """utils/losses.py:6-22: |source - target| summed over channels, averaged over the valid positions (all if no mask)."""
    assert source.ndim > 2 and source.shape == target.shape
    per_pos = (source - target).abs().sum(dim=1)
    if valid_mask is None:
        return per_pos.mean()
    assert valid_mask.dtype == torch.bool and valid_mask.shape == per_pos.shape
    return per_pos[valid_mask].sum() / valid_mask.sum()


def l1_seq_loss_channel_masked(source_list, target: Tensor, valid_mask: Optional[Tensor] = None, gamma: float = 0.8):
    """utils/losses.py:24-40: prediction i of I weighted by gamma^(I-1-i)."""
    total = 0
    count = len(source_list)
    for i, src in enumerate(source_list):
        total = total + gamma ** (count - i - 1) * l1_loss_channel_masked(src, target, valid_mask)
    return total


def l1_multi_seq_loss_channel_masked(src_list_list, target_list, valid_mask_list=None, gamma: float = 0.8):
    """utils/losses.py:42-62: per iteration the mean loss over the M supervision targets, iterations weighted as above."""
    total = 0
    n_iters = len(src_list_list)
    for it, per_iter in enumerate(src_list_list):
        assert len(per_iter) > 0 and len(per_iter) == len(target_list)
        acc = 0
        for m, src in enumerate(per_iter):
            acc = acc + l1_loss_channel_masked(src, target_list[m], None if valid_mask_list is None else valid_mask_list[m])
        total = total + gamma ** (n_iters - it - 1) * (acc / len(per_iter))
    return total


# --------------------------------------------------------------------------------------
# Configs of BASELINE.json
# --------------------------------------------------------------------------------------
def model_config(name: str) -> Dict[str, Any]:
    """The `config['model']` dict Hydra composes for the four shipped experiments
    (config/model/{base,raft_base,raft-spline}.yaml + config/experiment/**), with num_bins.correlation
    back-filled as the DataModule does (modules/data_loading.py:63-68)."""
    base = dict(name="raft-spline", detach_bezier=False, use_gma=False, num_iter=dict(train=12, test=12),
                hidden=dict(dim=128), context=dict(dim=128, norm="batch"),
                feature=dict(dim=256, norm="instance"), motion=dict(dim=128))
    if name in ("E_LU4_BD2", "E_I_LU4_BD2"):
        img = name.startswith("E_I")
        base.update(num_bins=dict(context=5, correlation=5), bezier_degree=2, use_boundary_images=img,
                    use_events=True,
                    correlation=dict(use_cosine_sim=False,
                                     ev=dict(target_indices=[1, 2, 3, 4], levels=[1, 1, 1, 4], radius=[4, 4, 4, 4]),
                                     img=dict(levels=4 if img else None, radius=4 if img else None)))
    elif name in ("E_LU5_BD10", "E_I_LU5_BD10"):
        img = name.startswith("E_I")
        base.update(num_bins=dict(context=41, correlation=25), bezier_degree=10, use_boundary_images=img,
                    use_events=True,
                    correlation=dict(use_cosine_sim=False,
                                     ev=dict(target_indices=[8, 16, 24, 32, 40], levels=[1, 1, 1, 1, 4],
                                             radius=[4, 4, 4, 4, 4]),
                                     img=dict(levels=4, radius=4)))
    else:
        raise KeyError(name)
    return base
