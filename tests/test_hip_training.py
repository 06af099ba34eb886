"""Training path on the GPU: the hand-written adjoints (csrc/backward.hip) against autograd over the CPU oracle,
and the full training step (forward + sequence loss + backward) against the gradients of the REFERENCE itself
(tests/golden/train_*.npz, losses.npz).  Floating point: tolerances are written at every comparison."""
import os

import numpy as np
import pytest
import torch

import bflow_amd
from bflow_amd import hip, training
from bflow_amd.bezier import BezierCurves
from bflow_amd.corr import CorrBlockParallelMultiTarget, CorrComputation
from bflow_amd.validation import DataLoading, DataSetType
from oracle import raft_spline_oracle as O

import train_common as TC

pytestmark = pytest.mark.gpu
DEV = "cuda"


def g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _lookup_case(seed, B, D, h, w, T):
    rs = np.random.RandomState(seed)
    f1 = rs.standard_normal((B, D, h, w)).astype(np.float32)
    f2 = rs.standard_normal((T, B, D, h, w)).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    base = np.stack([xs, ys], 0).astype(np.float32)
    # generic fractional positions (the bilinear kernel is not differentiable AT integer coordinates), some far outside the plane
    flow = (rs.standard_normal((T, B, 2, h, w)) * 3.0 + 0.37).astype(np.float32)
    flow[:, :, :, 0, 0] = -50.3
    flow[:, :, :, -1, -1] = 40.7
    return f1, f2, (base[None, None] + flow).astype(np.float32)


@pytest.mark.parametrize("levels,shape", [([1, 2, 3], (2, 32, 10, 12)), ([1, 1, 1, 4], (1, 64, 17, 26))])
def test_lookup_and_pool_backward_match_oracle_autograd(levels, shape):
    B, D, h, w = shape
    T = len(levels)
    f1, f2, coords = _lookup_case(11 + T, B, D, h, w, T)
    rs = np.random.RandomState(5)
    # ---- oracle: autograd through avg_pool2d pyramid + grid_sample look-up
    vol = O.corr_volume(torch.from_numpy(f1), torch.from_numpy(f2)).detach().requires_grad_(True)
    co = torch.from_numpy(coords).clone().requires_grad_(True)
    out = O.corr_lookup(O.corr_pyramid(vol, levels), co)
    gout = rs.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(gout))
    # ---- HIP: forward block, adjoint kernels
    blk = CorrBlockParallelMultiTarget(corr_computation_events=CorrComputation(cu(f1), cu(f2), num_levels_per_target=levels))
    np.testing.assert_allclose(blk(cu(coords)).cpu().numpy(), out.detach().numpy(), rtol=1e-5, atol=5e-5)
    grads = [torch.zeros_like(t) for t, _ in blk._pyramid]
    per_plane = [grads[lvl][k] for lvl, (_, idx) in enumerate(blk._pyramid) for k in range(len(idx))]
    table = hip.make_grad_table(per_plane)
    gc = hip.corr_lookup_bwd(blk._table, table, cu(coords), cu(gout))
    gcoords = torch.zeros((T, B, 2, h, w), device=DEV)
    for k, p in enumerate(blk._planes):
        gcoords[p["target"]] += gc[k]
    scale = float(co.grad.abs().max())
    # coordinate gradient: differences of volume values (O(6), fp32) times 81 x P window weights
    assert float((gcoords.cpu() - co.grad).abs().max()) < 2e-4 * scale
    for lvl in range(len(blk._pyramid) - 1, 0, -1):
        _, idx = blk._pyramid[lvl]
        _, prev = blk._pyramid[lvl - 1]
        for k, t in enumerate(idx):
            hip.corr_pool2x2_bwd(grads[lvl][k], grads[lvl - 1][prev.index(t)])
    gv = grads[0].cpu().view(vol.grad.shape)
    assert float((gv - vol.grad).abs().max()) < 1e-5 * float(vol.grad.abs().max()) + 1e-6
    # deterministic: a second pass adds exactly the same numbers
    grads2 = [torch.zeros_like(t) for t, _ in blk._pyramid]
    table2 = hip.make_grad_table([grads2[lvl][k] for lvl, (_, idx) in enumerate(blk._pyramid) for k in range(len(idx))])
    gc2 = hip.corr_lookup_bwd(blk._table, table2, cu(coords), cu(gout))
    assert torch.equal(gc, gc2)


def test_lookup_bezier_backward_matches_oracle_autograd():
    """Fused Bezier evaluation + look-up: gradient w.r.t. the Bezier parameters through the autograd Function of the product."""
    B, D, h, w, deg = 2, 32, 12, 16, 3
    levels, times = [1, 1, 3], [0.25, 0.5, 1.0]
    rs = np.random.RandomState(21)
    f1 = rs.standard_normal((B, D, h, w)).astype(np.float32)
    f2 = rs.standard_normal((3, B, D, h, w)).astype(np.float32)
    params = (rs.standard_normal((B, 2 * deg, h, w)) * 2.0 + 0.21).astype(np.float32)
    # oracle
    a, b = torch.from_numpy(f1).requires_grad_(True), torch.from_numpy(f2).requires_grad_(True)
    p = torch.from_numpy(params).requires_grad_(True)
    coords = O.coords_grid(B, h, w) + O.bezier_flow(p, times)
    out = O.corr_lookup(O.corr_pyramid(O.corr_volume(a, b), levels), coords)
    gout = rs.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(gout))
    # product
    ga, gb, gp = cu(f1).requires_grad_(True), cu(f2).requires_grad_(True), cu(params).requires_grad_(True)
    blk = training.TrainCorrBlock([(ga, gb, levels)])
    from bflow_amd.bezier import polynomial_coefficients
    coef = polynomial_coefficients(np.asarray(times, dtype="float64"), deg)
    res = blk.lookup_bezier(gp, coef)
    np.testing.assert_allclose(res.detach().cpu().numpy(), out.detach().numpy(), rtol=1e-5, atol=5e-5)
    res.backward(cu(gout))
    for mine, ref in ((gp.grad, p.grad), (ga.grad, a.grad), (gb.grad, b.grad)):
        assert float((mine.cpu() - ref).abs().max()) < 3e-4 * float(ref.abs().max())


def test_cvx_upsample_backward_matches_oracle_autograd():
    rs = np.random.RandomState(31)
    B, C, h, w = 2, 4, 6, 9
    data = rs.standard_normal((B, C, h, w)).astype(np.float32)
    mask = (rs.standard_normal((B, 576, h, w)) * 2).astype(np.float32)
    d, m = torch.from_numpy(data).requires_grad_(True), torch.from_numpy(mask).requires_grad_(True)
    up = O.cvx_upsample(d, m)
    gup = rs.standard_normal(tuple(up.shape)).astype(np.float32)
    up.backward(torch.from_numpy(gup))
    gd, gm = cu(data).requires_grad_(True), cu(mask).requires_grad_(True)
    mine = training.cvx_upsample(gd, gm)
    np.testing.assert_allclose(mine.detach().cpu().numpy(), up.detach().numpy(), rtol=1e-5, atol=1e-5)
    mine.backward(cu(gup))
    assert float((gd.grad.cpu() - d.grad).abs().max()) < 1e-5 * float(d.grad.abs().max())
    assert float((gm.grad.cpu() - m.grad).abs().max()) < 1e-5 * float(m.grad.abs().max())


def test_losses_match_reference_golden(golden_dir):
    d = g(golden_dir, "losses")
    srcs = [cu(d[f"src{i}"]).requires_grad_(True) for i in range(4)]
    tgt, valid = cu(d["tgt"]), cu(d["valid"])
    assert abs(float(training.l1_loss_channel_masked(srcs[0], tgt, valid).detach()) - float(d["l1_masked"])) < 1e-6
    assert abs(float(training.l1_loss_channel_masked(srcs[0], tgt).detach()) - float(d["l1_unmasked"])) < 1e-6
    for tag, m, gamma in (("seq_masked", valid, 0.8), ("seq_unmasked", None, 0.8), ("seq_masked_g085", valid, 0.85)):
        for s in srcs:
            s.grad = None
        loss = training.l1_seq_loss_channel_masked(srcs, tgt, m, gamma=gamma)
        loss.backward()
        assert abs(float(loss) - float(d[tag])) < 1e-5      # fp64 accumulation here, fp32 tree sum in the reference
        for i, s in enumerate(srcs):
            assert np.abs(s.grad.cpu().numpy() - d[f"{tag}_grad{i}"]).max() < 1e-8
    tgts = [cu(d[f"mtgt{m}"]) for m in range(3)]
    valids = [cu(d[f"mvalid{m}"]) for m in range(3)]
    multi = [[cu(d[f"msrc{it}_{m}"]) for m in range(3)] for it in range(2)]
    assert abs(float(training.l1_multi_seq_loss_channel_masked(multi, tgts, valids)) - float(d["multi_masked"])) < 1e-5
    assert abs(float(training.l1_multi_seq_loss_channel_masked(multi, tgts)) - float(d["multi_unmasked"])) < 1e-5


def _product_model(cfg):
    model = bflow_amd.RAFTSpline(cfg)
    model.load_state_dict(O.make_state_dict(cfg, seed=0))
    return model.to(DEV).train()


@pytest.mark.parametrize("name", TC.TRAIN_CASES)
def test_training_step_gradients_match_reference_golden(golden_dir, name):
    """model.train() forward (test_mode=False), sequence loss and backward on the GPU against the reference's own loss, parameter
    gradients and BatchNorm running statistics.  fp32 MIOpen convolutions vs fp32 CPU convolutions: 1e-2 of each parameter's
    gradient scale (the check is per element of a strided subsample and per parameter norm)."""
    d = g(golden_dir, name)
    B, H, W, iters, kind = int(d["B"]), int(d["H"]), int(d["W"]), int(d["iters"]), str(d["kind"])
    cfg, init = TC.case_setup(name, O.model_config(str(d["config"])), B, H, W)
    model = _product_model(cfg)
    vox, imgs = TC.inputs(cfg, B, H, W)
    gts, valids, times = TC.train_targets(B, H, W, kind)
    preds = model(voxel_grid=vox.to(DEV), images=None if imgs is None else [i.to(DEV) for i in imgs], iters=iters,
                  flow_init=None if init is None else BezierCurves(cu(init)), test_mode=False)
    assert len(preds) == iters
    if kind == "dsec":
        loss = training.l1_seq_loss_channel_masked([p.get_flow_from_reference(1.0) for p in preds], cu(gts[0]), cu(valids[0]))
    else:
        flows = [[p.get_flow_from_reference(t) for t in times] for p in preds]
        loss = training.l1_multi_seq_loss_channel_masked(flows, [cu(x) for x in gts])
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) < 1e-4 * abs(float(d["loss"]))
    assert np.abs(preds[-1].get_params().detach()[:, :, ::4, ::4].cpu().numpy() - d["last_params_sub"]).max() < 1e-3
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
    worst = TC.check_grads(grads, d, rel=2e-3)
    print(f"{name}: worst scaled gradient error {worst:.2e}")
    for k, v in model.named_buffers():
        if f"buf/{k}" in d:
            assert np.abs(v.cpu().numpy() - d[f"buf/{k}"]).max() < 1e-4


def test_freeze_bn_then_training_step_matches_oracle():
    """The reference API `RAFTSpline.freeze_bn()` (raft.py:75-78) followed by a training forward + backward (round-3 advisor finding: the
    eval-mode BatchNorm layers of cnet used to raise).  Frozen BatchNorm = per-channel affine from the running statistics; the oracle with
    training=False is exactly that network under autograd.  Also covers a crop whose 1/8-resolution plane is not 16-byte aligned
    ((H/8)*(W/8) % 4 != 0: the statistics kernel's fast path does not apply) and a FROZEN feature encoder (requires_grad False on every fnet
    parameter: its convolutions run with nothing requiring grad).  Loss 1e-4 relative, gradients 2e-3 of each parameter's scale, running
    statistics untouched."""
    cfg = O.model_config("E_LU4_BD2")
    B, H, W, iters = 1, 136, 152, 2                     # 17 x 19 = 323 pixels at 1/8: not a multiple of 4 (and every pyramid level >= 2 x 2:
                                                        # on a 1 x 1 level the reference's own normalisation divides by zero)
    model = _product_model(cfg)
    model.freeze_bn()
    for p in model.fnet_ev.parameters():
        p.requires_grad_(False)
    bufs0 = {k: v.clone() for k, v in model.named_buffers()}
    vox, _ = TC.inputs(cfg, B, H, W)
    gts, valids, _ = TC.train_targets(B, H, W, "dsec")
    preds = model(voxel_grid=vox.to(DEV), iters=iters, test_mode=False)
    loss = training.l1_seq_loss_channel_masked([p.get_flow_from_reference(1.0) for p in preds], cu(gts[0]), cu(valids[0]))
    loss.backward()
    for k, v in model.named_buffers():
        assert torch.equal(v, bufs0[k]), k                # frozen: running statistics and num_batches_tracked do not move
    sd = {k: v.clone() for k, v in O.make_state_dict(cfg, seed=0).items()}
    train_keys = [k for k, p in model.named_parameters() if p.requires_grad]
    for k in train_keys:
        sd[k].requires_grad_(True)
    ups = O.forward(sd, cfg, vox, None, iters=iters, test_mode=False, training=False)
    rloss = O.l1_seq_loss_channel_masked([O.bezier_flow(u, 1.0) for u in ups], torch.from_numpy(gts[0]), torch.from_numpy(valids[0]))
    rloss.backward()
    assert abs(float(loss) - float(rloss)) < 1e-4 * abs(float(rloss)), (float(loss), float(rloss))
    worst = 0.0
    for k, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, k
            continue
        want = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        scale = float(want.abs().max()) + 1e-12
        e = float((got.cpu() - want).abs().max()) / scale
        worst = max(worst, e)
        assert e < 2e-3 or scale < 1e-9, (k, e, scale)
    print(f"freeze_bn + frozen fnet training step: worst scaled gradient error {worst:.2e}")


def test_train_step_and_optimizer():
    """TrainStep (training_step without Lightning) + AdamW/OneCycleLR as configure_optimizers builds them: two steps run, the loss
    is finite and every parameter with a gradient moves."""
    cfg = O.model_config("E_LU4_BD2")
    model = _product_model(cfg)
    B, H, W = 1, 128, 160
    vox, _ = TC.inputs(cfg, B, H, W)
    gts, valids, _ = TC.train_targets(B, H, W, "dsec")
    batch = {DataLoading.FLOW: cu(gts[0]), DataLoading.FLOW_VALID: cu(valids[0]), DataLoading.EV_REPR: vox.to(DEV),
             DataLoading.DATASET_TYPE: [DataSetType.DSEC]}
    step = training.TrainStep(model, num_iter_train=2)
    opt, sch = training.configure_optimizers(model, dict(learning_rate=1e-4, weight_decay=1e-4,
                                                         lr_scheduler=dict(use=True, total_steps=10, pct_start=0.3)))
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    losses = []
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        out = step(batch)
        out["loss"].backward()
        opt.step()
        sch.step()
        losses.append(float(out["loss"]))
    assert all(np.isfinite(losses))
    moved = sum(int(not torch.equal(before[k], p.detach())) for k, p in model.named_parameters())
    assert moved > 100
    assert out["pred"].shape == (B, 2, H, W) and out["bezier_prediction"].get_params().requires_grad is False


def test_engine_gradients_follow_the_weights_across_optimizer_steps():
    """The packed (forward) and flipped (backward) filters of conv_train are cached per parameter version: after optimiser steps the
    gradients on the conv engine must still equal the ones torch convolutions compute from the SAME updated weights."""
    import copy
    from bflow_amd import conv_train as CT
    cfg = O.model_config("E_LU4_BD2")
    model = _product_model(cfg)
    B, H, W = 1, 96, 128
    vox, _ = TC.inputs(cfg, B, H, W)
    gts, valids, _ = TC.train_targets(B, H, W, "dsec")
    batch = {DataLoading.FLOW: cu(gts[0]), DataLoading.FLOW_VALID: cu(valids[0]), DataLoading.EV_REPR: vox.to(DEV),
             DataLoading.DATASET_TYPE: [DataSetType.DSEC]}
    step = training.TrainStep(model, num_iter_train=3)
    # (a small step: lr = 1e-3 on this random-init net drives the GRU into saturation within three steps, where ANY two fp32 convolution
    #  implementations differ by 1e-2 in the gradients -- measured for torch vs the engine with either weight-gradient path)
    opt = torch.optim.SGD(model.parameters(), lr=2e-5)
    for _ in range(3):                                       # three updates through the engine's own gradients
        opt.zero_grad(set_to_none=True)
        step(batch)["loss"].backward()
        opt.step()
    opt.zero_grad(set_to_none=True)
    loss_hip = step(batch)["loss"]
    loss_hip.backward()
    g_hip = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    ref = copy.deepcopy(model)                               # same weights / BatchNorm state, torch (MIOpen) convolutions
    # the deep copy shares nothing; undo the BatchNorm statistics update of the step just taken on `model` by copying BEFORE it is irrelevant:
    # gradients in train mode use batch statistics, not the running ones
    ref.zero_grad(set_to_none=True)
    try:
        CT.ENABLED = False
        loss_ref = training.TrainStep(ref, num_iter_train=3)(batch)["loss"]
        loss_ref.backward()
    finally:
        CT.ENABLED = True
    assert abs(float(loss_hip) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))
    worst = 0.0
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for k, p in ref.named_parameters():
        if p.grad is None:
            continue
        # (the bias of a convolution in front of a normalisation layer has an exactly zero gradient: what both paths return there is
        #  round-off at 1e-8 -- scale floors at 1e-5 of the largest gradient)
        scale = max(float(p.grad.abs().max()), 1e-5 * gmax)
        e = float((g_hip[k] - p.grad).abs().max()) / scale
        if e > 2e-3:
            print(f"   {k}: {e:.2e} (scale {scale:.2e})")
        worst = max(worst, e)
    print(f"engine vs torch convolutions after 3 optimiser steps: worst scaled gradient difference {worst:.2e}")
    # batch-1 BatchNorm in training mode and a saturating GRU make this comparison 10x looser than the golden single-step tests above
    assert worst < 2e-2


@pytest.mark.parametrize("tg,B,D,h,w", [(3, 2, 64, 9, 12), (1, 1, 256, 7, 11), (4, 3, 256, 36, 48)])
def test_volume_adjoint_on_the_engine_vs_fp64(tg, B, D, h, w):
    """training._volume_adjoint_engine (the adjoint of corr.py:264-272 as batched 1x1 convolutions with one filter per image, volume gradient
    pre-scaled by a device-side power of two) against the two fp64 GEMMs per target; N = h*w not a multiple of 32 in the small cases; the
    volume gradient has the magnitude it has in training (1e-7)."""
    torch.manual_seed(4)
    N = h * w
    dC = torch.randn(tg, B, N, N, device=DEV) * 1e-7
    f1, f2 = torch.randn(B, D, N, device=DEV), torch.randn(tg, B, D, N, device=DEV)
    s = 1.0 / np.sqrt(D)
    g1, g2 = training._volume_adjoint_engine(dC, f1, f2, s)
    r1 = (torch.matmul(f2.double(), dC.double().transpose(-1, -2)).sum(dim=0) * s)
    r2 = torch.matmul(f1.double().unsqueeze(0), dC.double()) * s
    e1 = float((g1.double() - r1).abs().max() / r1.abs().max())
    e2 = float((g2.double() - r2).abs().max() / r2.abs().max())
    print(f"volume adjoint tg={tg} B={B} D={D} N={N}: relative errors {e1:.2e} {e2:.2e}")
    assert g1.shape == (B, D, N) and g2.shape == (tg, B, D, N)
    assert e1 < 2e-6 and e2 < 2e-6


@pytest.mark.parametrize("kind,relu", [("instance", True), ("instance", False), ("batch", True), ("batch", False)])
def test_norm_train_kernels_match_torch_autograd_fp64(kind, relu):
    """norm_train.norm_act (csrc/norm_train.hip: InstanceNorm2d / training-mode BatchNorm2d, optional ReLU, forward + two-pass backward, running
    statistics) against the torch modules under fp64 autograd: values, input / affine gradients, running mean / variance, batch counter."""
    from bflow_amd import norm_train as NT
    torch.manual_seed(7)
    B, C, H, W = 3, 48, 10, 14
    x = (torch.randn(B, C, H, W, device=DEV) * 2 + 0.5).requires_grad_()
    wgt = torch.randn(B, C, H, W, device=DEV)
    mk = (lambda: torch.nn.InstanceNorm2d(C)) if kind == "instance" else (lambda: torch.nn.BatchNorm2d(C))
    m, ref = mk().to(DEV).train(), mk().to(DEV).double().train()
    if kind == "batch":
        with torch.no_grad():
            m.weight.copy_(torch.rand(C) + 0.5)
            m.bias.copy_(torch.randn(C) * 0.3)
            ref.weight.copy_(m.weight.double())
            ref.bias.copy_(m.bias.double())
    assert NT._supported(m, x) is not None
    for _ in range(2):                                            # two steps: the running statistics are updated twice
        y = NT.norm_act(m, x, relu)
    x.grad = None
    m.zero_grad()
    y = NT.norm_act(m, x, relu)
    (y * wgt).sum().backward()
    x64 = x.detach().double().requires_grad_()
    for _ in range(2):
        y64 = ref(x64)
        y64 = torch.relu(y64) if relu else y64
    x64.grad = None
    ref.zero_grad()
    y64 = ref(x64)
    y64 = torch.relu(y64) if relu else y64
    (y64 * wgt.double()).sum().backward()
    assert float((y.double() - y64).abs().max()) < 5e-6
    e = float((x.grad.double() - x64.grad).abs().max() / x64.grad.abs().max())
    assert e < 5e-6, e
    if kind == "batch":
        for a, b, name in ((m.weight.grad, ref.weight.grad, "dgamma"), (m.bias.grad, ref.bias.grad, "dbeta"), (m.running_mean, ref.running_mean, "running_mean"),
                           (m.running_var, ref.running_var, "running_var")):
            assert float((a.double() - b).abs().max() / b.abs().max()) < 5e-6, name
        assert int(m.num_batches_tracked) == int(ref.num_batches_tracked) == 3


def test_gru_gate_kernels_match_autograd_fp64():
    """csrc/gru_gates.hip (bflow_gru_zr_fwd / _bwd, bflow_gru_blend_fwd / _bwd) through their autograd Functions against the reference's
    chain of element-wise ops (update.py:38-47) under fp64 autograd: values to 2e-6, gradients to 2e-6 of their largest entry."""
    torch.manual_seed(3)
    B, C, H, W = 2, 128, 9, 14
    zr = (torch.randn(B, 2 * C, H, W, device=DEV) * 2).requires_grad_()
    h = torch.randn(B, C, H, W, device=DEV).requires_grad_()
    qp = (torch.randn(B, C, H, W, device=DEV) * 2).requires_grad_()
    w1, w2 = torch.randn(B, C, H, W, device=DEV), torch.randn(B, C, H, W, device=DEV)
    z, rh = training._GruZRFn.apply(zr, h)
    hn = training._GruBlendFn.apply(qp + 0.5 * rh, z, h)           # q_pre depends on r*h as in the GRU (through a convolution there)
    ((hn * w1).sum() + (rh * w2).sum()).backward()
    zr64, h64, qp64 = (t.detach().double().requires_grad_() for t in (zr, h, qp))
    z_ref, r_ref = torch.sigmoid(zr64[:, :C]), torch.sigmoid(zr64[:, C:])
    rh_ref = r_ref * h64
    q_ref = torch.tanh(qp64 + 0.5 * rh_ref)
    hn_ref = (1 - z_ref) * h64 + z_ref * q_ref
    ((hn_ref * w1.double()).sum() + (rh_ref * w2.double()).sum()).backward()
    assert float((z.double() - z_ref).abs().max()) < 2e-6 and float((rh.double() - rh_ref).abs().max()) < 2e-6
    assert float((hn.double() - hn_ref).abs().max()) < 2e-6
    for got, ref, name in ((zr.grad, zr64.grad, "zr_pre"), (h.grad, h64.grad, "h"), (qp.grad, qp64.grad, "q_pre")):
        e = float((got.double() - ref).abs().max() / ref.abs().max())
        assert e < 2e-6, (name, e)


def test_graphed_train_step_matches_eager_steps():
    """training.GraphedTrainStep: forward + loss + backward + AdamW recorded as ONE hipGraph and replayed must train like the eager
    step -- same losses step by step (the filter packs are re-built INSIDE the graph from the weights AdamW just wrote, the learning
    rate is read from the tensor the scheduler fills), and different batches of one signature go through the same graph."""
    from bflow_amd import configs, synthetic
    from bflow_amd.weights import deterministic_state_dict
    cfg = configs.model_config("E_LU4_BD2")
    B, H, W = 2, 64, 96
    tp = dict(learning_rate=2e-4, weight_decay=1e-4, lr_scheduler=dict(use=True, total_steps=50, pct_start=0.3))

    def batch(seed):
        rs = np.random.RandomState(seed)
        return {DataLoading.EV_REPR: cu(synthetic.voxel_grid(B, 9, H, W, seed=seed)),
                DataLoading.FLOW: cu(rs.standard_normal((B, 2, H, W)).astype(np.float32) * 4),
                DataLoading.FLOW_VALID: cu(rs.rand(B, H, W) > 0.2), DataLoading.DATASET_TYPE: [DataSetType.DSEC] * B}

    batches = [batch(5), batch(6), batch(5), batch(7)]

    def make(capturable):
        m = bflow_amd.RAFTSpline(cfg)
        m.load_state_dict(deterministic_state_dict(m, seed=0))
        m.to(DEV).train()
        opt, sch = training.configure_optimizers(m, tp, capturable=capturable)
        return m, opt, sch, training.TrainStep(m, num_iter_train=3)

    m1, opt1, sch1, step1 = make(False)
    eager = []
    for b in batches:
        opt1.zero_grad(set_to_none=True)
        out = step1(b)
        out["loss"].backward()
        opt1.step()
        sch1.step()
        eager.append(float(out["loss"].detach()))
    m2, opt2, sch2, step2 = make(True)
    vox_inf = batches[0][DataLoading.EV_REPR]
    with torch.no_grad():                                                # warm the INFERENCE engine's filter caches on the initial weights
        m2.eval()
        before = m2(voxel_grid=vox_inf, iters=3, test_mode=True)[1].get_params().clone()
        m2.train()
    gstep = training.GraphedTrainStep(step2, opt2, sch2)                 # its warm-up and capture are undone: the run starts from the same weights
    graphed = [float(gstep(b)["loss"].detach()) for b in batches]
    # replays rewrite the weights behind autograd's back: the inference path (packed filters keyed on parameter versions) must see them
    with torch.no_grad():
        m2.eval()
        after = m2(voxel_grid=vox_inf, iters=3, test_mode=True)[1].get_params().clone()
        m3 = bflow_amd.RAFTSpline(cfg)
        m3.load_state_dict(m2.state_dict())
        m3.to(DEV).eval()
        fresh = m3(voxel_grid=vox_inf, iters=3, test_mode=True)[1].get_params()
        m2.train()
    assert not torch.equal(before, after)
    assert torch.equal(after, fresh), float((after - fresh).abs().max())
    print("losses eager", eager, "graphed", graphed)
    assert eager[0] != eager[2]                                          # the weights moved between the two visits of batch 5
    for a, b in zip(eager, graphed):
        assert abs(a - b) < 2e-3 * abs(a), (eager, graphed)              # fp32 atomics order + three steps of drift; a frozen pack or lr is far off
    worst = 0.0
    for (n, p), q in zip(m1.named_parameters(), m2.parameters()):
        worst = max(worst, float(((p - q).abs().max() / (p.abs().max() + 1e-12)).detach()))
    print("worst relative parameter difference after 4 steps", worst)
    assert worst < 5e-2
    torch.cuda.synchronize()
    del gstep


def test_graphed_train_step_keeps_graphs_of_alternating_signatures():
    """A smaller last batch per epoch / alternating shapes: GraphedTrainStep keeps the captured graph of the most recently used signatures
    (`max_graphs=3`; the default keeps one) instead of re-capturing at every change, and the alternating replays still train like eager steps."""
    from bflow_amd import configs, synthetic
    from bflow_amd.weights import deterministic_state_dict
    cfg = configs.model_config("E_LU4_BD2")
    H, W = 64, 96
    tp = dict(learning_rate=2e-4, weight_decay=1e-4, lr_scheduler=dict(use=False))

    def batch(B, seed):
        rs = np.random.RandomState(seed)
        return {DataLoading.EV_REPR: cu(synthetic.voxel_grid(B, 9, H, W, seed=seed)),
                DataLoading.FLOW: cu(rs.standard_normal((B, 2, H, W)).astype(np.float32) * 4),
                DataLoading.FLOW_VALID: cu(rs.rand(B, H, W) > 0.2), DataLoading.DATASET_TYPE: [DataSetType.DSEC] * B}

    batches = [batch(2, 1), batch(1, 2), batch(2, 3), batch(1, 4), batch(2, 5)]

    def make(capturable):
        m = bflow_amd.RAFTSpline(cfg)
        m.load_state_dict(deterministic_state_dict(m, seed=0))
        m.to(DEV).train()
        opt, sch = training.configure_optimizers(m, tp, capturable=capturable)
        return m, opt, training.TrainStep(m, num_iter_train=2)

    m1, opt1, step1 = make(False)
    eager = []
    for b in batches:
        opt1.zero_grad(set_to_none=True)
        out = step1(b)
        out["loss"].backward()
        opt1.step()
        eager.append(float(out["loss"].detach()))
    m2, opt2, step2 = make(True)
    gstep = training.GraphedTrainStep(step2, opt2, max_graphs=3)       # (default 1: every graph owns a full set of activations)
    captures = []
    orig = gstep._capture
    gstep._capture = lambda b: (captures.append(1), orig(b))[1]
    graphed = [float(gstep(b)["loss"].detach()) for b in batches]
    print("losses eager", eager, "graphed", graphed, "captures", len(captures))
    assert len(captures) == 2                                            # one per signature, not one per change
    for a, b in zip(eager, graphed):
        assert abs(a - b) < 3e-3 * abs(a), (eager, graphed)
    torch.cuda.synchronize()
    del gstep


def test_graphed_train_step_multiflow_targets():
    """GraphedTrainStep on a MultiFlow batch (lists of GT tensors and timestamps, multi-target sequence loss, degree-10 curves): the
    timestamps are read on the host OUTSIDE the capture and frozen into the graph (part of its signature); losses equal the eager step's."""
    from bflow_amd import configs
    from bflow_amd.weights import deterministic_state_dict
    cfg = configs.model_config("E_LU5_BD10")
    B, H, W = 1, 64, 64
    vox, _ = TC.inputs(cfg, B, H, W)
    tp = dict(learning_rate=1e-4, weight_decay=1e-4, lr_scheduler=dict(use=False))

    def batch(seed):
        gts, _, times = TC.train_targets(B, H, W, "multiflow", seed=seed)
        return {DataLoading.EV_REPR: vox.to(DEV), DataLoading.FLOW: [cu(x) for x in gts],
                DataLoading.FLOW_TIMESTAMPS: [torch.full((B,), t, device=DEV) for t in times], DataLoading.DATASET_TYPE: [DataSetType.MULTIFLOW2D] * B}

    def make(capturable):
        m = bflow_amd.RAFTSpline(cfg)
        m.load_state_dict(deterministic_state_dict(m, seed=0))
        m.to(DEV).train()
        opt, _ = training.configure_optimizers(m, tp, capturable=capturable)
        return m, opt, training.TrainStep(m, num_iter_train=2)

    batches = [batch(11), batch(12), batch(11)]
    m1, opt1, step1 = make(False)
    eager = []
    for b in batches:
        opt1.zero_grad(set_to_none=True)
        out = step1(b)
        out["loss"].backward()
        opt1.step()
        eager.append(float(out["loss"].detach()))
    m2, opt2, step2 = make(True)
    gstep = training.GraphedTrainStep(step2, opt2)
    graphed = [float(gstep(b)["loss"].detach()) for b in batches]
    print("multiflow losses eager", eager, "graphed", graphed)
    for a, b in zip(eager, graphed):
        assert abs(a - b) < 2e-3 * abs(a), (eager, graphed)
    torch.cuda.synchronize()
    del gstep


def test_adjoint_identities_at_dsec_size():
    """Size-independent property at BASELINE C2 size (60x80 grid, 4 targets, 7 pyramid planes, 368.6 MB volume): look-up, pooling and
    up-sampling are LINEAR in the volume / the Bezier parameters, so <L x, y> == <x, L^T y> must hold for the adjoint kernels
    (fp32 dot products accumulated in fp64; 1e-5 relative)."""
    B, D, h, w, levels = 1, 256, 60, 80, [1, 1, 1, 4]
    T = len(levels)
    gen = torch.Generator(device="cpu").manual_seed(17)
    f1 = torch.randn(B, D, h, w, generator=gen).to(DEV)
    f2 = torch.randn(T, B, D, h, w, generator=gen).to(DEV)
    blk = CorrBlockParallelMultiTarget(corr_computation_events=CorrComputation(f1, f2, num_levels_per_target=levels))
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    coords = (torch.stack([xs, ys], 0).float()[None, None] + torch.randn(T, B, 2, h, w, generator=gen) * 6.0 + 0.31).to(DEV)
    out = blk(coords)                                              # L applied to the actual pyramid
    y = torch.randn(out.shape, generator=gen).to(DEV)
    grads = [torch.zeros_like(t) for t, _ in blk._pyramid]
    table = hip.make_grad_table([grads[lvl][k] for lvl, (_, idx) in enumerate(blk._pyramid) for k in range(len(idx))])
    hip.corr_lookup_bwd(blk._table, table, coords, y)
    lhs = float((out.double() * y.double()).sum())
    rhs = sum(float((t.double() * g.double()).sum()) for (t, _), g in zip(blk._pyramid, grads))
    assert abs(lhs - rhs) < 1e-5 * max(abs(lhs), float(out.double().norm() * y.double().norm()) * 1e-3), (lhs, rhs)
    # pooling: <pool(a), b> == <a, pool^T(b)>
    a = blk._pyramid[0][0][3]                                      # (B*N, 60, 80) slab of the 4-level target
    pooled = blk._pyramid[1][0][0]
    bb = torch.randn(pooled.shape, generator=gen).to(DEV)
    ga = torch.zeros_like(a)
    hip.corr_pool2x2_bwd(bb, ga)
    l2, r2 = float((pooled.double() * bb.double()).sum()), float((a.double() * ga.double()).sum())
    assert abs(l2 - r2) < 1e-6 * float(pooled.double().norm() * bb.double().norm())
    # convex up-sampling is linear in the data for a fixed mask
    data = torch.randn(B, 4, h, w, generator=gen).to(DEV)
    mask = (torch.randn(B, 576, h, w, generator=gen) * 2).to(DEV)
    up = hip.cvx_upsample(data, mask)
    gu = torch.randn(up.shape, generator=gen).to(DEV)
    gd, _ = hip.cvx_upsample_bwd(gu, data, mask)
    l3, r3 = float((up.double() * gu.double()).sum()), float((data.double() * gd.double()).sum())
    assert abs(l3 - r3) < 1e-5 * float(up.double().norm() * gu.double().norm())


@pytest.mark.parametrize("what", ["group_norm", "feature_dim_96"])
def test_training_forward_refuses_what_only_the_inference_engine_runs(what):
    """GroupNorm encoders and zero-padded feature dims exist on the inference engine only (round 5): a training-mode forward on the GPU says so
    BEFORE its first launch (training.forward_train) instead of failing inside a normalisation layer -- and the same model runs in eval()."""
    import copy
    from bflow_amd import configs, synthetic
    cfg = copy.deepcopy(configs.model_config("E_LU4_BD2"))
    if what == "group_norm":
        cfg["feature"]["norm"] = "group"
    else:
        cfg["feature"]["dim"] = 96
    m = bflow_amd.RAFTSpline(cfg)
    m.load_state_dict(O.make_state_dict(cfg, seed=2, gain=0.35))
    m.to(DEV)
    vox = torch.from_numpy(synthetic.voxel_grid(1, 9, 64, 96, seed=3)).to(DEV)
    m.train()
    with pytest.raises(hip.BflowHipError, match="training forward"):
        m(voxel_grid=vox, iters=2)
    m.eval()
    with torch.inference_mode():
        low, up = m(voxel_grid=vox, iters=2, test_mode=True)
    assert bool(torch.isfinite(up.get_params()).all())
