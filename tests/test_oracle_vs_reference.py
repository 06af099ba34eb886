"""Live pin of the oracle: run the reference (when /root/reference is mounted, i.e. in the build container)
and the oracle on the same inputs/weights.  Skipped on the GPU box, where the reference does not exist."""
import contextlib
import io

import numpy as np
import pytest
import torch

import refshim
from bflow_amd import synthetic
from oracle import raft_spline_oracle as O

pytestmark = pytest.mark.skipif(not refshim.reference_available(), reason="reference not mounted")


@pytest.fixture(scope="module")
def ref():
    return refshim.import_reference()


@pytest.mark.parametrize("cname,B,H,W,iters", [("E_LU4_BD2", 2, 128, 160, 3), ("E_I_LU5_BD10", 1, 128, 128, 2)])
def test_forward_and_state_dict(ref, cname, B, H, W, iters):
    cfg = O.model_config(cname)
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.RAFTSpline(cfg).eval()
    ref_sd = model.state_dict()
    shapes = O.param_shapes(cfg)
    assert list(ref_sd.keys()) == list(shapes.keys())
    assert all(tuple(ref_sd[k].shape) == shapes[k] for k in shapes)
    sd = O.make_state_dict(cfg, seed=3)
    model.load_state_dict(sd)
    C = cfg["num_bins"]["context"] + cfg["num_bins"]["correlation"] - 1
    vox = torch.from_numpy(synthetic.voxel_grid(B, C, H, W, seed=5))
    imgs = None
    if cfg["use_boundary_images"]:
        imgs = [torch.from_numpy(a) for a in synthetic.image_pair(B, H, W, seed=6)]
    with torch.inference_mode():
        lo, up = model(voxel_grid=vox, images=imgs, iters=iters, test_mode=True)
        olo, oup = O.forward(sd, cfg, vox, imgs, iters=iters, test_mode=True)
        ups = model(voxel_grid=vox, images=imgs, iters=iters, test_mode=False)
        oups = O.forward(sd, cfg, vox, imgs, iters=iters, test_mode=False)
    assert torch.equal(lo.get_params(), olo) and torch.equal(up.get_params(), oup)
    assert len(ups) == len(oups) == iters
    for a, b in zip(ups, oups):
        assert torch.equal(a.get_params(), b)
    for t in (0.0, 0.4, 1.0, [0.1, 0.9]):
        assert torch.equal(up.get_flow_from_reference(t), O.bezier_flow(oup, t))


@pytest.mark.parametrize("fnorm,cnorm", [("group", "none"), ("none", "group")])
def test_other_encoder_norms_match_reference(ref, fnorm, cnorm):
    """The rest of the reference's encoder constructor surface (extractor.py:13-37,63-70): norm_fn 'group' and 'none' -- state-dict keys /
    shapes and the forward, bit for bit."""
    import copy
    cfg = copy.deepcopy(O.model_config("E_LU4_BD2"))
    cfg["feature"]["norm"], cfg["context"]["norm"] = fnorm, cnorm
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.RAFTSpline(cfg).eval()
    ref_sd = model.state_dict()
    shapes = O.param_shapes(cfg)
    assert list(ref_sd.keys()) == list(shapes.keys())
    assert all(tuple(ref_sd[k].shape) == shapes[k] for k in shapes)
    sd = O.make_state_dict(cfg, seed=4, gain=0.35)     # (an encoder without normalisation overflows on the gains tuned for normalised ones)
    model.load_state_dict(sd)
    vox = torch.from_numpy(synthetic.voxel_grid(1, 9, 128, 160, seed=9))
    with torch.inference_mode():
        lo, up = model(voxel_grid=vox, iters=2, test_mode=True)
        olo, oup = O.forward(sd, cfg, vox, None, iters=2, test_mode=True)
    assert bool(torch.isfinite(oup).all()) and float(oup.abs().max()) > 1e-3
    assert torch.equal(lo.get_params(), olo) and torch.equal(up.get_params(), oup)


def test_flow_init_warm_start(ref):
    cfg = O.model_config("E_LU4_BD2")
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.RAFTSpline(cfg).eval()
    sd = O.make_state_dict(cfg, seed=1)
    model.load_state_dict(sd)
    vox = torch.from_numpy(synthetic.voxel_grid(1, 9, 128, 160, seed=8))
    init = torch.from_numpy(np.random.RandomState(2).standard_normal((1, 4, 16, 20)).astype(np.float32))
    with torch.inference_mode():
        lo, _ = model(voxel_grid=vox, iters=2, flow_init=ref.BezierCurves(init), test_mode=True)
        olo, _ = O.forward(sd, cfg, vox, None, iters=2, flow_init=init, test_mode=True)
    assert torch.equal(lo.get_params(), olo)


def test_voxel_and_norm(ref):
    for int_xy in (False, True):
        x, y, pol, t = synthetic.events(20000, 48, 64, 0, 120000, seed=3, int_xy=int_xy)
        a = ref.VoxelGrid(5, 48, 64).convert(*(torch.from_numpy(v) for v in (x, y, pol, t)), 10000, 110000)
        b = O.voxel_grid_convert(*(torch.from_numpy(v) for v in (x, y, pol, t)), 5, 48, 64, 10000, 110000)
        assert torch.equal(a, b)
        assert torch.equal(ref.norm_voxel_grid(a.clone()), O.norm_voxel_grid(b.clone()))


def test_validation_metrics_and_padder(ref):
    """SURVEY f-3 restatements against the live reference on fresh random data (bit-exact: same torch calls)."""
    g = torch.Generator().manual_seed(5)
    preds = [torch.randn(2, 2, 19, 23, generator=g) * 4 for _ in range(3)]
    gts = [p + torch.randn(p.shape, generator=g) * s for p, s in zip(preds, (0.1, 1.0, 5.0))]
    masks = [torch.rand(2, 19, 23, generator=g) < 0.6 for _ in range(3)]
    for p, t, m in zip(preds, gts, masks):
        for vm in (None, m):
            assert torch.equal(ref.ae_masked(p, t, vm), O.ae_masked(p, t, vm))
            assert torch.equal(ref.ae_masked(p, t, vm, degrees=False), O.ae_masked(p, t, vm, degrees=False))
            for n in (1, 2, 3):
                assert torch.equal(ref.n_pixel_error_masked(p, t, vm, n), O.n_pixel_error_masked(p, t, vm, n))
    assert torch.equal(ref.epe_masked_multi(preds, gts, masks), O.epe_masked_multi(preds, gts, masks))
    assert torch.equal(ref.ae_masked_multi(preds, gts, masks), O.ae_masked_multi(preds, gts, masks))
    assert torch.equal(ref.EPE_MULTI.compute_traj_len(gts), O.compute_traj_len(gts))
    for a, b in zip(ref.predictions_from_lin_assumption(preds[0], [0.2, 1.0]), O.predictions_from_lin_assumption(preds[0], [0.2, 1.0])):
        assert torch.equal(a, b)
    for (hh, ww), no_top in (((21, 30), False), ((37, 41), True), ((16, 24), False)):
        x = torch.randn(1, 2, hh, ww, generator=g)
        rp = ref.InputPadder(8, no_top)
        y = rp.pad(x)
        pad = O.input_pad_amounts(hh, ww, 8, no_top)
        assert pad == rp._pad and torch.equal(y, O.input_pad(x, pad)) and torch.equal(rp.unpad(y), O.input_unpad(y, pad))


def test_dsec_twostep_assembly_live():
    """SURVEY f-1 against the live reference (TwoStepSubSequence.__getitem__ driven on an in-memory stream, tests/refshim.py)."""
    import refshim
    if not refshim.reference_available():
        pytest.skip("reference not mounted")
    dns = refshim.import_reference_dsec()
    H, W, bins = 32, 48, 4
    rs = np.random.RandomState(11)
    n = 20000
    ev = dict(x=rs.randint(0, W, n).astype(np.uint16), y=rs.randint(0, H, n).astype(np.uint16), p=rs.randint(0, 2, n).astype(np.uint8),
              t=np.sort(rs.randint(4_900_000, 5_450_000, n)).astype(np.int64))
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    rect = np.stack([xx + rs.uniform(-2, 2, (H, W)), yy + rs.uniform(-2, 2, (H, W))], -1).astype(np.float32)
    ts = np.array([[5_050_000, 5_150_000], [5_150_000, 5_250_000], [5_250_000, 5_350_000]], dtype=np.int64)
    drv = refshim.ReferenceTwoStepDriver(dns, ev, rect, ts, bins, H, W)
    for idx in range(3):
        assert torch.equal(drv.sample(idx), O.dsec_twostep_sample(ev, rect, ts, idx, bins, H, W))
