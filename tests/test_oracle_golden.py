"""The CPU oracle (oracle/raft_spline_oracle.py) against the committed golden vectors, which were produced by
running the reference itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from bflow_amd import synthetic
from oracle import raft_spline_oracle as O

E2E = ["e2e_E_LU4_BD2", "e2e_E_I_LU4_BD2", "e2e_E_LU5_BD10", "e2e_E_I_LU5_BD10"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def e2e_inputs(g):
    cfg = O.model_config(str(g["config"]))
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    C = cfg["num_bins"]["context"] + cfg["num_bins"]["correlation"] - 1
    vox = torch.from_numpy(synthetic.voxel_grid(B, C, H, W, seed=1234))
    imgs = None
    if cfg["use_boundary_images"]:
        a, b = synthetic.image_pair(B, H, W, seed=4321)
        imgs = [torch.from_numpy(a), torch.from_numpy(b)]
    return cfg, vox, imgs


@pytest.mark.parametrize("name", E2E)
def test_e2e_forward_matches_reference_golden(golden_dir, name):
    g = _load(golden_dir, name)
    cfg, vox, imgs = e2e_inputs(g)
    sd = O.make_state_dict(cfg, seed=0)
    with torch.inference_mode():
        low, up = O.forward(sd, cfg, vox, imgs, iters=int(g["iters"]), test_mode=True)
        f1 = O.bezier_flow(up, 1.0)
        f05 = O.bezier_flow(up, 0.5)
    # same torch build produced the fixtures -> expect (near) bit equality; tolerance covers thread-count changes
    assert np.abs(low.numpy() - g["bezier_low"]).max() < 2e-4
    assert np.abs(f1.numpy() - g["flow_t1"]).max() < 2e-3
    assert float(O.epe_masked(f1, torch.from_numpy(g["flow_t1"]))) < 1e-4
    assert float(O.epe_masked(f05, torch.from_numpy(g["flow_t05"]))) < 1e-4
    assert np.abs(up[:, :, ::4, ::4].numpy() - g["bezier_up_sub"]).max() < 2e-3


def test_train_mode_returns_one_upsampled_prediction_per_iteration(golden_dir):
    g = _load(golden_dir, "e2e_E_LU4_BD2")
    cfg, vox, _ = e2e_inputs(g)
    sd = O.make_state_dict(cfg, seed=0)
    with torch.inference_mode():
        ups = O.forward(sd, cfg, vox, None, iters=3, test_mode=False)
        low, up = O.forward(sd, cfg, vox, None, iters=3, test_mode=True)
    assert len(ups) == 3 and torch.equal(ups[-1], up)


def test_corr_1toN(golden_dir):
    g = _load(golden_dir, "corr_1toN")
    vol = O.corr_volume(torch.from_numpy(g["f1"]), torch.from_numpy(g["f2"]))
    pyr = O.corr_pyramid(vol, g["levels"].tolist())
    assert [t for _, t in pyr] == [[0, 1, 2], [1, 2], [2]]
    for L, (c, _) in enumerate(pyr):
        np.testing.assert_allclose(c.numpy(), g[f"pyr{L}"], rtol=1e-5, atol=1e-5)
    out = O.corr_lookup(pyr, torch.from_numpy(g["coords"]))
    assert out.shape == g["lookup"].shape == (2, 6 * 81, 10, 12)
    np.testing.assert_allclose(out.numpy(), g["lookup"], rtol=1e-5, atol=1e-5)


def test_corr_MtoN_odd_pyramid(golden_dir):
    g = _load(golden_dir, "corr_MtoN")
    f1e, f2e = torch.from_numpy(g["f1_ev"]), torch.from_numpy(g["f2_ev"])
    f1i, f2i = torch.from_numpy(g["f1_img"]), torch.from_numpy(g["f2_img"])
    f1 = torch.cat([f1e.unsqueeze(0).expand(4, -1, -1, -1, -1), f1i.unsqueeze(0)], 0)
    f2 = torch.cat([f2e, f2i.unsqueeze(0)], 0)
    vol = O.corr_volume(f1, f2)
    pyr = O.corr_pyramid(vol, g["levels"].tolist())
    assert [t for _, t in pyr] == [[0, 1, 2, 3, 4], [3, 4], [3, 4], [3, 4]]
    assert [tuple(c.shape[-2:]) for c, _ in pyr] == [(18, 22), (9, 11), (4, 5), (2, 2)]
    np.testing.assert_allclose(pyr[0][0].numpy()[:, ::7], g["pyr0_rows7"], rtol=1e-5, atol=1e-5)
    for L in (1, 2, 3):
        np.testing.assert_allclose(pyr[L][0].numpy(), g[f"pyr{L}"], rtol=1e-5, atol=1e-5)
    out = O.corr_lookup(pyr, list(torch.from_numpy(g["coords"])))
    assert out.shape == (1, 11 * 81, 18, 22)
    np.testing.assert_allclose(out.numpy(), g["lookup"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("deg", [2, 10])
def test_bezier(golden_dir, deg):
    g = _load(golden_dir, "bezier")
    p = torch.from_numpy(g[f"params_d{deg}"])
    ts = g[f"times_d{deg}"].tolist()
    np.testing.assert_allclose(O.bezier_flow(p, ts).numpy(), g[f"flow_list_d{deg}"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(O.bezier_flow(p, 0.0).numpy(), g[f"flow_0_d{deg}"])
    np.testing.assert_array_equal(O.bezier_flow(p, 1.0).numpy(), g[f"flow_1_d{deg}"])
    np.testing.assert_allclose(O.bezier_flow(p, 0.3).numpy(), g[f"flow_03_d{deg}"], rtol=1e-6, atol=1e-6)
    # the Bernstein weights (incl. the implicit P0 = 0 term) sum to one
    c = O.bezier_coeffs([0.37], deg)
    assert abs(c.sum() + (1 - 0.37) ** deg - 1) < 1e-12


def test_cvx_upsample(golden_dir):
    g = _load(golden_dir, "cvx_upsample")
    out = O.cvx_upsample(torch.from_numpy(g["data"]), torch.from_numpy(g["mask"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag", ["f", "i"])
def test_voxel_grid(golden_dir, tag):
    g = _load(golden_dir, "voxel")
    ts, te, t0c, t1c = g["window"].tolist()
    assert O.extended_time_window(t0c, t1c, 5) == (ts, te)
    grid = O.voxel_grid_convert(torch.from_numpy(g[f"x_{tag}"]), torch.from_numpy(g[f"y_{tag}"]),
                                torch.from_numpy(g[f"pol_{tag}"]), torch.from_numpy(g[f"t_{tag}"]), 5, 24, 32, t0c, t1c)
    np.testing.assert_array_equal(grid.numpy(), g[f"grid_{tag}"])
    np.testing.assert_allclose(O.norm_voxel_grid(grid.clone()).numpy(), g[f"norm_{tag}"], rtol=1e-6, atol=1e-6)


def test_norm_voxel_grid_edge_cases(golden_dir):
    g = _load(golden_dir, "voxel")
    np.testing.assert_array_equal(O.norm_voxel_grid(torch.zeros(3, 4, 5)).numpy(), g["norm_allzero"])
    np.testing.assert_array_equal(O.norm_voxel_grid(torch.from_numpy(g["norm_std0_in"].copy())).numpy(), g["norm_std0"])


def test_epe(golden_dir):
    g = _load(golden_dir, "epe")
    a, b, m = torch.from_numpy(g["a"]), torch.from_numpy(g["b"]), torch.from_numpy(g["mask"])
    np.testing.assert_allclose(O.epe_masked(a, b).numpy(), g["epe"], rtol=1e-6)
    np.testing.assert_allclose(O.epe_masked(a, b, m).numpy(), g["epe_masked"], rtol=1e-6)
    assert O.epe_masked(a, b, torch.zeros_like(m)) is None


def test_validation_metrics(golden_dir):
    """SURVEY f-3: AE / NPE / EPE_MULTI / AE_MULTI / linear-assumption predictions vs the reference's outputs."""
    g = _load(golden_dir, "metrics")
    M = int(g["M"])
    T = lambda k: torch.from_numpy(g[k])
    preds, gts, masks = [T(f"pred{i}") for i in range(M)], [T(f"gt{i}") for i in range(M)], [T(f"mask{i}") for i in range(M)]
    for i in range(M):
        np.testing.assert_array_equal(O.ae_masked(preds[i], gts[i]).numpy(), g[f"ae{i}"])
        np.testing.assert_array_equal(O.ae_masked(preds[i], gts[i], None, degrees=False).numpy(), g[f"ae_rad{i}"])
        for n in (1, 2, 3):
            np.testing.assert_array_equal(O.n_pixel_error_masked(preds[i], gts[i], None, n).numpy(), g[f"npe{n}_{i}"])
        if bool(masks[i].any()):
            np.testing.assert_array_equal(O.ae_masked(preds[i], gts[i], masks[i]).numpy(), g[f"ae_m{i}"])
            for n in (1, 2, 3):
                np.testing.assert_array_equal(O.n_pixel_error_masked(preds[i], gts[i], masks[i], n).numpy(), g[f"npe{n}_m{i}"])
    np.testing.assert_array_equal(O.epe_masked_multi(preds, gts).numpy(), g["epe_multi"])
    np.testing.assert_array_equal(O.epe_masked_multi(preds, gts, masks).numpy(), g["epe_multi_m"])
    assert O.epe_masked_multi(preds[3:], gts[3:], masks[3:]) is None          # only an empty mask
    np.testing.assert_array_equal(O.ae_masked_multi(preds, gts).numpy(), g["ae_multi"])
    np.testing.assert_array_equal(O.ae_masked_multi(preds[:3], gts[:3], masks[:3]).numpy(), g["ae_multi_m3"])
    np.testing.assert_array_equal(O.compute_traj_len(gts).numpy(), g["traj_len"])
    tl = O.compute_traj_len(gts[:3])
    ok = (tl >= 4.0) & (tl <= 30.0)
    e = O.epe_masked_multi(preds[:3], gts[:3], [m & ok for m in masks[:3]])
    np.testing.assert_array_equal(e.double().float().numpy(), g["epe_multi_traj_4_30"])
    lin = O.predictions_from_lin_assumption(preds[3], list(g["lin_ts"]))
    np.testing.assert_array_equal(O.epe_masked_multi(lin, gts).numpy(), g["epe_multi_lin"])


def test_input_padder(golden_dir):
    g = _load(golden_dir, "padder")
    for tag in "abcd":
        x = torch.from_numpy(g[f"x_{tag}"])
        pad = O.input_pad_amounts(x.shape[-2], x.shape[-1], 8, bool(g[f"no_top_{tag}"]))
        assert pad == list(g[f"pad_{tag}"])
        y = O.input_pad(x, pad)
        np.testing.assert_array_equal(y.numpy(), g[f"y_{tag}"])
        assert y.shape[-2] % 8 == 0 and y.shape[-1] % 8 == 0
        assert torch.equal(O.input_unpad(y, pad), x)


def test_dsec_twostep_assembly(golden_dir):
    """SURVEY f-1: raw events -> rectify -> two voxel grids -> merge -> normalise, vs the reference's own __getitem__ outputs."""
    g = _load(golden_dir, "dsec_twostep")
    ev = {k: g[k] for k in ("x", "y", "p", "t")}
    rect, ts, bins = g["rectify_map"], g["forward_flow_timestamps"], int(g["num_bins"])
    H, W = rect.shape[:2]
    for tag, norm, merge in (("nm", True, True), ("m", False, True), ("n", True, False)):
        for idx in (0, 1):
            out = O.dsec_twostep_sample(ev, rect, ts, idx, bins, H, W, normalize=norm, merge=merge)
            np.testing.assert_array_equal(out.numpy(), g[f"sample_{tag}_{idx}"])
    assert O.twostep_windows(ts, 0) == [(2_100_000, 2_200_000), (2_000_000, 2_100_000)]     # previous interval extrapolated
    assert O.twostep_windows(ts, 1) == [(2_200_000, 2_300_000), (2_100_000, 2_200_000)]
    assert list(O.event_window_indices(ev["t"], 2_150_000, 2_250_000)) == list(g["offsets_2150000_2250000"])
    assert list(O.event_window_indices(ev["t"], 2_500_000, 2_600_000)) == list(g["offsets_past_end"])


def test_param_inventory_counts():
    # 5,344,832 parameters for the events-only DSEC model
    cfg = O.model_config("E_LU4_BD2")
    shapes = O.param_shapes(cfg)
    n = sum(int(np.prod(s)) for k, s in shapes.items()
            if not k.endswith(("running_mean", "running_var", "num_batches_tracked")) and ".downsample.1." not in k)
    assert n == 5344832
    assert O.num_corr_planes(cfg) == 567 and O.num_corr_planes(O.model_config("E_I_LU5_BD10")) == 972


# ----------------------------------------------------------------------------------------------- training path
def test_losses_match_reference_golden(golden_dir):
    g = _load(golden_dir, "losses")
    T = torch.from_numpy
    srcs = [T(g[f"src{i}"]).requires_grad_(True) for i in range(4)]
    tgt, valid = T(g["tgt"]), T(g["valid"])
    assert abs(float(O.l1_loss_channel_masked(srcs[0], tgt, valid)) - float(g["l1_masked"])) < 1e-6
    assert abs(float(O.l1_loss_channel_masked(srcs[0], tgt)) - float(g["l1_unmasked"])) < 1e-6
    for tag, m, gamma in (("seq_masked", valid, 0.8), ("seq_unmasked", None, 0.8), ("seq_masked_g085", valid, 0.85)):
        for s in srcs:
            s.grad = None
        loss = O.l1_seq_loss_channel_masked(srcs, tgt, m, gamma=gamma)
        loss.backward()
        assert abs(float(loss) - float(g[tag])) < 1e-5
        for i, s in enumerate(srcs):
            assert np.abs(s.grad.numpy() - g[f"{tag}_grad{i}"]).max() < 1e-8
    tgts = [T(g[f"mtgt{m}"]) for m in range(3)]
    valids = [T(g[f"mvalid{m}"]) for m in range(3)]
    multi = [[T(g[f"msrc{it}_{m}"]) for m in range(3)] for it in range(2)]
    assert abs(float(O.l1_multi_seq_loss_channel_masked(multi, tgts, valids)) - float(g["multi_masked"])) < 1e-5
    assert abs(float(O.l1_multi_seq_loss_channel_masked(multi, tgts)) - float(g["multi_unmasked"])) < 1e-5


@pytest.mark.parametrize("name", ["train_E_LU4_BD2", "train_E_I_LU4_BD2", "train_E_LU5_BD10", "train_E_LU4_BD2_detach_init"])
def test_training_step_gradients_match_reference_golden(golden_dir, name):
    """Training-mode forward (BatchNorm on batch statistics) + sequence loss + autograd of the oracle == the reference's."""
    import train_common as TC
    g = _load(golden_dir, name)
    cfg, init = TC.case_setup(name, O.model_config(str(g["config"])), int(g["B"]), int(g["H"]), int(g["W"]))
    loss, grads, bufs, last = TC.oracle_train_step(cfg, int(g["B"]), int(g["H"]), int(g["W"]), int(g["iters"]), str(g["kind"]),
                                                   flow_init=init)
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert np.abs(last[:, :, ::4, ::4].numpy() - g["last_params_sub"]).max() < 1e-4
    TC.check_grads(grads, g, rel=2e-4)
    for k, v in bufs.items():
        if f"buf/{k}" in g:
            assert np.abs(v.numpy() - g[f"buf/{k}"]).max() < 1e-5
