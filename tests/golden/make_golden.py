"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE (uzh-rpg/bflow, mounted
read-only at /root/reference) in the build container.  Fixtures hold data only: inputs (or the seeds that
regenerate them) and the reference's outputs.  Re-run:  python tests/golden/make_golden.py

Weights come from oracle.raft_spline_oracle.make_state_dict (numpy RandomState, independent of torch's RNG);
inputs from bflow_amd.synthetic.  Both are loaded INTO the reference modules; every stored output is computed
by reference code only.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import refshim  # noqa: E402
from bflow_amd import synthetic  # noqa: E402
from oracle import raft_spline_oracle as O  # noqa: E402

# (config, B, H, W, iters) -- frames sized so that the 4-level pyramid exercises odd-size flooring
E2E_CASES = {
    "e2e_E_LU4_BD2": ("E_LU4_BD2", 1, 176, 208, 12),
    "e2e_E_I_LU4_BD2": ("E_I_LU4_BD2", 2, 176, 208, 6),
    "e2e_E_LU5_BD10": ("E_LU5_BD10", 1, 144, 176, 4),
    "e2e_E_I_LU5_BD10": ("E_I_LU5_BD10", 1, 144, 176, 3),
}


def e2e_inputs(cfg, B, H, W):
    C = cfg["num_bins"]["context"] + cfg["num_bins"]["correlation"] - 1
    vox = torch.from_numpy(synthetic.voxel_grid(B, C, H, W, seed=1234))
    imgs = None
    if cfg["use_boundary_images"]:
        a, b = synthetic.image_pair(B, H, W, seed=4321)
        imgs = [torch.from_numpy(a), torch.from_numpy(b)]
    return vox, imgs


def corr_case_inputs(seed, B, D, h, w, T):
    rs = np.random.RandomState(seed)
    f1 = rs.standard_normal((B, D, h, w)).astype(np.float32)
    f2 = rs.standard_normal((T, B, D, h, w)).astype(np.float32)
    # coords: grid + flow with large excursions (out of bounds on all four sides, negative, fractional)
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    base = np.stack([xs, ys], 0).astype(np.float32)
    flow = (rs.standard_normal((T, B, 2, h, w)) * 3.0).astype(np.float32)
    flow[:, :, :, 0, 0] = -50.0      # far outside
    flow[:, :, :, -1, -1] = 40.0
    flow[:, :, :, 1, 1] = 0.0        # exactly integer coordinates
    coords = base[None, None] + flow
    return f1, f2, coords.astype(np.float32)


# (config, B, H, W, iters, loss kind) -- training-mode forward + loss + backward of the REFERENCE
TRAIN_CASES = {
    "train_E_LU4_BD2": ("E_LU4_BD2", 2, 128, 160, 3, "dsec"),
    "train_E_I_LU4_BD2": ("E_I_LU4_BD2", 1, 128, 160, 2, "dsec"),
    "train_E_LU5_BD10": ("E_LU5_BD10", 1, 128, 128, 2, "multiflow"),
    # detach_bezier=True (raft.py:167-168) and a warm start through flow_init (raft.py:152-153)
    "train_E_LU4_BD2_detach_init": ("E_LU4_BD2", 1, 128, 160, 3, "dsec"),
}
GRAD_STRIDE = 97          # every parameter gradient is stored as (L2 norm, sum, every 97th element)


def train_targets(B, H, W, kind, seed=99):
    """Ground truth of the synthetic training sample: DSEC = one flow + valid mask, MultiFlow = 3 flows at times .4, .7, 1."""
    rs = np.random.RandomState(seed)
    if kind == "dsec":
        return [synthetic.gt_flow(B, H, W, seed=seed)], [rs.rand(B, H, W) < 0.8], [1.0]
    times = [0.4, 0.7, 1.0]
    return [synthetic.gt_flow(B, H, W, seed=seed + k) * t for k, t in enumerate(times)], None, times


def train_flow_init(B, C, h, w, seed=55):
    return (np.random.RandomState(seed).standard_normal((B, C, h, w)) * 1.5).astype(np.float32)


def training_goldens(ns, out_dir):
    T = torch.from_numpy
    # ---------------- losses (utils/losses.py) ----------------
    rs = np.random.RandomState(91)
    ls = {}
    srcs = [rs.standard_normal((2, 2, 12, 16)).astype(np.float32) for _ in range(4)]
    tgt = rs.standard_normal((2, 2, 12, 16)).astype(np.float32)
    srcs[1][0, :, 3, 4] = tgt[0, :, 3, 4]                      # exact zeros of the difference (sign(0) = 0 in the gradient)
    valid = rs.rand(2, 12, 16) < 0.7
    ls.update(tgt=tgt, valid=valid, **{f"src{i}": a for i, a in enumerate(srcs)})
    ts = [T(a).requires_grad_(True) for a in srcs]
    ls["l1_masked"] = ns.l1_loss_channel_masked(ts[0], T(tgt), T(valid)).detach().numpy()
    ls["l1_unmasked"] = ns.l1_loss_channel_masked(ts[0], T(tgt)).detach().numpy()
    for tag, m, gamma in (("seq_masked", T(valid), 0.8), ("seq_unmasked", None, 0.8), ("seq_masked_g085", T(valid), 0.85)):
        for t in ts:
            t.grad = None
        loss = ns.l1_seq_loss_channel_masked(ts, T(tgt), m, gamma=gamma)
        loss.backward()
        ls[tag] = loss.detach().numpy()
        for i, t in enumerate(ts):
            ls[f"{tag}_grad{i}"] = t.grad.numpy().copy()
    tgts = [rs.standard_normal((2, 2, 12, 16)).astype(np.float32) for _ in range(3)]
    valids = [rs.rand(2, 12, 16) < 0.6 for _ in range(3)]
    multi = [[rs.standard_normal((2, 2, 12, 16)).astype(np.float32) for _ in range(3)] for _ in range(2)]
    for m, a in enumerate(tgts):
        ls[f"mtgt{m}"], ls[f"mvalid{m}"] = a, valids[m]
    for it, row in enumerate(multi):
        for m, a in enumerate(row):
            ls[f"msrc{it}_{m}"] = a
    ls["multi_masked"] = ns.l1_multi_seq_loss_channel_masked([[T(a) for a in r] for r in multi], [T(a) for a in tgts], [T(v) for v in valids]).numpy()
    ls["multi_unmasked"] = ns.l1_multi_seq_loss_channel_masked([[T(a) for a in r] for r in multi], [T(a) for a in tgts]).numpy()
    np.savez_compressed(os.path.join(out_dir, "losses.npz"), **ls)

    # ---------------- training-mode forward + loss + backward of the reference model ----------------
    for fname, (cname, B, H, W, iters, kind) in TRAIN_CASES.items():
        cfg = O.model_config(cname)
        flow_init = None
        if fname.endswith("detach_init"):
            cfg["detach_bezier"] = True
            flow_init = ns.BezierCurves(torch.from_numpy(train_flow_init(B, 2 * cfg["bezier_degree"], H // 8, W // 8)))
        with contextlib.redirect_stdout(io.StringIO()):
            model = ns.RAFTSpline(cfg).train()
        model.load_state_dict(O.make_state_dict(cfg, seed=0))
        vox, imgs = e2e_inputs(cfg, B, H, W)
        gts, valids, times = train_targets(B, H, W, kind)
        preds = model(voxel_grid=vox, images=imgs, iters=iters, flow_init=flow_init, test_mode=False)
        if kind == "dsec":
            flows = [p.get_flow_from_reference(1.0) for p in preds]
            loss = ns.l1_seq_loss_channel_masked(flows, T(gts[0]), T(valids[0]))
        else:
            flows = [[p.get_flow_from_reference(t) for t in times] for p in preds]
            loss = ns.l1_multi_seq_loss_channel_masked(flows, [T(g) for g in gts])
        loss.backward()
        out = dict(config=cname, B=B, H=H, W=W, iters=iters, kind=kind, loss=loss.detach().numpy(),
                   last_params_sub=preds[-1].get_params().detach()[:, :, ::4, ::4].numpy())
        for name, prm in model.named_parameters():
            g = prm.grad if prm.grad is not None else torch.zeros_like(prm)
            out[f"gnorm/{name}"] = g.norm().double().numpy()
            out[f"gsum/{name}"] = g.double().sum().numpy()
            out[f"gsub/{name}"] = g.flatten()[::GRAD_STRIDE].numpy().copy()
        for name, buf in model.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                out[f"buf/{name}"] = buf.detach().numpy().copy()
        np.savez_compressed(os.path.join(out_dir, fname + ".npz"), **out)
        print(fname, "loss", float(loss))


def main():
    ns = refshim.import_reference()
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "training":
        training_goldens(ns, HERE)
        return
    torch.manual_seed(0)
    out_dir = HERE

    # ---------------- end-to-end ----------------
    for fname, (cname, B, H, W, iters) in E2E_CASES.items():
        cfg = O.model_config(cname)
        with contextlib.redirect_stdout(io.StringIO()):
            model = ns.RAFTSpline(cfg).eval()
        model.load_state_dict(O.make_state_dict(cfg, seed=0))
        vox, imgs = e2e_inputs(cfg, B, H, W)
        with torch.inference_mode():
            low, up = model(voxel_grid=vox, images=imgs, iters=iters, test_mode=True)
            flow1 = up.get_flow_from_reference(1.0)
            flow_half = up.get_flow_from_reference(0.5)
        np.savez_compressed(os.path.join(out_dir, fname + ".npz"),
                            config=cname, B=B, H=H, W=W, iters=iters,
                            bezier_low=low.get_params().numpy(),
                            flow_t1=flow1.numpy(), flow_t05=flow_half.numpy(),
                            bezier_up_sub=up.get_params()[:, :, ::4, ::4].numpy())
        print(fname, "flow |mean|", float(flow1.abs().mean()))

    # ---------------- correlation volume / pyramid / lookup ----------------
    with torch.inference_mode():
        # 1-to-N, levels [1,2,3]
        f1, f2, coords = corr_case_inputs(11, 2, 32, 10, 12, 3)
        cc = ns.CorrComputation(torch.from_numpy(f1), torch.from_numpy(f2), num_levels_per_target=[1, 2, 3])
        blk = ns.CorrBlockParallelMultiTarget(corr_computation_events=cc)
        look = blk(torch.from_numpy(coords))
        pyr = {f"pyr{L}": d.corr.numpy() for L, d in enumerate(blk._corr_pyramid)}
        np.savez_compressed(os.path.join(out_dir, "corr_1toN.npz"), f1=f1, f2=f2, coords=coords,
                            levels=np.array([1, 2, 3]), lookup=look.numpy(), **pyr)
        # M-to-N: events [1,1,1,4] + frames 4 ; odd sizes 18x22 -> 9x11 -> 4x5 -> 2x2
        f1e, f2e, coords = corr_case_inputs(12, 1, 16, 18, 22, 5)
        rs = np.random.RandomState(13)
        f1i = rs.standard_normal(f1e.shape).astype(np.float32)
        cce = ns.CorrComputation(torch.from_numpy(f1e), torch.from_numpy(f2e[:4]), num_levels_per_target=[1, 1, 1, 4])
        cci = ns.CorrComputation(torch.from_numpy(f1i), torch.from_numpy(f2e[4]), num_levels_per_target=4)
        blk = ns.CorrBlockParallelMultiTarget(corr_computation_events=cce, corr_computation_frames=cci)
        look = blk(list(torch.from_numpy(coords)))
        pyr = {f"pyr{L}": d.corr.numpy() for L, d in enumerate(blk._corr_pyramid) if L > 0}
        pyr["pyr0_rows7"] = blk._corr_pyramid[0].corr.numpy()[:, ::7]   # every 7th query pixel (fixture size)
        np.savez_compressed(os.path.join(out_dir, "corr_MtoN.npz"), f1_ev=f1e, f2_ev=f2e[:4], f1_img=f1i,
                            f2_img=f2e[4], coords=coords, levels=np.array([1, 1, 1, 4, 4]), lookup=look.numpy(), **pyr)

        # ---------------- bezier ----------------
        bz = {}
        for deg in (2, 10):
            rs = np.random.RandomState(20 + deg)
            p = rs.standard_normal((2, 2 * deg, 5, 6)).astype(np.float32)
            curves = ns.BezierCurves(torch.from_numpy(p))
            ts = [0.25, 0.5, 0.75, 1.0] if deg == 2 else [0.2, 0.4, 0.6, 0.8, 1.0, 1]
            bz[f"params_d{deg}"] = p
            bz[f"times_d{deg}"] = np.array(ts, dtype=np.float64)
            bz[f"flow_list_d{deg}"] = curves.get_flow_from_reference(ts).numpy()
            bz[f"flow_0_d{deg}"] = curves.get_flow_from_reference(0.0).numpy()
            bz[f"flow_1_d{deg}"] = curves.get_flow_from_reference(1.0).numpy()
            bz[f"flow_03_d{deg}"] = curves.get_flow_from_reference(0.3).numpy()
        np.savez_compressed(os.path.join(out_dir, "bezier.npz"), **bz)

        # ---------------- convex upsampling ----------------
        rs = np.random.RandomState(31)
        data = rs.standard_normal((2, 4, 5, 6)).astype(np.float32)
        mask = (rs.standard_normal((2, 576, 5, 6)) * 2).astype(np.float32)
        np.savez_compressed(os.path.join(out_dir, "cvx_upsample.npz"), data=data, mask=mask,
                            out=ns.cvx_upsample(torch.from_numpy(data), torch.from_numpy(mask)).numpy())

        # ---------------- voxel grid + normalisation ----------------
        vg = {}
        C, Hh, Ww = 5, 24, 32
        t0c, t1c = 1_000_000, 1_100_000
        conv = ns.VoxelGrid(C, Hh, Ww)
        ts, te = conv.get_extended_time_window(t0c, t1c)
        vg["window"] = np.array([ts, te, t0c, t1c], dtype=np.int64)
        for tag, int_xy in (("f", False), ("i", True)):
            x, y, pol, t = synthetic.events(6000, Hh, Ww, ts - 3000, te + 3000, seed=41, int_xy=int_xy)
            g = conv.convert(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(pol), torch.from_numpy(t), t0c, t1c)
            vg[f"x_{tag}"], vg[f"y_{tag}"], vg[f"pol_{tag}"], vg[f"t_{tag}"] = x, y, pol, t
            vg[f"grid_{tag}"] = g.numpy().copy()
            vg[f"norm_{tag}"] = ns.norm_voxel_grid(g.clone()).numpy()
        z = torch.zeros(3, 4, 5)
        vg["norm_allzero"] = ns.norm_voxel_grid(z.clone()).numpy()
        one = torch.zeros(3, 4, 5)
        one[1, 2, 3] = 2.5
        one[0, 0, 0] = 2.5          # two equal non-zeros -> std == 0 branch
        vg["norm_std0_in"] = one.numpy().copy()
        vg["norm_std0"] = ns.norm_voxel_grid(one.clone()).numpy()
        np.savez_compressed(os.path.join(out_dir, "voxel.npz"), **vg)

        # ---------------- EPE ----------------
        rs = np.random.RandomState(51)
        a = rs.standard_normal((3, 2, 9, 11)).astype(np.float32) * 4
        b = rs.standard_normal((3, 2, 9, 11)).astype(np.float32) * 4
        m = rs.uniform(size=(3, 9, 11)) < 0.6
        np.savez_compressed(os.path.join(out_dir, "epe.npz"), a=a, b=b, mask=m,
                            epe=ns.epe_masked(torch.from_numpy(a), torch.from_numpy(b)).numpy(),
                            epe_masked=ns.epe_masked(torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(m)).numpy())

        # ---------------- validation metrics (SURVEY f-3): AE, NPE, EPE_MULTI / AE_MULTI, linear-assumption predictions ----------------
        rs = np.random.RandomState(61)
        M = 4
        preds = [rs.standard_normal((2, 2, 13, 17)).astype(np.float32) * 3 for _ in range(M)]
        gts = [(p + rs.standard_normal(p.shape).astype(np.float32) * s) for p, s in zip(preds, (0.05, 0.5, 2.0, 6.0))]
        gts[1][0, :, 0, 0] = 0.0                       # zero ground-truth vector: clip(|gt|, 1e-6) branch of NPE
        preds[2][1, :, 3, 3] = gts[2][1, :, 3, 3]      # exact hit: cosine clamps at 1
        masks = [rs.uniform(size=(2, 13, 17)) < q for q in (0.7, 0.5, 0.9, 0.0)]   # the last mask is EMPTY
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        mt = {"M": np.int64(M)}
        for i in range(M):
            mt[f"pred{i}"], mt[f"gt{i}"], mt[f"mask{i}"] = preds[i], gts[i], masks[i]
            mt[f"ae{i}"] = ns.ae_masked(T(preds[i]), T(gts[i])).numpy()
            mt[f"ae_rad{i}"] = ns.ae_masked(T(preds[i]), T(gts[i]), None, degrees=False).numpy()
            for n in (1, 2, 3):
                mt[f"npe{n}_{i}"] = ns.n_pixel_error_masked(T(preds[i]), T(gts[i]), None, n).numpy()
            if masks[i].any():
                mt[f"ae_m{i}"] = ns.ae_masked(T(preds[i]), T(gts[i]), T(masks[i])).numpy()
                for n in (1, 2, 3):
                    mt[f"npe{n}_m{i}"] = ns.n_pixel_error_masked(T(preds[i]), T(gts[i]), T(masks[i]), n).numpy()
        mt["epe_multi"] = ns.epe_masked_multi([T(p) for p in preds], [T(g) for g in gts]).numpy()
        mt["epe_multi_m"] = ns.epe_masked_multi([T(p) for p in preds], [T(g) for g in gts], [T(m) for m in masks]).numpy()   # skips the empty mask
        mt["ae_multi"] = ns.ae_masked_multi([T(p) for p in preds], [T(g) for g in gts]).numpy()
        mt["ae_multi_m3"] = ns.ae_masked_multi([T(p) for p in preds[:3]], [T(g) for g in gts[:3]], [T(m) for m in masks[:3]]).numpy()
        mt["traj_len"] = ns.EPE_MULTI.compute_traj_len([T(g) for g in gts]).numpy()
        em = ns.EPE_MULTI(min_traj_len=4.0, max_traj_len=30.0)
        em.update([T(p) for p in preds[:3]], [T(g) for g in gts[:3]], [T(m) for m in masks[:3]])
        mt["epe_multi_traj_4_30"] = em.compute().numpy()
        ts = [0.25, 0.5, 0.75, 1.0]
        lin = ns.predictions_from_lin_assumption(T(preds[3]), ts)
        mt["lin_ts"] = np.array(ts, dtype=np.float64)
        mt["epe_multi_lin"] = ns.epe_masked_multi(lin, [T(g) for g in gts]).numpy()
        np.savez_compressed(os.path.join(out_dir, "metrics.npz"), **mt)

        # ---------------- InputPadder (replicate padding to a multiple of 8) ----------------
        pd = {}
        rs = np.random.RandomState(71)
        for tag, (hh, ww), no_top in (("a", (21, 30), False), ("b", (21, 30), True), ("c", (24, 32), False), ("d", (17, 9), False)):
            x = rs.standard_normal((2, 3, hh, ww)).astype(np.float32)
            padder = ns.InputPadder(min_size=8, no_top_padding=no_top)
            y = padder.pad(T(x))
            pd[f"x_{tag}"], pd[f"y_{tag}"], pd[f"pad_{tag}"] = x, y.numpy(), np.array(padder._pad, dtype=np.int64)
            pd[f"no_top_{tag}"] = np.bool_(no_top)
            assert np.array_equal(padder.unpad(y).numpy(), x)
        np.savez_compressed(os.path.join(out_dir, "padder.npz"), **pd)

        # ---------------- DSEC two-step sample assembly (SURVEY f-1): the reference's TwoStepSubSequence.__getitem__ on an in-memory stream ----
        dns = refshim.import_reference_dsec()
        Hd, Wd, bins = 40, 56, 5
        rs = np.random.RandomState(81)
        n_ev = 30000
        ev = dict(x=rs.randint(0, Wd, n_ev).astype(np.uint16), y=rs.randint(0, Hd, n_ev).astype(np.uint16),
                  p=rs.randint(0, 2, n_ev).astype(np.uint8), t=np.sort(rs.randint(2_000_000, 2_400_000, n_ev)).astype(np.int64))
        yy, xx = np.meshgrid(np.arange(Hd), np.arange(Wd), indexing="ij")
        # rectification map: identity + smooth distortion + jitter, partly pointing OUTSIDE the image (dropped corners of the tri-linear splat)
        rect = np.stack([xx * 1.03 - 1.2 + rs.uniform(-0.4, 0.4, (Hd, Wd)), yy * 0.97 + 0.8 + rs.uniform(-0.4, 0.4, (Hd, Wd))], -1).astype(np.float32)
        ts = np.array([[2_100_000, 2_200_000], [2_200_000, 2_300_000]], dtype=np.int64)
        ds = dict(x=ev["x"], y=ev["y"], p=ev["p"], t=ev["t"], rectify_map=rect, forward_flow_timestamps=ts, num_bins=np.int64(bins))
        for tag, norm, merge in (("nm", True, True), ("m", False, True), ("n", True, False)):
            drv = refshim.ReferenceTwoStepDriver(dns, ev, rect, ts, bins, Hd, Wd, normalize=norm, merge=merge)
            for idx in (0, 1):
                ds[f"sample_{tag}_{idx}"] = drv.sample(idx).numpy()
        ds["offsets_2150000_2250000"] = np.array(dns.EventSlicer.get_time_indices_offsets(ev["t"], 2_150_000, 2_250_000), dtype=np.int64)
        ds["offsets_past_end"] = np.array(dns.EventSlicer.get_time_indices_offsets(ev["t"], 2_500_000, 2_600_000), dtype=np.int64)
        np.savez_compressed(os.path.join(out_dir, "dsec_twostep.npz"), **ds)
    training_goldens(ns, out_dir)
    print("golden fixtures written to", out_dir)


if __name__ == "__main__":
    main()
