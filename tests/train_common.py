"""Shared by the CPU (oracle vs reference goldens) and GPU (HIP path vs oracle) training tests."""
import numpy as np
import torch

from bflow_amd import synthetic
from oracle import raft_spline_oracle as O

TRAIN_CASES = ["train_E_LU4_BD2", "train_E_I_LU4_BD2", "train_E_LU5_BD10", "train_E_LU4_BD2_detach_init"]
GRAD_STRIDE = 97


def train_targets(B, H, W, kind, seed=99):
    """Same synthetic ground truth as tests/golden/make_golden.py::train_targets."""
    rs = np.random.RandomState(seed)
    if kind == "dsec":
        return [synthetic.gt_flow(B, H, W, seed=seed)], [rs.rand(B, H, W) < 0.8], [1.0]
    times = [0.4, 0.7, 1.0]
    return [synthetic.gt_flow(B, H, W, seed=seed + k) * t for k, t in enumerate(times)], None, times


def case_setup(name, cfg, B, H, W):
    """(cfg possibly with detach_bezier, flow_init array or None) -- mirrors tests/golden/make_golden.py."""
    if name.endswith("detach_init"):
        cfg = dict(cfg, detach_bezier=True)
        return cfg, (np.random.RandomState(55).standard_normal((B, 2 * cfg["bezier_degree"], H // 8, W // 8)) * 1.5).astype(np.float32)
    return cfg, None


def inputs(cfg, B, H, W):
    C = cfg["num_bins"]["context"] + cfg["num_bins"]["correlation"] - 1
    vox = torch.from_numpy(synthetic.voxel_grid(B, C, H, W, seed=1234))
    imgs = None
    if cfg["use_boundary_images"]:
        a, b = synthetic.image_pair(B, H, W, seed=4321)
        imgs = [torch.from_numpy(a), torch.from_numpy(b)]
    return vox, imgs


def oracle_train_step(cfg, B, H, W, iters, kind, seed=0, flow_init=None):
    """Training-mode forward + loss + backward on the CPU oracle -> (loss, {param: grad}, {buffer: value}, last prediction)."""
    sd = {k: v.clone() for k, v in O.make_state_dict(cfg, seed=seed).items()}
    shapes = O.param_shapes(cfg)
    params = [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]
    for k in params:
        sd[k].requires_grad_(True)
    vox, imgs = inputs(cfg, B, H, W)
    gts, valids, times = train_targets(B, H, W, kind)
    ups = O.forward(sd, cfg, vox, imgs, iters=iters, test_mode=False, training=True,
                    flow_init=None if flow_init is None else torch.from_numpy(flow_init))
    if kind == "dsec":
        flows = [O.bezier_flow(u, 1.0) for u in ups]
        loss = O.l1_seq_loss_channel_masked(flows, torch.from_numpy(gts[0]), torch.from_numpy(valids[0]))
    else:
        flows = [[O.bezier_flow(u, t) for t in times] for u in ups]
        loss = O.l1_multi_seq_loss_channel_masked(flows, [torch.from_numpy(g) for g in gts])
    loss.backward()
    grads = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in params}
    bufs = {k: v.detach() for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
    return loss.detach(), grads, bufs, ups[-1].detach()


def check_grads(grads, golden, rel=2e-3):
    """grads {name: tensor} against the golden's (norm, sum, strided subsample) per parameter; returns the worst relative error."""
    grads = {k: g for k, g in grads.items() if f"gnorm/{k}" in golden}      # aliases of one module (norm3 / downsample.1) appear once
    assert len(grads) > 50
    worst = 0.0
    total = np.sqrt(sum(float(golden[f"gnorm/{k}"]) ** 2 for k in grads))
    for k, g in grads.items():
        g = g.detach().float().cpu()
        ref_norm = float(golden[f"gnorm/{k}"])
        sub = g.flatten()[::GRAD_STRIDE].numpy()
        ref_sub = golden[f"gsub/{k}"]
        # error of the subsample relative to the parameter's own gradient scale (floor: 1e-6 of the whole gradient's norm)
        scale = max(ref_norm / np.sqrt(max(g.numel(), 1)), 1e-6 * total / np.sqrt(max(g.numel(), 1)))
        err = float(np.abs(sub - ref_sub).max()) / (scale * 30.0)
        nerr = abs(float(g.norm()) - ref_norm) / max(ref_norm, 1e-6 * total)
        worst = max(worst, err, nerr)
        assert nerr < rel * 5, f"{k}: gradient norm {float(g.norm())} vs reference {ref_norm}"
        assert err < rel * 5, f"{k}: gradient subsample deviates (scaled error {err})"
    return worst
