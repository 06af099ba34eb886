#!/usr/bin/env python
"""Training-step throughput of the f-4 path on one MI355X at the reference's DSEC training shape (E_LU4_BD2, batch 3, crop 288x384,
12 iterations, AdamW; config/general.yaml, data/dsec/subsequence/base.py:60).  Tools only.  Usage: python tools/train_probe.py [steps]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bflow_amd
from bflow_amd import configs, synthetic, training
from bflow_amd.validation import DataLoading, DataSetType
from bflow_amd.weights import deterministic_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, H, W, ITERS = 3, 288, 384, 12
dev = torch.device("cuda:0")
model = bflow_amd.RAFTSpline(configs.model_config("E_LU4_BD2"))
model.load_state_dict(deterministic_state_dict(model, seed=0)); model.to(dev).train()
opt, sch = training.configure_optimizers(model, dict(learning_rate=1e-4, weight_decay=1e-4, lr_scheduler=dict(use=True, total_steps=1000, pct_start=0.01)))
batch = {DataLoading.EV_REPR: torch.from_numpy(synthetic.voxel_grid(B, 9, H, W, seed=1)).to(dev),
         DataLoading.FLOW: torch.from_numpy(synthetic.gt_flow(B, H, W, seed=2)).to(dev),
         DataLoading.FLOW_VALID: torch.rand(B, H, W, device=dev) < 0.8, DataLoading.DATASET_TYPE: [DataSetType.DSEC]}
step = training.TrainStep(model, num_iter_train=ITERS)
ev = lambda: torch.cuda.Event(enable_timing=True)
def one():
    e = [ev() for _ in range(4)]
    opt.zero_grad(set_to_none=True)
    e[0].record(); out = step(batch); e[1].record(); out["loss"].backward(); e[2].record(); opt.step(); sch.step(); e[3].record()
    return e, out["loss"]
for _ in range(3): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
recs = [one() for _ in range(steps)]
t_host = (time.perf_counter() - t0) / steps          # time the host needs to ENQUEUE a step (== the step time when the host is the bound)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
f = sum(e[0].elapsed_time(e[1]) for e, _ in recs) / steps; b = sum(e[1].elapsed_time(e[2]) for e, _ in recs) / steps; o = sum(e[2].elapsed_time(e[3]) for e, _ in recs) / steps
print(f"train step B={B} {H}x{W} iters={ITERS}: {dt*1e3:.1f} ms/step = {B/dt:.1f} samples/s  (forward+loss {f:.1f} ms, backward {b:.1f} ms, AdamW {o:.1f} ms); "
      f"loss {float(recs[0][1].detach()):.4f} -> {float(recs[-1][1].detach()):.4f}; peak memory {torch.cuda.max_memory_allocated()/2**30:.2f} GiB; host enqueue {t_host*1e3:.1f} ms/step")
if os.environ.get("BFLOW_TRAIN_PROBE_GRAPH"):        # the same step as ONE hipGraph (training.GraphedTrainStep)
    for tag, on in (("engine", True), ("torch convolutions", False)):
        if os.environ["BFLOW_TRAIN_PROBE_GRAPH"] in ("engine", "torch") and (os.environ["BFLOW_TRAIN_PROBE_GRAPH"] == "engine") != on:
            continue
        from bflow_amd import conv_train
        conv_train.ENABLED = on
        m2 = bflow_amd.RAFTSpline(configs.model_config("E_LU4_BD2"))
        m2.load_state_dict(deterministic_state_dict(m2, seed=0)); m2.to(dev).train()
        o2, s2 = training.configure_optimizers(m2, dict(learning_rate=1e-4, weight_decay=1e-4, lr_scheduler=dict(use=True, total_steps=1000, pct_start=0.01)), capturable=True)
        g = training.GraphedTrainStep(training.TrainStep(m2, num_iter_train=ITERS), o2, s2)
        l0 = float(g(batch)["loss"].detach())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): out = g(batch)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print(f"  one hipGraph per step, {tag}: {dt*1e3:.1f} ms/step = {B/dt:.1f} samples/s; loss {l0:.4f} -> {float(out['loss'].detach()):.4f}")
        del g, m2, o2, s2
    conv_train.ENABLED = True
if os.environ.get("BFLOW_TRAIN_PROBE_AB"):           # the same step on torch / MIOpen convolutions
    from bflow_amd import conv_train
    conv_train.ENABLED = False
    for _ in range(3): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): one()
    t_host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"  torch convolutions (conv_train.ENABLED = False): {dt*1e3:.1f} ms/step, host enqueue {t_host*1e3:.1f} ms/step")
