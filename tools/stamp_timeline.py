#!/usr/bin/env python
"""Timeline of one hipGraph replay of the forward WITHOUT the profiler: one-thread kernels write the 100 MHz wall clock at
stage boundaries (tools/stamp/).  Usage: python tools/stamp_timeline.py [batch]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bflow_amd
from bflow_amd import configs, synthetic
from bflow_amd.weights import deterministic_state_dict
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "stamp", "libstamp.so"))
lib.stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = bflow_amd.RAFTSpline(configs.model_config("E_LU4_BD2")).eval()
model.load_state_dict(deterministic_state_dict(model, seed=0)); model.to(dev); model.enable_hipgraph()
slots = torch.zeros(64, dtype=torch.int64, device=dev); names = []
def probe(name):
    if name not in names: names.append(name)
    lib.stamp(slots.data_ptr() + 8 * names.index(name), torch.cuda.current_stream().cuda_stream)
model._probe = probe
vox = torch.from_numpy(synthetic.voxel_grid(B, 9, 480, 640, seed=1234)).to(dev)
for _ in range(6): model(voxel_grid=vox, iters=12, test_mode=True)
torch.cuda.synchronize()
t = slots.cpu().tolist(); t0 = min(t[i] for i in range(len(names)))
for i, n in sorted(enumerate(names), key=lambda kv: t[kv[0]]):
    print(f"{(t[i] - t0) / 100:9.1f} us  {n}")
