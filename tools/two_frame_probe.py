#!/usr/bin/env python
"""Can two batch-1 forwards overlap inside ONE hipGraph (tools only)?  Captures (a) one forward, (b) two forwards one after the other, (c) two
forwards as parallel branches of one graph, and times the replays."""
import faulthandler, gc, os, sys, time
faulthandler.enable()
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bflow_amd
from bflow_amd import configs, synthetic, hip
from bflow_amd.weights import deterministic_state_dict
dev = torch.device("cuda:0")
m = bflow_amd.RAFTSpline(configs.model_config("E_LU4_BD2")).eval()
m.load_state_dict(deterministic_state_dict(m, 0)); m.to(dev)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 1
va = torch.from_numpy(synthetic.voxel_grid(BATCH, 9, 480, 640, seed=1234)).to(dev)
vb = torch.from_numpy(synthetic.voxel_grid(BATCH, 9, 480, 640, seed=4321)).to(dev)
def fwd(v):
    return m._forward_impl(v, None, 12, None, True)
with torch.no_grad():
    for _ in range(3):
        fwd(va); fwd(vb)
    torch.cuda.synchronize()
    def capture(fn):
        g = torch.cuda.CUDAGraph()
        gc.collect(); gc.disable()
        try:
            with torch.cuda.graph(g):
                out = fn()
        finally:
            gc.enable()
        return g, out
    def one(): return fwd(va)
    def seq(): return fwd(va), fwd(vb)
    outer = torch.cuda.Stream()
    def par():
        cur = torch.cuda.current_stream()
        outer.wait_stream(cur)
        with torch.cuda.stream(outer):
            a = fwd(va)
        b = fwd(vb)
        cur.wait_stream(outer)
        return a, b
    res = {}
    print("warm", flush=True)
    def par_mixed():                      # outer frame without inner branches, main frame with them: 3 streams, all forked from the capture stream
        cur = torch.cuda.current_stream()
        outer.wait_stream(cur)
        with torch.cuda.stream(outer):
            hip.BRANCHING = False
            try:
                a = fwd(va)
            finally:
                hip.BRANCHING = True
        b = fwd(vb)
        cur.wait_stream(outer)
        return a, b
    def par_flat():
        hip.BRANCHING = False
        try:
            return par()
        finally:
            hip.BRANCHING = True
    def one_flat():
        hip.BRANCHING = False
        try:
            return fwd(va)
        finally:
            hip.BRANCHING = True
    for name, fn, frames in (("one", one, 1), ("one, no inner branches", one_flat, 1), ("two sequential", seq, 2), ("two parallel, no inner branches", par_flat, 2), ("two parallel, flat 3 streams", par_mixed, 2)):
        print('capturing', name, flush=True)
        g, out = capture(fn)
        print('captured', flush=True)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20): g.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20)
        print(f"{name:24s}: {best*1e3:7.3f} ms per replay = {frames*BATCH/best:7.1f} frames/s")
        res[name] = out
    a_seq, b_seq = res["two sequential"]
    a_par, b_par = res["two parallel, no inner branches"]
    print("max |diff| parallel vs sequential:", float((a_seq[0] - a_par[0]).abs().max()), float((b_seq[0] - b_par[0]).abs().max()))
