"""Deterministic, torch-RNG-independent weights for benchmarks and parity tests (there is no network to fetch the
published checkpoints, README.md:63-95 of the reference; when a real .ckpt is available, `load_lightning_checkpoint`
loads it unchanged because the parameter names are the reference's).

`deterministic_state_dict(module, seed)` walks the module's own state dict in SORTED name order and fills every tensor
from numpy RandomState(seed): conv weights ~ N(0, sqrt(2/fan_out)) x a per-layer gain that keeps the recurrent loop in
a numerically meaningful regime (un-saturated GRU, ~0.2 px updates per iteration), biases ~ U(-0.05, 0.05), BatchNorm
affine/running statistics near identity.  tests/test_host_logic.py checks it equals the oracle's independent filler."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

LAYER_GAINS = (("update_block.bezier_head.conv2.", 0.1), ("update_block.gru.", 0.2),
               ("update_block.encoder.convc1.", 0.1), ("cnet.conv2.", 0.1))


def deterministic_state_dict(module: torch.nn.Module, seed: int = 0) -> Dict[str, torch.Tensor]:
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    rs = np.random.RandomState(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(100, dtype=torch.int64)
            continue
        if ".downsample.1." in name:   # same module as .norm3 (extractor.ResidualBlock)
            continue
        leaf = name.rsplit(".", 1)[1]
        if len(shp) == 4:
            g = 1.0
            for pfx, lg in LAYER_GAINS:
                if name.startswith(pfx):
                    g *= lg
            arr = rs.standard_normal(shp) * math.sqrt(2.0 / (shp[0] * shp[2] * shp[3])) * g
        elif leaf == "running_var":
            arr = rs.uniform(0.5, 1.5, shp)
        elif leaf == "running_mean":
            arr = rs.uniform(-0.1, 0.1, shp)
        elif leaf == "weight":
            arr = rs.uniform(0.9, 1.1, shp)
        else:
            arr = rs.uniform(-0.05, 0.05, shp)
        sd[name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
    for name in shapes:
        if ".downsample.1." in name:
            sd[name] = sd[name.replace(".downsample.1.", ".norm3.")]
    return {k: sd[k] for k in shapes}


def load_lightning_checkpoint(module: torch.nn.Module, path: str, strict: bool = True):
    """Loads a reference checkpoint (Lightning `state_dict` with `net.` prefixes: modules/raft_spline.py:24, val.py:58)."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt.get("state_dict", ckpt)
    sd = {k[len("net."):]: v for k, v in sd.items() if k.startswith("net.")} or sd
    return module.load_state_dict(sd, strict=strict)
