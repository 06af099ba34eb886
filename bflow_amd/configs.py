"""The reference's Hydra config surface without Hydra (hydra/omegaconf are optional and absent in this image).

`bflow_amd/config/` ships the same YAML tree (same keys and values as the reference's `config/`, so `val.py`-style
`model=raft-spline dataset=dsec +experiment/dsec/raft_spline=E_LU4_BD2_lowpyramid` selections mean the same thing);
`compose()` implements the small subset of Hydra's defaults-list semantics the tree uses: `defaults:` inheritance inside a
group, `# @package _global_` experiment files merged over the root, `override /model: ...`, deep dict merge.  The result
`cfg['model']` is the plain dict `RAFTSpline(model_params)` consumes (models/raft_spline/raft.py:15-53).  When Hydra IS
installed the YAML tree can be used with it directly."""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, Optional

import yaml

CONFIG_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")

# num_bins.correlation is null in the YAML and back-filled by the DataModule (modules/data_loading.py:63-68):
# DSEC -> same as context (data/dsec/provider.py:24-25,70-71); MultiFlow -> table of data/multiflow2d/sample.py:41-46
MULTIFLOW_CORR_BINS = {6: 4, 11: 7, 21: 13, 41: 25}


def _merge(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _load_group(group: str, name: str) -> Dict[str, Any]:
    with open(os.path.join(CONFIG_ROOT, group, name + ".yaml")) as f:
        raw = yaml.safe_load(f) or {}
    out: Dict[str, Any] = {}
    for d in raw.pop("defaults", []) or []:
        if isinstance(d, str) and d != "_self_":
            _merge(out, _load_group(group, d))
    return _merge(out, raw)


def _load_root(name: str) -> Dict[str, Any]:
    """A root-level file (train.yaml / val.yaml / general.yaml): root-level entries of its defaults list are merged in order, group
    entries (`dataset: ???`, `model: ???`) are filled by compose(), `_self_` marks where the file's own keys go (Hydra: last by default)."""
    with open(os.path.join(CONFIG_ROOT, name + ".yaml")) as f:
        raw = yaml.safe_load(f) or {}
    out: Dict[str, Any] = {}
    own_done = False
    for d in raw.pop("defaults", []) or []:
        if d == "_self_":
            _merge(out, raw)
            own_done = True
        elif isinstance(d, str):
            _merge(out, _load_root(d))
    if not own_done:
        _merge(out, raw)
    return out


def _resolve_interpolations(node: Any, root: Dict[str, Any], path=()):
    """The one OmegaConf feature the tree uses: `${a.b}` (absolute) and `${..key}` (relative: one dot = the containing node, each further
    dot one level up) as a WHOLE value -- general.yaml: training.lr_scheduler.total_steps = ${..max_steps}."""
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve_interpolations(node[k], root, path + (k,))
        return node
    if isinstance(node, str) and node.startswith("${") and node.endswith("}"):
        ref = node[2:-1]
        if ref.startswith("."):
            ups = len(ref) - len(ref.lstrip("."))
            base = list(path[:-1])                       # the containing node of this key
            base = base[:len(base) - (ups - 1)] if ups > 1 else base
            keys = base + [k for k in ref.lstrip(".").split(".") if k]
        else:
            keys = ref.split(".")
        cur: Any = root
        for k in keys:
            cur = cur[k]
        return _resolve_interpolations(copy.deepcopy(cur), root, tuple(keys))
    return node


def compose(model: str = "raft-spline", dataset: Optional[str] = None, experiment: Optional[str] = None,
            overrides: Optional[Dict[str, Any]] = None, root: Optional[str] = None) -> Dict[str, Any]:
    """experiment: path under config/experiment without .yaml, e.g. 'dsec/raft_spline/E_LU4_BD2_lowpyramid'.
    root: 'train' | 'val' -- start from that entry file (train.yaml pulls in general.yaml: `training`, `hardware`, `logging`, `wandb`,
    `debugging`), as `python train.py model=... dataset=... +experiment/...=...` does in the reference (train.py:26, val.py:22)."""
    cfg: Dict[str, Any] = _load_root(root) if root is not None else {}
    if dataset is not None:
        cfg["dataset"] = _load_group("dataset", dataset)
    if experiment is not None:
        with open(os.path.join(CONFIG_ROOT, "experiment", experiment + ".yaml")) as f:
            exp = yaml.safe_load(f) or {}
        for d in exp.pop("defaults", []) or []:
            if isinstance(d, dict) and "override /model" in d:
                model = d["override /model"]
        cfg["model"] = _load_group("model", model)
        _merge(cfg, exp)
    else:
        cfg["model"] = _load_group("model", model)
    if overrides:
        _merge(cfg, overrides)
    _resolve_interpolations(cfg, cfg)
    nb = cfg["model"]["num_bins"]
    if nb.get("correlation") is None:
        ds = (cfg.get("dataset") or {}).get("name", "dsec")
        nb["correlation"] = nb["context"] if ds == "dsec" else MULTIFLOW_CORR_BINS[nb["context"]]
    return cfg


EXPERIMENTS = {
    "E_LU4_BD2": ("dsec", "dsec/raft_spline/E_LU4_BD2_lowpyramid"),
    "E_I_LU4_BD2": ("dsec", "dsec/raft_spline/E_I_LU4_BD2_lowpyramid"),
    "E_LU5_BD10": ("multiflow_regen", "multiflow/raft_spline/E_LU5_BD10_lowpyramid"),
    "E_I_LU5_BD10": ("multiflow_regen", "multiflow/raft_spline/E_I_LU5_BD10_lowpyramid"),
}


def model_config(short_name: str, overrides: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """config['model'] of one of the four shipped experiments (BASELINE.json configs)."""
    dataset, exp = EXPERIMENTS[short_name]
    return compose(dataset=dataset, experiment=exp, overrides=None if overrides is None else {"model": overrides})["model"]


# BASELINE.json `configs`, in its order: experiment, input size, batch per GPU, GRU iterations and the model-key overrides the entry names.
# configs[4] asks for "fp16 MFMA correlation" on top of E_I_LU5_BD10 (whose YAML, like the reference's raft.py:122, says nothing about
# precision): it selects `correlation.precision = "f16/w"` -- plain fp16 operands, ONE fp16 MFMA pass, fp32 accumulation AND an fp32 tiled
# volume -- the fp16-MFMA arithmetic that stays inside the north star's 1e-3 px (5.3e-4 px at 1024^2; an fp16 VOLUME does not: 2.3e-3 px,
# "f16", opt-in).
BASELINE_CONFIGS = (
    dict(name="C1", experiment="E_LU5_BD10", height=384, width=384, batch=1, iters=4, overrides=None),
    dict(name="C2", experiment="E_LU4_BD2", height=480, width=640, batch=1, iters=12, overrides=None),
    dict(name="C3", experiment="E_I_LU4_BD2", height=480, width=640, batch=8, iters=12, overrides=None),
    dict(name="C4", experiment="E_LU4_BD2", height=480, width=640, batch=8, iters=12, overrides=None),
    dict(name="C5", experiment="E_I_LU5_BD10", height=1024, width=1024, batch=1, iters=20, overrides={"correlation": {"precision": "f16/w"}}),
)


def baseline_config(index: int) -> Dict[str, Any]:
    """BASELINE.json configs[index] -> dict(name, experiment, height, width, batch, iters, model = the composed config['model'])."""
    entry = dict(BASELINE_CONFIGS[index])
    entry["model"] = model_config(entry["experiment"], entry["overrides"])
    return entry
