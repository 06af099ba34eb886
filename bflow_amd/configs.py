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


def compose(model: str = "raft-spline", dataset: Optional[str] = None, experiment: Optional[str] = None,
            overrides: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """experiment: path under config/experiment without .yaml, e.g. 'dsec/raft_spline/E_LU4_BD2_lowpyramid'."""
    cfg: Dict[str, Any] = {}
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


def model_config(short_name: str) -> Dict[str, Any]:
    """config['model'] of one of the four shipped experiments (BASELINE.json configs)."""
    dataset, exp = EXPERIMENTS[short_name]
    return compose(dataset=dataset, experiment=exp)["model"]
