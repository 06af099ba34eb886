"""Test infrastructure: import the upstream reference (uzh-rpg/bflow) from /root/reference.

Only usable in the build container (the reference never travels to the GPU box).
Two third-party imports of the reference are absent here and are stubbed:
  * numba.jit      (models/raft_spline/bezier.py:8,148)  -> identity decorator
  * omegaconf.ListConfig (models/raft_utils/corr.py:8,147) -> dummy class (isinstance check only)
  * torchmetrics.Metric  (utils/metrics.py:5) -> dummy base (only pure functions are used)
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("BFLOW_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "raft_spline", "raft.py"))


def _install_stubs():
    if "numba" not in sys.modules:
        numba = types.ModuleType("numba")

        def jit(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            return lambda f: f

        numba.jit = jit
        sys.modules["numba"] = numba
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")

        class ListConfig(list):
            pass

        class DictConfig(dict):
            pass

        oc.ListConfig = ListConfig
        oc.DictConfig = DictConfig
        sys.modules["omegaconf"] = oc
    if "torchmetrics" not in sys.modules:
        tm = types.ModuleType("torchmetrics")

        class Metric:
            def __init__(self, *a, **k):
                pass

            def add_state(self, name, default, dist_reduce_fx=None):
                setattr(self, name, default)

        tm.Metric = Metric
        sys.modules["torchmetrics"] = tm


def import_reference():
    """Returns a namespace with the reference's hot-path classes/functions."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    _install_stubs()
    import torch
    nthreads = torch.get_num_threads()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    from models.raft_spline.raft import RAFTSpline
    from models.raft_spline.bezier import BezierCurves
    from models.raft_spline.update import BasicUpdateBlock
    from models.raft_utils.corr import CorrComputation, CorrBlockParallelMultiTarget
    from models.raft_utils.extractor import BasicEncoder
    from models.raft_utils.utils import bilinear_sampler, coords_grid, cvx_upsample
    # NOTE: importing representations pins torch to 1 thread (representations.py:5-6); undo it.
    from data.utils.representations import VoxelGrid, norm_voxel_grid
    torch.set_num_threads(nthreads)
    from utils.metrics import (epe_masked, epe_masked_multi, ae_masked, ae_masked_multi, n_pixel_error_masked, predictions_from_lin_assumption,
                               EPE_MULTI)
    from modules.utils import InputPadder
    ns.RAFTSpline = RAFTSpline
    ns.BezierCurves = BezierCurves
    ns.BasicUpdateBlock = BasicUpdateBlock
    ns.CorrComputation = CorrComputation
    ns.CorrBlockParallelMultiTarget = CorrBlockParallelMultiTarget
    ns.BasicEncoder = BasicEncoder
    ns.bilinear_sampler = bilinear_sampler
    ns.coords_grid = coords_grid
    ns.cvx_upsample = cvx_upsample
    ns.VoxelGrid = VoxelGrid
    ns.norm_voxel_grid = norm_voxel_grid
    ns.epe_masked = epe_masked
    ns.epe_masked_multi, ns.ae_masked, ns.ae_masked_multi = epe_masked_multi, ae_masked, ae_masked_multi
    ns.n_pixel_error_masked, ns.predictions_from_lin_assumption = n_pixel_error_masked, predictions_from_lin_assumption
    ns.EPE_MULTI, ns.InputPadder = EPE_MULTI, InputPadder
    return ns
