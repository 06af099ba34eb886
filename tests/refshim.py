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
    from utils.losses import l1_loss_channel_masked, l1_seq_loss_channel_masked, l1_multi_seq_loss_channel_masked
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
    ns.l1_loss_channel_masked, ns.l1_seq_loss_channel_masked = l1_loss_channel_masked, l1_seq_loss_channel_masked
    ns.l1_multi_seq_loss_channel_masked = l1_multi_seq_loss_channel_masked
    return ns


def import_reference_dsec():
    """The reference's DSEC sample-assembly classes.  Their modules import I/O libraries that are absent here
    (h5py, imageio, cv2, skimage, torchvision); those are stubbed as EMPTY modules -- every function of theirs the assembly would
    call (file reads) is replaced by the test's synthetic-data providers, the arithmetic is the reference's own."""
    _install_stubs()
    for name in ("h5py", "imageio", "cv2", "skimage"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].img_as_ubyte = lambda x: x
    sys.modules["h5py"].File = type("File", (), {})                              # only used in type annotations on this path
    if "pytorch_lightning" not in sys.modules:                                   # utils/general.py:7 imports a callback class it never uses here
        pl = types.ModuleType("pytorch_lightning")
        plc = types.ModuleType("pytorch_lightning.callbacks")
        plc.ModelCheckpoint = type("ModelCheckpoint", (), {})
        pl.callbacks = plc
        sys.modules["pytorch_lightning"], sys.modules["pytorch_lightning.callbacks"] = pl, plc
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tvt.ColorJitter = type("ColorJitter", (), {"__init__": lambda self, *a, **k: None})
        tv.transforms = tvt
        sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    import torch
    nthreads = torch.get_num_threads()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from data.dsec.subsequence.base import BaseSubSequence
    from data.dsec.subsequence import twostep as twostep_module
    from data.dsec.eventslicer import EventSlicer
    from data.utils.representations import VoxelGrid, norm_voxel_grid
    from data.utils.keys import DataLoading, DataSetType
    torch.set_num_threads(nthreads)
    ns = types.SimpleNamespace(BaseSubSequence=BaseSubSequence, twostep_module=twostep_module, TwoStepSubSequence=twostep_module.TwoStepSubSequence,
                               EventSlicer=EventSlicer, VoxelGrid=VoxelGrid, norm_voxel_grid=norm_voxel_grid, DataLoading=DataLoading,
                               DataSetType=DataSetType)
    return ns


class ReferenceTwoStepDriver:
    """Runs the reference's TwoStepSubSequence.__getitem__ on an in-memory event stream: a bare instance (no __init__, which
    wants a DSEC directory) whose file readers are replaced by providers over synthetic arrays.  Everything between the readers
    and the returned sample -- window arithmetic, event slicing offsets, rectification, voxel grid, merge, normalisation --
    is reference code."""

    def __init__(self, ns, events, rectify_map, forward_flow_timestamps, num_bins, H, W, normalize=True, merge=True):
        import numpy as np
        import torch
        self.ns = ns
        seq = ns.TwoStepSubSequence.__new__(ns.TwoStepSubSequence)
        seq.height, seq.width, seq.num_bins = H, W, num_bins
        seq.voxel_grid = ns.VoxelGrid(num_bins, H, W)
        seq.normalize_voxel_grid = ns.norm_voxel_grid if normalize else None
        seq.merge_grids = merge
        seq.augmentor = None
        seq.rectify_events_map = rectify_map
        seq.version = 1
        seq.load_voxel_grid = False
        seq.img_dir_ev_left = None
        seq.forward_flow_timestamps = forward_flow_timestamps
        seq.forward_flow_list = [types.SimpleNamespace(stem=f"{2 * (i + 1):06d}") for i in range(len(forward_flow_timestamps))]
        seq.h5f_opened = True                                                    # skip __open_h5f
        t_all = events["t"]

        class Slicer:                                                            # EventSlicer minus h5 I/O: same offset arithmetic
            def get_start_time_us(self_inner):
                return int(t_all[0])

            def get_final_time_us(self_inner):
                return int(t_all[-1])

            def get_events(self_inner, t0, t1):
                i0, i1 = ns.EventSlicer.get_time_indices_offsets(t_all, t0, t1)
                return {k: events[k][i0:i1] for k in ("p", "x", "y", "t")}

        seq.event_slicer = Slicer()
        self.seq = seq
        H_, W_ = H, W
        ns.twostep_module.load_flow = lambda path: (np.zeros((H_, W_, 2), dtype="float32"), np.ones((H_, W_), dtype=bool))

    def sample(self, index):
        out = self.seq[index]
        return out[self.ns.DataLoading.EV_REPR]
