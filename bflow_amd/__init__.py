"""bflow_amd -- MI355X-native (gfx950 / CDNA4) RAFT-spline inference hot path of uzh-rpg/bflow.

Host side: Python on PyTorch-ROCm mirroring the reference's operator interface; device side: hand-written HIP kernels
behind the C ABI of include/bflow_hip.h (bflow_amd/lib/libbflow_hip.so), loaded with ctypes (bflow_amd/hip.py)."""
from .bezier import BezierCurves  # noqa: F401
from .corr import CorrBlockParallelMultiTarget, CorrComputation  # noqa: F401
from .raft_spline import RAFTSpline  # noqa: F401

__all__ = ["RAFTSpline", "BezierCurves", "CorrComputation", "CorrBlockParallelMultiTarget"]
