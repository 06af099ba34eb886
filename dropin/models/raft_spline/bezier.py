"""Drop-in for models/raft_spline/bezier.py (imported by callbacks/logger.py:20)."""
from bflow_amd.bezier import BezierCurves  # noqa: F401
