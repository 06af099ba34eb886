"""Drop-in for the reference's models/raft_spline/raft.py: put `<repo>/dropin` (and `<repo>`) BEFORE the reference on
PYTHONPATH and `from models.raft_spline.raft import RAFTSpline, BezierCurves` (modules/raft_spline.py:9) resolves here."""
from bflow_amd.bezier import BezierCurves  # noqa: F401
from bflow_amd.raft_spline import RAFTSpline  # noqa: F401
