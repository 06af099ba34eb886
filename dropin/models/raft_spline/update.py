"""Drop-in for models/raft_spline/update.py."""
from bflow_amd.update import BasicMotionEncoder, BasicUpdateBlock, BezierHead, SepConvGRU  # noqa: F401
