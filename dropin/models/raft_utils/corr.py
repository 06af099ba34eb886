"""Drop-in for models/raft_utils/corr.py."""
from bflow_amd.corr import CorrBlockParallelMultiTarget, CorrComputation  # noqa: F401
