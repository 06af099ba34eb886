"""Drop-in for models/raft_utils/extractor.py."""
from bflow_amd.extractor import BasicEncoder, ResidualBlock  # noqa: F401
