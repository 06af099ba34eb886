"""Drop-in for the reference's utils/losses.py (imported at modules/raft_spline.py:13): the masked L1 sequence losses on the HIP
kernels, differentiable (forward bflow_l1_masked_accumulate, backward bflow_l1_masked_grad)."""
from bflow_amd.training import (l1_loss_channel_masked, l1_multi_seq_loss_channel_masked,  # noqa: F401
                                l1_seq_loss_channel_masked)
