"""Drop-in for models/raft_utils/utils.py:5-48."""
import torch

from bflow_amd import hip


def coords_grid(batch, ht, wd, device):
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


def cvx_upsample(data: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    return hip.cvx_upsample(data.contiguous().float(), mask.contiguous().float())


def bilinear_sampler(img, coords):
    """Generic pixel-coordinate grid_sample wrapper (utils.py:5-21).  The hot path does not use it: the correlation look-up
    has its own gather kernel (bflow_corr_lookup)."""
    H, W = img.shape[-2:]
    xg, yg = coords.split([1, 1], dim=-1)
    grid = torch.cat([2 * xg / (W - 1) - 1, 2 * yg / (H - 1) - 1], dim=-1)
    return torch.nn.functional.grid_sample(img, grid, align_corners=True)
