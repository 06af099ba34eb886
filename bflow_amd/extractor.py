"""Feature / context encoder (K4) -- the drop-in for models/raft_utils/extractor.py:5-125.

Same module tree and parameter names as the reference (so its checkpoints load), inference only.  The dense
convolutions are deliberately left to MIOpen through PyTorch-ROCm (not one of the
hand-written kernels of the north star); what runs between them is arranged for few launches.
"""
from __future__ import annotations

from typing import List, Sequence, Union

import torch
import torch.nn as nn
import torch.nn.functional as F


def _make_norm(kind: str, channels: int) -> nn.Module:
    if kind == "group":
        return nn.GroupNorm(num_groups=channels // 8, num_channels=channels)
    if kind == "batch":
        return nn.BatchNorm2d(channels)
    if kind == "instance":
        return nn.InstanceNorm2d(channels)
    if kind == "none":
        return nn.Sequential()
    raise NotImplementedError(kind)


class ResidualBlock(nn.Module):
    """extractor.py:5-55: two 3x3 convs (+norm+relu), 1x1 strided shortcut when stride != 1."""

    def __init__(self, in_planes: int, planes: int, norm_fn: str = "group", stride: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.norm1 = _make_norm(norm_fn, planes)
        self.norm2 = _make_norm(norm_fn, planes)
        self.downsample = None
        if stride != 1:
            # registered under both names, exactly like the reference (norm3 and downsample.1 share parameters)
            self.norm3 = _make_norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = F.relu_(self.norm1(self.conv1(x)))
        y = F.relu_(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu_(x + y)


class BasicEncoder(nn.Module):
    """extractor.py:58-125: 7x7/2 stem, three stages of two residual blocks (64, 96/2, 128/2), 1x1 projection."""

    def __init__(self, input_dim: int = 3, output_dim: int = 128, norm_fn: str = "batch"):
        super().__init__()
        self.norm_fn = norm_fn
        if norm_fn == "group":
            self.norm1 = nn.GroupNorm(num_groups=8, num_channels=64)
        else:
            self.norm1 = _make_norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(input_dim, 64, kernel_size=7, stride=2, padding=3)
        planes = [(64, 1), (96, 2), (128, 2)]
        cin = 64
        for idx, (dim, stride) in enumerate(planes, start=1):
            setattr(self, f"layer{idx}", nn.Sequential(ResidualBlock(cin, dim, norm_fn, stride=stride),
                                                       ResidualBlock(dim, dim, norm_fn, stride=1)))
            cin = dim
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)
        for m in self.modules():   # extractor.py:85-92
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x: Union[torch.Tensor, Sequence[torch.Tensor]], project: bool = True):
        """A list input is stacked along the batch axis and split again (extractor.py:106-110,122-123).
        project=False returns the 128-channel trunk output (the caller applies conv2 itself)."""
        as_list = isinstance(x, (list, tuple))
        if as_list:
            nb, count = x[0].shape[0], len(x)
            x = torch.cat(list(x), dim=0)
        x = F.relu_(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        if project:
            x = self.conv2(x)
        if as_list:
            return list(torch.split(x, [nb] * count, dim=0))
        return x
