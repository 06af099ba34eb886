"""Batch sharding over the GPUs of one node (one process per GPU) and the single exchange step of the path.

Samples are independent (InstanceNorm in fnet, eval-mode BatchNorm in cnet, per-sample correlation/GRU: SURVEY.md
section 8e), so a global batch is cut into contiguous shards with NO data-path collective.  The only communication is the
metric state sync that torchmetrics performs for the reference (utils/metrics.py:34-35: two scalars, dist_reduce_fx="sum"):
one all-gather of the (epe_sum, count) record per rank -- RCCL over xGMI on the GPU box ("nccl" backend), gloo in CPU
tests.  16 B per rank: latency-bound, issued once per evaluation, outside any timed frame loop."""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the default process group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [start, stop) of a global batch; the first (global_batch % world) ranks get one extra sample."""
    base, extra = divmod(global_batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_records(record: torch.Tensor) -> torch.Tensor:
    """record: (k,) float64 on this rank -> (world, k).  One all-gather (RCCL ncclAllGather on GPU tensors)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return record.unsqueeze(0)
    out = [torch.empty_like(record) for _ in range(dist.get_world_size())]
    dist.all_gather(out, record.contiguous())
    return torch.stack(out, dim=0)


def reduce_epe(epe_sum: torch.Tensor, count: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Sum the per-rank metric states like torchmetrics' dist_reduce_fx='sum' and return (mean, epe_sum, count)."""
    rec = torch.stack([epe_sum.double().reshape(()), count.double().reshape(())])
    allrec = all_gather_records(rec)
    tot = allrec.sum(dim=0)
    return tot[0] / tot[1], tot[0], tot[1]


def reduce_metric_states(state: torch.Tensor) -> torch.Tensor:
    """Sum an (M, 2) [value sum, batch count] metric-state table over the ranks (dist_reduce_fx="sum" for every metric of
    utils/metrics.py) with ONE all-gather; returns the summed table.  Epoch value of row m = table[m, 0] / table[m, 1]."""
    flat = state.double().reshape(-1)
    return all_gather_records(flat).sum(dim=0).reshape(state.shape)


def evaluate_sharded(forward_flow: Callable[[int, int], torch.Tensor], gt_flow: Callable[[int, int], torch.Tensor],
                     epe_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], global_batch: int,
                     micro_batch: int, rank: int, world: int, device=None):
    """Runs this rank's shard in micro-batches and reduces EPE over ranks.
      forward_flow(first_sample, n) -> predicted flow (n, 2, H, W) for global samples [first, first+n)
      gt_flow(first_sample, n)      -> ground truth of the same samples
      epe_fn(pred, gt)              -> mean EPE of the micro-batch (a 0-dim tensor)
    The metric state follows the reference: sum of per-(micro-)batch means and their count (metrics.py:42-49)."""
    start, stop = shard_range(global_batch, rank, world)
    epe_sum = torch.zeros((), dtype=torch.float64, device=device)
    count = torch.zeros((), dtype=torch.float64, device=device)
    s = start
    while s < stop:
        n = min(micro_batch, stop - s)
        e = epe_fn(forward_flow(s, n), gt_flow(s, n))
        epe_sum = epe_sum + e.double().to(epe_sum.device)
        count = count + 1
        s += n
    return reduce_epe(epe_sum, count)
