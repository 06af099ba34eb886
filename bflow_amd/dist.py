"""Batch sharding over the GPUs of one node (one process per GPU) and the single exchange step of the path.

Samples are independent (InstanceNorm in fnet, eval-mode BatchNorm in cnet, per-sample correlation/GRU: SURVEY.md
section 8e), so a global batch is cut into contiguous shards with NO data-path collective.  The only communication is the
metric state sync that torchmetrics performs for the reference (utils/metrics.py:34-35: two scalars, dist_reduce_fx="sum"):
one all-gather of the (epe_sum, count) record per rank -- RCCL over xGMI on the GPU box ("nccl" backend), gloo in CPU
tests.  16 B per rank: latency-bound, issued once per evaluation, outside any timed frame loop."""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; initialises the default process group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("BFLOW_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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
    if record.is_cuda and dist.get_backend() == "gloo":      # gloo ranks on a GPU box (the one-GPU test of the N > 1 path): exchange on the host
        host = record.detach().cpu().contiguous()
        out = [torch.empty_like(host) for _ in range(dist.get_world_size())]
        dist.all_gather(out, host)
        return torch.stack(out, dim=0).to(record.device)
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


# ----------------------------------------------------------------------------------------------- training
def per_gpu_batch_size(batch_size: int, world: int) -> int:
    """train.py:41-52 splits config.training.batch_size over the GPUs; its own check is written the wrong way round
    (`batch_size * num_gpus == per_gpu_batch_size`, :52) and trips for every multi-GPU run -- this is the check it means."""
    assert batch_size > 0 and world > 0
    per_gpu = batch_size // world
    assert per_gpu * world == batch_size, f"Batch size ({batch_size}) must be divisible by number of gpus ({world})"
    return per_gpu


def broadcast_module_state(module: torch.nn.Module, src: int = 0):
    """Parameters and buffers of rank `src` to every rank (what DDP does at construction; buffers -- the BatchNorm running
    statistics of cnet -- again before each forward when `broadcast_buffers` is on, the reference's DDPStrategy default)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


class GradientBuckets:
    """Data-parallel gradient averaging, one process per GPU: parameters are packed into flat buckets in REVERSE registration
    order (the order their gradients become ready in backward); a bucket's all-reduce (RCCL ring over xGMI; gloo in the CPU
    test) is launched asynchronously from the hook of its last-arriving gradient, so it overlaps the rest of the backward pass;
    `finish()` waits, divides by the world size and scatters the buckets back into `.grad`.

    xGMI is point-to-point (per-link bound rings), so few LARGE buckets beat many small ones: the whole RAFT-spline model is
    5.3 M parameters = 21 MB fp32, i.e. one 32-MB bucket by default."""

    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 32 << 20):
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        params = [p for p in module.parameters() if p.requires_grad]
        self._buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self._buckets.append(cur)
        self._where = {id(p): b for b, ps in enumerate(self._buckets) for p in ps}
        self._pending = [0] * len(self._buckets)
        self._work: List[Optional[Tuple[object, torch.Tensor]]] = [None] * len(self._buckets)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params] if self.world > 1 else []
        self.reset()

    @property
    def num_buckets(self) -> int:
        return len(self._buckets)

    def reset(self):
        self._pending = [len(ps) for ps in self._buckets]
        self._work = [None] * len(self._buckets)

    def _launch(self, b: int):
        ps = self._buckets[b]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
        self._work[b] = (dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat)

    def _on_grad(self, p: torch.nn.Parameter):
        b = self._where[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def finish(self):
        """Call after loss.backward(): launches the buckets whose parameters got no gradient this step, waits for all of them and
        writes the rank-averaged gradients back."""
        if self.world == 1:
            return
        for b in range(len(self._buckets)):
            if self._work[b] is None:
                self._launch(b)
        for b, ps in enumerate(self._buckets):
            work, flat = self._work[b]
            work.wait()
            flat.div_(self.world)
            off = 0
            for p in ps:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
        self.reset()
