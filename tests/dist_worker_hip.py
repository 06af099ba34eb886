"""Rank worker of tests/test_multi_gpu.py: this rank's contiguous shard of a global batch on the HIP path (one process per GPU,
torch.distributed backend "nccl" = RCCL), EPE state all-gathered once.  Launched by torch.distributed.run; rank 0 prints one RESULT line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bflow_amd  # noqa: E402
from bflow_amd import configs, dist as bdist, synthetic  # noqa: E402
from bflow_amd.metrics import epe_masked  # noqa: E402
from bflow_amd.weights import deterministic_state_dict  # noqa: E402

G, H, W, ITERS, MICRO = 4, 128, 160, 3, 1


def main():
    rank, world, local = bdist.init_from_env("nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = configs.model_config("E_LU4_BD2")
    model = bflow_amd.RAFTSpline(cfg).eval()
    model.load_state_dict(deterministic_state_dict(model, seed=0))
    model.to(dev).enable_hipgraph()

    def fwd(first, n):
        vox = torch.from_numpy(synthetic.voxel_grid(n, 9, H, W, seed=1234, first_sample=first)).to(dev)
        _, up = model(voxel_grid=vox, iters=ITERS, test_mode=True)
        return up.get_flow_from_reference(1.0).clone()

    def gt(first, n):
        return torch.from_numpy(synthetic.gt_flow(n, H, W, seed=99, first_sample=first)).to(dev)

    mean, s, c = bdist.evaluate_sharded(fwd, gt, epe_masked, G, MICRO, rank, world, device=dev)
    if rank == 0:
        print("RESULT " + json.dumps(dict(mean=float(mean), sum=float(s), count=float(c), world=world, device=torch.cuda.get_device_name(local))), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
