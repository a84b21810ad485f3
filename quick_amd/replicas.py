"""Multi-GPU policy of this path: REPLICAS ONLY.

One dense per-layer GEMM has no independent units to shard and the reference performs no exchange step;
``bench.py --gpus N`` therefore runs N independent copies, one process per GPU, and only the *measurement* is collective:
a barrier around the timed region and the max over ranks of the step time.  These helpers hold that logic so that it can be
exercised without GPUs (gloo, world_size 2, tests/test_replicas_cpu.py).
"""
import os


def world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend="gloo"):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them).  gloo on
    CPU tensors: the path has no exchange step, so no RCCL communicator is ever created (north star: "no RCCL")."""
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return None
    dist.init_process_group(backend)
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value, device="cpu"):
    """The slowest rank defines the step time of the job."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_rank_per_step, ms_per_step_max, n_ranks):
    """Whole-job rate of N replicas: every rank processed the same units in (at most) the slowest rank's time."""
    return units_per_rank_per_step * n_ranks / (ms_per_step_max * 1e-3)
