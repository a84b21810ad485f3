"""bench.py's multi-GPU contract (replicas only: barrier + max-over-ranks + aggregate) on CPU with gloo, world_size 2."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from quick_amd import replicas
    dist = replicas.init("gloo")
    assert dist is not None and replicas.world() == (rank, rank, 2)
    replicas.barrier(dist)
    ms = replicas.max_over_ranks(dist, 1.0 + rank)            # rank 1 is the slow one
    out.put((rank, ms, replicas.job_throughput(10.0, ms, 2)))
    replicas.barrier(dist)
    dist.destroy_process_group()


def test_two_replicas_report_the_slowest_rank_and_the_aggregate_rate():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [2.0, 2.0]                   # both see the max
    assert all(abs(r[2] - 10.0 * 2 / 2e-3) < 1e-6 for r in res)


def test_single_process_is_a_no_op():
    sys.path.insert(0, ROOT)
    from quick_amd import replicas
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert replicas.init("gloo") is None and replicas.world() == (0, 0, 1)
    assert replicas.max_over_ranks(None, 3.5) == 3.5
