"""N>1 control path on CPU: world_size-2 gloo processes run the seed-sharding / barrier / max-over-ranks / gather logic
that bench.py and a 10-seed sweep use on the GPUs (there the backend is nccl = RCCL)."""
import os
import socket

import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist            # noqa: E402
import torch.multiprocessing as mp          # noqa: E402

from subreg_hip import sweep                # noqa: E402


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = sweep.assign_seeds(range(1, 11), world)[rank]
    sweep.barrier()
    t = sweep.max_over_ranks(1.0 + rank)                      # rank 1 is the slow one
    res = sweep.gather_results({"rank": rank, "seeds": seeds, "episodes": 8 * len(seeds)})
    if rank == 0:
        q.put((t, res))
    dist.destroy_process_group()


def test_two_rank_seed_sharding_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t, res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0                                           # max over ranks
    assert sorted(sum((r["seeds"] for r in res), [])) == list(range(1, 11))   # every seed exactly once
    assert sum(r["episodes"] for r in res) == 80


def test_seed_assignment_and_makespan():
    a = sweep.assign_seeds(range(1, 11), 8)
    assert [len(x) for x in a] == [2, 2, 1, 1, 1, 1, 1, 1]
    assert sweep.makespan_units(10, 8) == 2 and sweep.makespan_units(10, 1) == 10 and sweep.makespan_units(8, 8) == 1
    assert sweep.max_over_ranks(3.5) == 3.5 and sweep.gather_results("x") == ["x"]   # single-process degenerate forms
