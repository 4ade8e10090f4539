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


def _grad_worker(rank, world, port, q):
    """Pretraining's data-parallel step on 2 ranks: the backbone's gradients are views of ONE flat buffer whose stage ranges
    are all-reduced asynchronously while the backward continues (train.BackboneTrainFn.backward -> GradientSync.stage_ready),
    the classifier's is a tensor of its own (GradientSync.finish).  SUM, not mean: the ranks pre-scale their losses."""
    from subreg_hip import pretrain as pt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    ps = [torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(5))]
    ps[0].grad, ps[1].grad = flat[0:6].view(2, 3), flat[6:10].view(4)
    ps[2].grad = torch.full((5,), float(rank))
    frozen = torch.nn.Parameter(torch.zeros(3))               # no gradient: skipped
    sync = pt.GradientSync()
    sync.stage_ready(flat[6:10])                              # "layer 4" first, in flight ...
    sync.stage_ready(flat[0:6])                               # ... then the earlier stage
    sync.finish(ps + [frozen])                                # the classifier, then wait for everything
    # a second step WITHOUT stage hooks: one collective per gradient storage
    flat2 = torch.ones(10) * (rank + 1)
    ps[0].grad, ps[1].grad, ps[2].grad = flat2[0:6].view(2, 3), flat2[6:10].view(4), torch.ones(5)
    calls1 = sync.calls
    sync.finish(ps)
    x, y = pt.shard_batch(torch.arange(7)[:, None], torch.arange(7), rank, world)
    if rank == 0:
        q.put((calls1, sync.calls - calls1, ps[0].grad.tolist(), flat.tolist(), ps[2].grad.tolist(), y.tolist(), pt.shard_sizes(7, 2)))
    dist.destroy_process_group()


def test_two_rank_gradient_sync_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    calls1, calls2, g0, flat, g2, y, sizes = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert calls1 == 3 and calls2 == 2                        # two stage ranges + the classifier; then flat buffer + classifier
    assert flat == [3.0 * i for i in range(10)]               # SUM of 1x and 2x arange, every element reduced exactly once
    assert g0 == [[3.0] * 3] * 2 and g2 == [2.0] * 5          # second step: 1 + 2, 1 + 1
    assert y == [0, 1, 2, 3] and sizes == [4, 3]              # balanced uneven shards: nobody gets an empty slice
    with pytest.raises(ValueError):
        from subreg_hip import pretrain as pt
        pt.shard_batch(torch.zeros(1, 1), torch.zeros(1), 0, 2)


def _shard_worker(rank, world, port, q):
    """Level-2 helpers: row slices + feature gather (uneven rows), backbone broadcast."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = sweep.RowShard()
    x = torch.arange(7 * 3, dtype=torch.float32).view(7, 3)
    fwd = lambda t: t @ torch.tensor([[1.0, 2.0], [0.5, -1.0], [3.0, 0.0]])          # noqa: E731  (stand-in "backbone")
    lo, hi, per = sh.rows(7)
    local = torch.zeros(per, 2)
    local[:hi - lo] = fwd(x[lo:hi])
    buf = torch.zeros(world * per, 2)
    full = sh.gather(local, 7, out=buf)
    # a module with fp32 parameters / buffers AND int64 counters (BatchNorm.num_batches_tracked): one flat collective per dtype
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    with torch.no_grad():
        lin[0].weight.fill_(float(rank + 1))
        lin[1].running_var.fill_(0.5 + rank)
        lin[1].num_batches_tracked.fill_(7 + rank)
    w_id = lin[0].weight.data_ptr()
    n_coll = sweep.broadcast_module(lin, 0)
    ok_b = (n_coll == 2 and lin[0].weight.data_ptr() == w_id and float(lin[1].running_var[0]) == 0.5
            and int(lin[1].num_batches_tracked) == 7 and lin[1].num_batches_tracked.dtype == torch.int64)
    if rank == 1:
        q.put(((lo, hi, per), torch.equal(full, fwd(x)), full.data_ptr() == buf.data_ptr(), float(lin[0].weight[0, 0]) if ok_b else -1.0))
    dist.destroy_process_group()


def test_two_rank_row_shard_and_broadcast_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rows, equal, in_place, w = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert rows == (4, 7, 4) and equal and in_place and w == 1.0


def test_sweep_plan_reaches_the_scaling_target():
    plan = sweep.plan_sweep(range(1, 11), 8)
    assert [len(r) for r in plan] == [8, 2]
    assert plan[0][3] == (4, [3]) and plan[1] == [(9, [0, 1, 2, 3]), (10, [4, 5, 6, 7])]
    assert sweep.plan_sweep(range(3), 8)[0] == [(0, [0, 1, 2]), (1, [3, 4, 5]), (2, [6, 7])]
    assert sweep.plan_sweep(range(16), 8) == [[(s, [s % 8]) for s in range(8)], [(s, [s % 8]) for s in range(8, 16)]]
    assert 10 / sweep.makespan_units(10, 8) == 5.0                   # seed sharding alone
    assert sweep.sweep_speedup(10, 8) > 6.0                          # + intra-seed data parallelism for the last two seeds
    assert abs(sweep.sweep_speedup(8, 8) - 8.0) < 1e-9 and abs(sweep.sweep_speedup(10, 1) - 1.0) < 1e-9


def test_bench_self_launch_and_sweep_plan_gloo():
    """`python bench.py --gpus 2` WITHOUT a launcher: the parent starts two ranks itself (torch.distributed.run child), the ranks
    rendezvous (gloo here, nccl on GPUs), walk sweep.plan_sweep incl. the shared-seed group of the last round (its group
    broadcast), and rank 0's single JSON line is relayed.  --selftest-host replaces the GPU work with nothing."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--selftest-host", "--sweep-seeds", "3"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["sweep"]["n_ranks_seen"] == 2
    assert out["sweep"]["seeds_done"] == [1, 2, 3]
    assert out["sweep"]["plan"] == [[[1, 1], [2, 1]], [[3, 2]]]          # two own-seed runs, then one seed shared by both ranks
    # a mismatching launcher is refused loudly
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--selftest-host"], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, env=env2, timeout=120)
    assert p2.returncode != 0 and "WORLD_SIZE" in p2.stderr


@pytest.mark.gpu
def test_rccl_single_rank_collectives_on_device():
    """Backend "nccl" (RCCL) with one rank on cuda:0, in a fresh process: every collective the build issues (barrier, fp64 MAX,
    state_dict broadcast incl. int64 counters, feature all-gather, sub-group, staged async gradient all-reduce behind the HIP
    backward) runs on the device and reproduces the identity exactly (tools/rccl_smoke.py).  Two-rank semantics: the gloo tests
    above and tools/dp_check.py / dp_pretrain_check.py."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(repo, "tools", "rccl_smoke.py")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "-> OK" in p.stdout and "backend nccl" in p.stdout


@pytest.mark.gpu
def test_row_sharded_sweep_path_two_ranks_on_device():
    """Level 2 of the 10-seed sweep (BASELINE.json configs[3]; scripts/continual/slurm_subspace_reg.sh:8,19-31 shards whole seeds,
    the build also shares the last seeds among idle GPUs): IncrementalRunner(row_shard=RowShard()) with TWO processes on the
    device - rank 1 starts from a perturbed backbone that the group broadcast must repair, every eval-mode forward is cut into
    two row slices (hipGraph-replayed), the [rows, 640] features are all-gathered, the train-mode forward / classifier step /
    validation run redundantly - and the result must equal the single-process run BIT FOR BIT (losses, accuracies, classifier
    rows; tools/dp_check.py).  The pool's boxes have one GPU, so both ranks share cuda:0 and the collectives go over gloo
    (host-staged); with one GPU per rank the same code gathers device to device over RCCL."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(repo, "tools", "dp_check.py")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "single-process run: True" in p.stdout
    done, total = [int(x) for x in p.stdout.split("rank 0:")[1].split(")")[0].replace("of", " ").split()]
    assert 0 < done < total                                    # rank 0 really forwarded only its share of the images


@pytest.mark.gpu
def test_row_sharded_seed_with_pre_freeze_epochs_two_ranks_on_device():
    """freeze_backbone_at = 3 on a row-sharded seed (eval/language_eval.py:242-295, eval/util.py:62-69): the two whole-network
    epochs run on both ranks, the leader's network is re-broadcast after each step (one flat buffer), the query forwards stay
    sharded.  Against the single-process run: same epochs and accuracies, losses and classifier rows within 1e-4 (the
    weight-gradient kernels' float atomics make that run itself not bit-reproducible)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(repo, "tools", "dp_check.py"), "hw32_freeze3"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "single-process run: True" in p.stdout


@pytest.mark.gpu
def test_pretrain_gradient_sync_two_ranks_on_device():
    """BASELINE.json configs[4], 'data-parallel RCCL allreduce': pretrain.GradientSync with TWO processes on the device
    (tools/dp_pretrain_check.py) - uneven shards 4 + 3 of a global batch, loss pre-scaled by n_local / n_global, the staged
    backward with one asynchronous SUM all-reduce per stage, fused SGD - must equal the single-process emulation of what
    nn.DataParallel does (train_supervised.py:141-142) up to the measured run-to-run noise of the float atomics.  One GPU per
    box: both ranks share cuda:0 over gloo; with one GPU per rank the same code all-reduces over RCCL."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(repo, "tools", "dp_pretrain_check.py")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "EQUIVALENT" in p.stdout and "MISMATCH" not in p.stdout


def test_sweep_speedup_from_measured_seed_times():
    """sweep.sweep_speedup_measured: the 10-seed plan's speed-up from measured seed-run times by group size (bench.py::sweep_model)."""
    ms = {1: 4000.0, 2: 2200.0, 4: 1400.0, 8: 900.0}
    assert sweep.sweep_speedup_measured(10, 1, ms) == pytest.approx(1.0)
    assert sweep.sweep_speedup_measured(10, 2, ms) == pytest.approx(2.0)                     # five full rounds of two seeds
    assert sweep.sweep_speedup_measured(10, 8, ms) == pytest.approx(10 * 4000.0 / (4000.0 + 1400.0))   # 8 x 1 rank, then 2 seeds x 4 ranks
    assert sweep.sweep_speedup_measured(10, 4, ms) == pytest.approx(10 * 4000.0 / (2 * 4000.0 + 2200.0))   # two full rounds, then 2 seeds x 2 ranks
    # with perfect sharding (time / g) the measured form equals the efficiency-1 model
    perfect = {g: 4000.0 / g for g in (1, 2, 4, 8)}
    for w in (2, 4, 8):
        assert sweep.sweep_speedup_measured(10, w, perfect) == pytest.approx(sweep.sweep_speedup(10, w, dp_efficiency=1.0))
