#!/usr/bin/env python3
"""The 10-seed x 8-session sweep of scripts/continual/slurm_subspace_reg.sh on the GPUs of one node (synthetic data).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_sweep.py --seeds 10
Round plan = subreg_hip.sweep.plan_sweep: every rank runs its own seed while there are at least `world` seeds left; the
remaining seeds are then shared by groups of ranks (RCCL broadcast of the seed's backbone to its group, each rank forwards
a row slice of every epoch's batch, one all-gather of the features per forward).  Rank 0 prints one JSON line.

Robustness (the reference's SLURM array re-queues a failed task): with --out DIR every finished seed's result
is written to DIR/seed_<n>.json by the leader of its group as soon as the seed ends (atomic rename), and a re-run with the same
DIR skips the seeds already there - a crashed sweep loses the seeds in flight, not the finished ones.  --checkpoint PATTERN (e.g.
'ckpt/seed{seed}/resnet18_last.pth') loads each seed's real backbone on the group leader (subreg_hip.checkpoint) instead of the
synthetic one; the group broadcast hands it to the helpers."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd")]
import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

import bench                                    # noqa: E402  (make_net / make_run_inputs: the bench's synthetic workload)
from subreg_hip import sweep                    # noqa: E402
from subreg_hip.incremental import IncrementalRunner   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=10)
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--base-batch", type=int, default=1000)
    ap.add_argument("--out", default="", help="directory of per-seed result files (seed_<n>.json); finished seeds are skipped on a re-run")
    ap.add_argument("--checkpoint", default="", help="path pattern with {seed} of each seed's pretrained backbone (reference layout)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    seeds = list(range(1, args.seeds + 1))
    done_before = []
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        done_before = [sd for sd in seeds if os.path.exists(os.path.join(args.out, "seed_%d.json" % sd))]   # same answer on every rank (shared fs)
        seeds = [sd for sd in seeds if sd not in done_before]
    plan = sweep.plan_sweep(seeds, world)
    # every rank creates every group, in the same order (torch.distributed.new_group is collective over the world)
    groups = {tuple(ranks): (dist.new_group(ranks) if world > 1 and len(ranks) > 1 else None)
              for rnd in plan for _seed, ranks in rnd}
    results = []
    torch.cuda.synchronize()
    sweep.barrier()
    t0 = time.perf_counter()
    for rnd in plan:
        mine = [(seed, ranks) for seed, ranks in rnd if rank in ranks]
        for seed, ranks in mine:
            group = groups[tuple(ranks)]
            net, opt = bench.make_net(args, seed, dev)              # same seed -> same synthetic backbone on every rank ...
            if args.checkpoint and rank == ranks[0]:                # ... or the seed's real one, read by the group leader only
                from subreg_hip import checkpoint as ck
                net.load_state_dict(ck.load_checkpoint(args.checkpoint.format(seed=seed))["model"])   # the reference's {'opt', 'model', ...} dict
                net.hip_backbone().refresh(force=True)
            if group is not None:
                sweep.broadcast_module(net, ranks[0], group)        # ... a real sweep loads it on the leader only
            meta, base = bench.make_run_inputs(seed, dev, args.base_batch)
            shard = sweep.RowShard(group) if group is not None else None
            r = IncrementalRunner(net, meta, base, opt, None, None, None, args.epochs, False, verbose=False,
                                  row_shard=shard).start()
            for idx in range(r.iter_num):
                r.run_session(idx)
            novel_avg, base_avg = r.finish()
            if rank == ranks[0]:
                rec = {"seed": seed, "ranks": ranks, "novel_avg": novel_avg, "base_avg": base_avg,
                       "weighted": net.last_run["weighted_avg"]}
                results.append(rec)
                if args.out:                                        # on disk the moment the seed is done
                    tmp = os.path.join(args.out, ".seed_%d.json.%d" % (seed, os.getpid()))
                    with open(tmp, "w") as f:
                        json.dump(rec, f)
                    os.replace(tmp, os.path.join(args.out, "seed_%d.json" % seed))
    torch.cuda.synchronize()
    sweep.barrier()
    dt = sweep.max_over_ranks(time.perf_counter() - t0, dev)
    allres = sweep.gather_results(results)
    if rank == 0:
        flat = [x for r in allres for x in r]
        for sd in done_before:                                      # seeds an earlier (interrupted) run finished
            with open(os.path.join(args.out, "seed_%d.json" % sd)) as f:
                flat.append(json.load(f))
        flat = sorted(flat, key=lambda x: x["seed"])
        print(json.dumps({"sweep": "%d seeds x 8 sessions" % args.seeds, "n_gpus": world, "seconds": dt,
                          "episodes_per_s": len(seeds) * 8 / dt, "seeds_skipped_already_done": done_before, "plan": [[(s, len(rk)) for s, rk in rnd] for rnd in plan],
                          "model_speedup_over_1_gpu": sweep.sweep_speedup(args.seeds, world), "results": flat}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
