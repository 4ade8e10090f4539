"""Seed-sharded sweep across the GPUs of one node (one process per GPU, torch.distributed).

The reference scales the incremental path only by SLURM array tasks - one seed (with its own backbone) per GPU, no
communication (scripts/continual/slurm_subspace_reg.sh:8,19-31).  This module is that launcher's in-process
counterpart: every rank takes the seeds `seeds[rank::world]`, runs them independently, and the only collectives are
the barrier / max-over-ranks time of the benchmark contract and the gather of the per-seed result lists.
Backend: "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def assign_seeds(seeds, world):
    """Equal-cost seeds dealt round-robin: rank r runs seeds[r::world] (10 seeds on 8 GPUs -> two ranks run 2)."""
    seeds = list(seeds)
    return [seeds[r::world] for r in range(world)]


def makespan_units(n_seeds, world):
    """Sequential seed-runs on the busiest rank (speed-up over 1 GPU = n_seeds / makespan_units)."""
    return max(len(s) for s in assign_seeds(range(n_seeds), world))


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds, device=None):
    """The benchmark contract's max-over-ranks wall time."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(obj):
    """Per-rank python objects (per-seed accuracy lists) -> list over ranks on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Level 2: intra-seed data parallelism for the sweep's tail (10 seeds on 8 GPUs leave 2 seeds for a second round; without
# it the makespan is 2 seed-runs = 5x over one GPU).  The backbone is frozen, eval-mode images are independent, so a group
# of ranks holding the same backbone each forward a contiguous slice of every epoch's image batch and exchange the
# [rows, 640] fp32 features with ONE all-gather (1.8 MB per epoch at 700 images); the classifier step and the validation
# run redundantly on every rank of the group, so no further communication is needed and all ranks take identical stop
# decisions.  The only other collective is the one-time broadcast of the seed's backbone to its group.
class RowShard:
    """Contiguous row slices of a batch over the ranks of a process group + the feature all-gather."""

    def __init__(self, group=None):
        self.group = group
        self.size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.host_staged = dist.is_initialized() and dist.get_backend(group) == "gloo"   # CPU tests / single-GPU checks
        self.leader = dist.get_global_rank(group, 0) if (dist.is_initialized() and group is not None) else 0   # global rank of group rank 0

    def rows(self, n):
        """(lo, hi, per): this rank forwards rows [lo, hi) of n; every rank's slot in the gathered buffer has `per` rows."""
        per = (n + self.size - 1) // self.size
        lo = min(self.rank * per, n)
        return lo, min(lo + per, n), per

    def gather(self, local, n, out=None):
        """local [per, D] (rows beyond this rank's slice are padding) -> [n, D] on every rank.  `out`, if given, must have
        size*per rows; the result is its first n rows (no copy)."""
        per, d = local.shape
        if out is None:
            out = torch.empty(self.size * per, d, dtype=local.dtype, device=local.device)
        if self.size == 1:
            out[:per].copy_(local)
        elif self.host_staged:
            tmp = torch.empty(self.size * per, d, dtype=local.dtype)
            dist.all_gather_into_tensor(tmp, local.cpu().contiguous(), group=self.group)
            out.copy_(tmp)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)      # RCCL over xGMI
        return out[:n]


def broadcast_module(module, src, group=None, force=False):
    """The sweep's one data-path collective besides the feature gather: the seed's backbone + classifier from the group
    leader (global rank `src`) to the ranks that will help with it.  ONE collective per dtype over a flat buffer (the 133
    state_dict tensors of a ResNet18: 105 MB of fp32 in one RCCL broadcast - 0.7 ms on one xGMI link - and one int64 buffer of
    the 22 num_batches_tracked counters), not one launch per tensor.  Returns the number of collectives issued.
    force: issue the collectives on a one-rank group too (tools/rccl_smoke.py: the code path on the only GPU a box has)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return 0
    staged = dist.get_backend(group) == "gloo"
    by_dtype = {}
    for _name, t in sorted(module.state_dict().items()):
        by_dtype.setdefault(t.dtype, []).append(t)
    n = 0
    for dtype, tensors in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
        dev = tensors[0].device
        flat = torch.cat([t.detach().reshape(-1) for t in tensors])              # leader: its values; the others: overwritten
        if staged and flat.is_cuda:
            flat = flat.cpu()                                                    # gloo: host-staged (CPU tests, one-GPU checks)
        dist.broadcast(flat, src, group=group)
        n += 1
        flat = flat.to(dev)
        off = 0
        with torch.no_grad():
            for t in tensors:
                k = t.numel()
                t.copy_(flat[off:off + k].view_as(t))                           # in place: parameters and buffers keep their identity
                off += k
    return n


def plan_sweep(seeds, world):
    """Rounds of (seed, ranks): full rounds give every rank its own seed; the last, partial round splits all ranks evenly
    over the seeds that are left (10 seeds, 8 ranks -> 8 x 1 rank, then 2 seeds x 4 ranks)."""
    seeds = list(seeds)
    rounds = []
    while len(seeds) >= world:
        rounds.append([(seeds[r], [r]) for r in range(world)])
        seeds = seeds[world:]
    if seeds:
        k = len(seeds)
        base, extra = divmod(world, k)
        groups, start = [], 0
        for i in range(k):
            n = base + (1 if i < extra else 0)
            groups.append((seeds[i], list(range(start, start + n))))
            start += n
        rounds.append(groups)
    return rounds


def sweep_speedup(n_seeds, world, dp_efficiency=0.9):
    """Speed-up over one GPU of `plan_sweep`: a seed shared by g ranks takes 1 / (1 + (g - 1) * dp_efficiency) seed-times
    (the backbone forward is >= 94 % of a run and shards perfectly; the redundant classifier step and the gather do not)."""
    t = 0.0
    for rnd in plan_sweep(range(n_seeds), world):
        t += max(1.0 / (1.0 + (len(ranks) - 1) * dp_efficiency) for _seed, ranks in rnd)
    return n_seeds / t


def sweep_speedup_measured(n_seeds, world, seed_ms_by_group_size):
    """Speed-up over one GPU of `plan_sweep` from MEASURED seed-run times: seed_ms_by_group_size[g] = time of one seed's 8-session
    run when g ranks share it (bench.py::sweep_model builds it from forwards timed at the per-rank batch sizes)."""
    t = 0.0
    for rnd in plan_sweep(range(n_seeds), world):
        t += max(float(seed_ms_by_group_size[len(ranks)]) for _seed, ranks in rnd)
    return n_seeds * float(seed_ms_by_group_size[1]) / t
