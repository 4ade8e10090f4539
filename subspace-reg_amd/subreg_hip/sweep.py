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
