#!/usr/bin/env python3
"""Headline benchmark: incremental episodes/sec, ResNet18, miniImageNet-shaped 5-way 5-shot FSCIL.

A STEP is one incremental episode (= one session of eval/language_eval.py:145-395): E fine-tune epochs, each
recomputing the frozen backbone on the support set and on every query set so far (exactly the forwards the
reference makes - no feature caching), the fused classifier/regularizer step, per-epoch validation, and the
1000-image base evaluation.  Workload = BASELINE.json configs[1]: 8-session FSCIL with the subspace regularizer,
-M (no replay), bf16, one MI355X; step i is session (i mod 8) of a run, and a new run (with its initial base
evaluation) starts every 8 steps.  Because the reference's stop rule is data-dependent, E is FIXED
(--epochs, default 100 = SURVEY.md section 8d headline); the stop rule still runs on the device.
With N GPUs every rank runs its own seed (the reference shards seeds over SLURM array tasks,
scripts/continual/slurm_subspace_reg.sh:8,19-27): weak scaling, no data-path collective.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family (conv_fwd_kernel, implicit-GEMM MFMA):
achieved = algorithmic conv FLOPs (8.1219 GFLOP/image x images forwarded in the timed region) / the summed
HIP-event duration of those backbone forwards on the launch stream.  `cpu_baseline` times the NumPy oracle
(a port, not the reference) on a bounded sample on this host.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "subspace-reg_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np   # noqa: E402
import torch         # noqa: E402

FLOP_PER_IMAGE = 8.1219e9          # 22 convs, 2*MAC, 84x84
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # MI355X_MICROARCH.md: dense MFMA peaks


class _Loader(list):
    def __init__(self, items, label2human):
        super().__init__(items)
        self.dataset = SimpleNamespace(label2human=label2human)


def make_opt(args, seed):
    return SimpleNamespace(
        no_dropblock=True, linear_bias=False, dataset="miniImageNet", set_seed=seed, memory_replay=0, neval_episodes=8,
        continual=False, n_ways=5, n_shots=5, n_queries=25, label_pull=1.0, pulling="regularize",
        attraction_override="distance2subspace", classifier="linear", lmbd_reg_transform_w=0.2, lmbd_reg_novel=0.1,
        target_train_loss=0.0, convergence_epsilon=1e-4, stable_epochs=args.epochs + 1, max_novel_epochs=args.epochs,
        min_novel_epochs=20, learning_rate=0.002, momentum=0.9, weight_decay=5e-4, adam=False, freeze_backbone_at=1,
        hip_dtype=args.dtype)


def make_run_inputs(seed, dev, n_base):
    """Synthetic episodes already resident in HBM (randn images, the reference's label layout)."""
    from subreg_hip import synthetic as syn
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + seed)
    items = []
    for s in range(8):
        sy, qy = syn.session_labels(s)
        sx = torch.randn(1, 125, 3, 84, 84, device=dev, generator=g)
        qx = torch.randn(1, 125, 3, 84, 84, device=dev, generator=g)
        items.append((sx, torch.from_numpy(sy)[None], qx, torch.from_numpy(qy)[None]))
    meta = _Loader(items, ["n%d" % i for i in range(100)])
    bx = torch.randn(n_base, 3, 84, 84, device=dev, generator=g)
    by = torch.randint(0, 60, (n_base,), device=dev, generator=g)
    base = _Loader([(bx, by, torch.arange(n_base))], ["b%d" % i for i in range(60)] + [""] * 40)
    return meta, base


def make_net(args, seed, dev):
    """Seeded random-init backbone (kaiming-normal convs) with BN running stats warmed by train-mode passes."""
    from subreg_hip import synthetic as syn
    from subreg_hip.resnet_language import create_model
    opt = make_opt(args, seed)
    net = create_model("resnet18", 60, opt)
    sd = syn.make_state_dict(seed, randomize_bn=False)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net = net.to(dev)
    for p in net.parameters():
        p.requires_grad = False
    net.classifier.weight.requires_grad = True
    g = torch.Generator(device=dev)
    g.manual_seed(7 + seed)
    net.train()
    with torch.no_grad():
        for _ in range(10):
            net.features(torch.randn(64, 3, 84, 84, device=dev, generator=g))
    net.eval()
    net.hip_backbone().nbt = [0] * 6
    return net, opt


def images_per_episode(s, epochs, n_base):
    """-M: support 125 + (s+1) query sets of 125 per epoch, plus the base evaluation."""
    return epochs * (125 + 125 * (s + 1)) + n_base


def cpu_baseline(args, n_base):
    """NumPy oracle (port) eval-mode forward on a bounded sample, converted to episodes/s of the same workload."""
    from oracle.resnet_ref import ResNetRef
    from subreg_hip import synthetic as syn
    net = ResNetRef(syn.make_state_dict(1))
    net.features(syn.make_images(0, 2, 84))                  # page in BLAS
    n, t = 4, 0.0
    while True:
        x = syn.make_images(1, n, 84)
        t0 = time.time()
        net.features(x)
        t = time.time() - t0
        if t >= args.cpu_seconds / 2 or n >= 512:
            break
        n = min(512, max(n * 2, int(n * args.cpu_seconds / max(t, 1e-3) * 0.8)))
    img_s = n / t
    avg_imgs = np.mean([images_per_episode(s, args.epochs, n_base) for s in range(8)]) + n_base / 8.0   # + run-start base eval
    try:                                                      # threads the oracle's BLAS calls actually ran on
        from threadpoolctl import threadpool_info
        threads = max([int(p.get("num_threads", 1)) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        threads = os.cpu_count()
    return {"value": img_s / avg_imgs, "unit": "episodes/s", "cores": threads, "kind": "port",
            "sample": "NumPy oracle eval-mode forward of %d 84x84 images in %.1f s (%.1f img/s), scaled by the mean "
                      "%.0f image-forwards per episode" % (n, t, img_s, avg_imgs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--epochs", type=int, default=100, help="fixed fine-tune epochs per episode (E)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--base-batch", type=int, default=1000)
    ap.add_argument("--epochs-per-sync", type=int, default=0, help="0 = queue the whole episode")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reuse-features", action="store_true",
                    help="NOT the headline: opt-in frozen-feature reuse (reported in DESIGN.md only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)       # RCCL; used only for the barrier and the max-over-ranks time

    from subreg_hip import sweep
    from subreg_hip.incremental import IncrementalRunner
    seed = rank + 1                                           # one seed per GPU, like the SLURM array
    net, opt = make_net(args, seed, dev)
    meta, base = make_run_inputs(seed, dev, args.base_batch)
    eps = args.epochs_per_sync or args.epochs

    def new_runner(profile):
        # fresh run state: base classifier rows, BN stats keep evolving across runs (synthetic, values do not matter)
        with torch.no_grad():
            net.classifier.weight = torch.nn.Parameter(net.classifier.weight.detach()[:60].clone())
        return IncrementalRunner(net, meta, base, opt, None, None, None, eps, args.reuse_features, verbose=False,
                                 profile=profile).start()

    # ---- warm-up: W untimed episodes (kernel load, func attributes, workspace allocation)
    r = new_runner(False)
    for i in range(args.warmup):
        if i and i % 8 == 0:
            r = new_runner(False)
        r.run_session(i % 8)
    # ---- timed: exactly K episodes
    torch.cuda.synchronize()
    sweep.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runners = []
    for i in range(args.steps):
        if i % 8 == 0:
            r = new_runner(True)
            runners.append(r)
        r.run_session(i % 8)
    torch.cuda.synchronize()
    sweep.barrier()
    torch.cuda.synchronize()
    dt = sweep.max_over_ranks(time.perf_counter() - t0, dev)

    imgs = sum(rr.images_forwarded for rr in runners)
    fwd_ms = sum(e0.elapsed_time(e1) for rr in runners for (e0, e1, _n) in rr.fwd_events)
    n_fwd = sum(len(rr.fwd_events) for rr in runners)
    achieved = imgs * FLOP_PER_IMAGE / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = (json.load(open(tpath)).get(args.dtype) or {}).get("bytes_per_image")   # HBM bytes per image forwarded
    out = {
        "metric": "incremental episodes/sec, ResNet18 miniImageNet 5w5s", "value": args.steps * world / dt,
        "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "8-session FSCIL, subspace regularizer, -M (BASELINE.json configs[1]); one seed per GPU",
                   "epochs_per_episode": args.epochs, "images_per_gpu": imgs, "base_batch": args.base_batch,
                   "feature_reuse": bool(args.reuse_features), "backbone": "ResNet18 (RFS ResNet-12 family, 8.1219 GFLOP/img)"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                     "frac": achieved / PEAK_TFLOPS[args.dtype], "traffic": traffic,
                     "kernel": "conv_fwd_kernel family over %d backbone forwards (%.1f ms each)" % (n_fwd, fwd_ms / max(n_fwd, 1))},
        "images_per_s": imgs * world / dt,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.base_batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
