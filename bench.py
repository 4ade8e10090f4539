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
scripts/continual/slurm_subspace_reg.sh:8,19-27): weak scaling, no data-path collective.  `python bench.py --gpus N`
without a launcher starts its own N ranks (a parent process that never touches the GPU runs
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child and relays rank 0's line); under
torchrun (RANK / WORLD_SIZE in the environment) it is a rank.

After the timed region (the headline is echoed to stderr first) the 10-seed x 8-session sweep of
scripts/continual/slurm_subspace_reg.sh:8,19-31 (BASELINE.json configs[3]; `--sweep-seeds`, 0 = skip) runs - in a child
process at one rank, on the same process group at N ranks - with subreg_hip.sweep.plan_sweep: one seed per rank while >= N seeds remain, then the remaining seeds shared by groups of
ranks (RCCL broadcast of the seed's backbone to its group, row-sliced forwards, one feature all-gather per forward).
Its episodes/s is reported under "sweep" in the same line (strong scaling: the work is fixed), so that
sweep.value at N GPUs / sweep.value at 1 GPU is the speed-up the north star's >= 6x target is about.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family (conv_fwd_kernel, implicit-GEMM MFMA):
achieved = algorithmic conv FLOPs (8.1219 GFLOP/image x images forwarded in the timed region) / the summed
HIP-event duration of those backbone forwards on the launch stream.  `cpu_baseline` times a torch-CPU restatement
(oracle/torch_ref.py: the library the reference computes with) of one fine-tune epoch on a bounded sample on this
host; the NumPy oracle's forward rate is kept beside it.
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


def make_net(args, seed, dev, sd=None):
    """Seeded random-init backbone (kaiming-normal convs) with BN running stats warmed by train-mode passes.  `sd`: the seed's
    state_dict if the caller already prepared it (the sweep builds the next seed's on a host thread while this seed runs)."""
    from subreg_hip import synthetic as syn
    from subreg_hip.resnet_language import create_model
    opt = make_opt(args, seed)
    net = create_model("resnet18", 60, opt)
    if sd is None:
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
    """One real fine-tune epoch of a -M session-1 episode (support forward, CE + three regularizers + SGD step, query
    validation: language_eval.py:252-326) on the torch-CPU restatement, repeated for about --cpu-seconds, converted to
    episodes/s of the benchmark's workload with the same images-per-episode count.  The NumPy oracle's forward rate is
    reported beside it (`port_numpy`)."""
    from oracle import torch_ref as tr
    from oracle.resnet_ref import ResNetRef
    from subreg_hip import synthetic as syn
    sd = syn.make_state_dict(1, randomize_bn=False)
    threads = torch.get_num_threads()
    net = tr.TorchCpuRef(sd)
    g = torch.Generator().manual_seed(5)
    n_img = 125 if args.cpu_seconds >= 8 else 25                # per set; the real episode has 125 support + 125 query
    sy, qy = syn.session_labels(0)
    sel = torch.arange(0, 125, 125 // n_img)[:n_img]
    sx, qx = torch.randn(n_img, 3, 84, 84, generator=g), torch.randn(n_img, 3, 84, 84, generator=g)
    lab_s = torch.from_numpy(sy)[sel] - int(sy.min()) + 60       # session-0 labels -> classifier rows 60..64
    lab_q = torch.from_numpy(qy)[sel] - int(qy.min()) + 60
    wb = torch.randn(60, 640, generator=g) * 0.05
    W = torch.cat([wb, torch.randn(5, 640, generator=g) * 0.03])
    hp = dict(lmbd_base=0.2, lmbd_prev=0.1, pull=1.0, lr=0.002, momentum=0.9, wd=5e-4)
    net.features(sx[:2])                                         # page in MKL-DNN
    # A FAIR baseline: the thread count is swept first (torch's default on the GPU box is every hardware thread, 128 - which ran the
    # 125-image batches SLOWER than the 8-core build container runs the reference itself: over-subscribed).  ~2 s per candidate on a
    # 32-image eval forward, the best one runs the timed epochs.
    max_threads = threads
    sweep = {}
    if args.cpu_seconds >= 8:
        xs = sx[:32]
        for nt in sorted({t_ for t_ in (8, 16, 32, 64, 128) if t_ <= max_threads} | {max_threads}):
            torch.set_num_threads(nt)
            with torch.no_grad():
                net.features(xs[:4])
                t0, n = time.time(), 0
                while time.time() - t0 < 1.5:
                    net.features(xs)
                    n += 1
            sweep[nt] = n * xs.shape[0] / (time.time() - t0)
        threads = max(sweep, key=sweep.get)
        torch.set_num_threads(threads)
    mom, epochs, t0 = None, 0, time.time()
    while True:
        _loss, _accs, mom = tr.finetune_epoch(net, W, mom, wb, None, sx, lab_s, [(qx, lab_q)], hp)
        epochs += 1
        t = time.time() - t0
        if t >= args.cpu_seconds * 0.6 or epochs >= 20:
            break
    torch.set_num_threads(max_threads)
    img_s = epochs * 2 * n_img / t
    avg_imgs = np.mean([images_per_episode(s, args.epochs, n_base) for s in range(8)]) + n_base / 8.0   # + run-start base eval
    out = {"value": img_s / avg_imgs, "unit": "episodes/s", "cores": threads, "kind": "port",
           "sample": "torch-CPU restatement (F.conv2d / batch_norm / autograd, %d threads): %d fine-tune epoch(s) of a -M "
                     "session-1 episode (%d support + %d query 84x84 images, step, validation) in %.1f s = %.1f img/s, "
                     "scaled by the mean %.0f image-forwards per episode" % (threads, epochs, n_img, n_img, t, img_s, avg_imgs),
           "thread_sweep_img_per_s": {str(k): round(v, 1) for k, v in sorted(sweep.items())}, "hardware_threads": max_threads}
    # secondary: the NumPy oracle (the parity checker itself), forward only
    onet = ResNetRef(syn.make_state_dict(1))
    x = syn.make_images(1, 8, 84)
    onet.features(x[:2])
    t0 = time.time()
    onet.features(x)
    tn = time.time() - t0
    try:                                                      # threads the oracle's BLAS calls actually ran on
        from threadpoolctl import threadpool_info
        nthreads = max([int(p.get("num_threads", 1)) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        nthreads = os.cpu_count()
    out["port_numpy"] = {"value": 8 / tn / avg_imgs, "unit": "episodes/s", "cores": nthreads, "kind": "port-numpy",
                         "sample": "NumPy oracle eval-mode forward of 8 images in %.1f s (%.1f img/s)" % (tn, 8 / tn)}
    return out


def measure_forward_ms(net, dev, batches, iters=6):
    """Eval-mode backbone forward time (ms, HIP events on the launch stream, random 84x84 images) at each batch size in `batches`."""
    hb = net.hip_backbone()
    out = {}
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    for n in sorted(set(int(b) for b in batches)):
        x = torch.randn(n, 3, 84, 84, device=dev, generator=g)
        for _ in range(2):
            hb.forward(x, check_params=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            hb.forward(x, check_params=False)
        e1.record()
        torch.cuda.synchronize()
        out[n] = e0.elapsed_time(e1) / iters
    return out


def sweep_model(args, net, dev, session_seconds):
    """Inputs of the level-2 sweep model measured on THIS GPU: the eval-mode forward at the per-rank batches
    a seed shared by g ranks produces (ceil(125 (s + 2) / g) images per epoch of session s, ceil(base / g) for the base
    evaluation), and the per-epoch cost that does not shard (classifier step + validation, from the timed sessions: session
    seconds minus forward time).  The feature all-gather itself cannot be measured on one GPU: 30 us per call is assumed
    (1.8 MB at most over 7 xGMI links x ~153 GB/s is ~2-12 us + launch) and stated."""
    from subreg_hip import sweep
    E, nb = args.epochs, args.base_batch
    groups = sorted({len(rk) for w in (2, 4, 8) for rnd in sweep.plan_sweep(range(args.sweep_seeds or 10), w) for _sd, rk in rnd} | {1})
    need = set()
    for g in groups:
        need |= {-(-(125 * (s + 2)) // g) for s in range(8)} | {-(-nb // g)}
    t = measure_forward_ms(net, dev, need)
    fixed_ms = None
    if session_seconds:
        # per-epoch cost beside the forward, from the timed region: (session time - forwards) / epochs, averaged over sessions
        fx = [(1e3 * session_seconds[s] - E * t[125 * (s + 2)] - t[nb]) / E for s in sorted(session_seconds)]
        fixed_ms = max(0.0, float(np.mean(fx)))
    gather_ms = 0.030
    seed_ms = {}
    for g in groups:
        seed_ms[g] = sum(E * (t[-(-(125 * (s + 2)) // g)] + (fixed_ms or 0.1) + (gather_ms if g > 1 else 0.0)) + t[-(-nb // g)] for s in range(8))
    out = {"forward_ms_at_batch": {str(k): round(v, 4) for k, v in sorted(t.items())}, "fixed_ms_per_epoch": fixed_ms,
           "assumed_allgather_ms": gather_ms, "seed_run_ms_by_group_size": {str(g): round(v, 1) for g, v in seed_ms.items()},
           "dp_efficiency_by_group_size": {str(g): round((seed_ms[1] / seed_ms[g] - 1.0) / (g - 1), 4) for g in groups if g > 1}}
    for w in (2, 4, 8):
        out["speedup_at_%d_gpus" % w] = round(sweep.sweep_speedup_measured(args.sweep_seeds or 10, w, seed_ms), 3)
    return out


def sweep_model_wall(args, model, sweep_leg):
    """speedup_at_N_gpus above compares seed RUNS only.  A sweep's wall clock also holds, per seed, its set-up (measured: the
    one-rank sweep leg's wall time minus its runs, per seed - building the synthetic backbone and inputs stands in for reading a
    checkpoint) and, for a seed shared by g > 1 ranks, the broadcast of its backbone (ONE flat 105 MB RCCL broadcast since round 5:
    modelled at the per-link xGMI rate, 153 GB/s, + 0.1 ms; it has never run between two GPUs - no 8-GPU node in this pool).
    NO N > 1 RUN EXISTS: every figure here is this GPU's measurements put through sweep.plan_sweep."""
    from subreg_hip import sweep
    n = args.sweep_seeds or 10
    seed_ms = {int(k): float(v) for k, v in model["seed_run_ms_by_group_size"].items()}
    setup_ms = max(0.0, 1e3 * (float(sweep_leg["seconds"]) - float(sweep_leg["seconds_runs_only"])) / n)
    bcast_ms = 105.2e6 / 153e9 * 1e3 + 0.1
    out = {"setup_ms_per_seed_measured": round(setup_ms, 1), "group_broadcast_ms_modelled": round(bcast_ms, 3),
           "note": "no run with more than one rank exists (single-GPU pool): modelled from one GPU's measurements"}
    t1 = n * (setup_ms + seed_ms[1])
    for w in (2, 4, 8):
        t = 0.0
        for rnd in sweep.plan_sweep(range(n), w):
            t += max(setup_ms + (bcast_ms if len(rk) > 1 else 0.0) + seed_ms[len(rk)] for _sd, rk in rnd)
        out["wall_speedup_at_%d_gpus" % w] = round(t1 / t, 3)
    return out


def route_a(args, net, opt, meta, dev, n_epochs=16):
    """Throughput of drop-in route A: the reference's UNCHANGED loop statements (language_eval.py:242-326)
    over the drop-in modules - nn.Module forward through the HIP backbone, torch autograd for CE + regloss + LangPuller,
    torch.optim.SGD, one forward per query set - at the shape of the LAST session (125 support images, 8 query sets of 125,
    100 classifier rows).  The fused loop (IncrementalRunner, the headline) runs the same epoch as one batched forward + three
    launches; both are reported as fine-tune epochs per second."""
    from subreg_hip.resnet_language import LangPuller
    with torch.no_grad():
        net.classifier.weight = torch.nn.Parameter(net.classifier.weight.detach()[:60].clone())
    base_weight, base_bias = net._get_base_weights()
    for _ in range(8):
        net.augment_base_classifier_(5)
    support_xs = meta[7][0][0]
    support_ys = torch.randint(60, 100, (125,), device=dev)
    queries = [(meta[s][2][0], torch.randint(60, 100, (125,), device=dev)) for s in range(8)]
    puller = LangPuller(opt, ["b"] * 60, ["n"] * 5)
    for name, prm in net.named_parameters():                          # freeze_backbone_weights (eval/util.py:62-69)
        prm.requires_grad = name.startswith("classifier")
    optimizer = torch.optim.SGD(net.parameters(), lr=opt.learning_rate, momentum=opt.momentum, weight_decay=opt.weight_decay)
    criterion = torch.nn.CrossEntropyLoss()
    net.eval()

    def epoch():
        output = net(support_xs)
        loss = criterion(output, support_ys)
        loss = loss + net.regloss(opt.lmbd_reg_transform_w, base_weight, base_bias)
        pullers = puller.get_projected_weight(base_weight, net.classifier.weight[60:, :])
        loss = loss + puller.loss1(opt.label_pull, pullers, net.classifier.weight[60:, :])
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        lv = loss.item()                                               # the reference reads the loss every epoch (:296)
        accs = []
        with torch.no_grad():
            for qx, qy in queries:                                     # validate(): one forward per query set (:18-43)
                out = net(qx)
                accs.append((out.argmax(1) == qy).float().sum().item())
        return lv, accs
    hb = net.hip_backbone()

    def timed_epochs():
        for _ in range(4):         # (the module's per-shape graphs and its input-sequence prefetch are in place after three epochs)
            epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_epochs):
            epoch()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_epochs
    # the same loop with the module's prefetch switched off (what round 5 measured), then as shipped - same process, same box
    depth = int(hb.EVAL_PREFETCH)
    hb.EVAL_PREFETCH = 0
    dt_off = timed_epochs()
    hb.EVAL_PREFETCH = depth
    hb.prefetch_hits = 0
    dt = timed_epochs()
    with torch.no_grad():
        net.classifier.weight = torch.nn.Parameter(net.classifier.weight.detach()[:60].clone())
    return {"epochs_per_s": 1.0 / dt, "ms_per_epoch": dt * 1e3, "images_per_epoch": 1125, "epochs_per_s_without_prefetch": 1.0 / dt_off,
            "prefetch": {"depth": int(hb.EVAL_PREFETCH), "forwards_served_from_a_prefetch": int(hb.prefetch_hits),
                         "forwards": 9 * (n_epochs + 4), "cached_graphs": len(hb._graphs),
                         "streams_found_parallel": getattr(hb, "prefetch_streams_calibrated", None)},
            "shape": "session 8 of 8: 125 support + 8 x 125 query images, 9 backbone forwards, torch autograd + SGD on classifier.weight [100, 640]"}


def pretrain_leg(args):
    """BASELINE.json configs[2]: the train_supervised.py step (train-mode forward with stash, full backward, SGD on all 26.29 M
    parameters; train_supervised.py:205-268) at the reference's batch 64 and at 128, bf16, synthetic 84x84 images.
    Algorithmic work 24.339 GFLOP per image."""
    from subreg_hip import synthetic as syn
    from subreg_hip.resnet_language import create_model
    from subreg_hip.train import SGD, GraphedStep
    dev = torch.device("cuda", torch.cuda.current_device())
    out = {"workload": "train_supervised.py step, ResNet18, 60 classes, SGD(0.05, 0.9, 5e-4), synthetic 84x84 (BASELINE.json configs[2])",
           "flop_per_image": 24.339e9, "dtype": args.dtype, "batches": {}}
    for B in (64, 128):
        net = create_model("resnet18", 60, SimpleNamespace(no_dropblock=True, linear_bias=False, hip_dtype=args.dtype))
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in syn.make_state_dict(1, randomize_bn=False).items()})
        net = net.to(dev).train()
        sgd = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
        crit = torch.nn.CrossEntropyLoss()
        x = torch.randn(B, 3, 84, 84, device=dev)
        y = torch.randint(0, 60, (B,), device=dev)

        def eager_step():
            loss = crit(net(x), y)
            sgd.zero_grad()
            loss.backward()
            sgd.step()
        # the step as the pretraining driver runs it (subreg_hip.pretrain.train): eager launches on the library's two streams.  Beside
        # it the opt-in form (opt.hip_graph): ONE hipGraph per batch shape and learning rate, replayed (train.GraphedStep: same kernels,
        # same two streams; inputs copied into the graph's buffers every step) - it frees the host, it does not shorten the step.
        graphed = GraphedStep(net, sgd, lambda xx, yy: crit(net(xx), yy))

        def step():
            graphed(x, y)

        def timed(fn, n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        for _ in range(8):
            eager_step()
        dt = min(timed(eager_step, 10), timed(eager_step, 10))
        for _ in range(5):                                             # (warm-up calls, the capture, first replays)
            step()
        dt_graph = timed(step, 20)
        is_graph = graphed.replays > 0
        tf = B * 24.339e9 / dt / 1e12
        # kernel launches of ONE step, counted live by the profiler's kernel records (library kernels and torch's alike); None when
        # the profiler is not usable on this box (profiles/r05_train_timeline_two_streams.txt holds the rocprofv3 count)
        launches = None
        try:
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                eager_step()
                torch.cuda.synchronize()
            launches = sum(1 for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA")
                           and "memcpy" not in e.name.lower() and "memset" not in e.name.lower()) or None
        except Exception:
            launches = None
        out["batches"][str(B)] = {"ms_per_step": dt * 1e3, "images_per_s": B / dt, "launches_per_step": launches,
                                  "hip_graph": False, "ms_per_step_as_one_hip_graph": dt_graph * 1e3 if is_graph else None,
                                  "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                                               "frac": tf / PEAK_TFLOPS[args.dtype]}}
        del net, sgd, x, y, graphed
    return out


def leg_in_child(args, flag, key, deadline):
    """An extra leg in a child process (started, never exec'ed, from this GPU-holding process), killed by PID at the deadline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), flag, "--dtype", args.dtype]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        so, se = proc.communicate(timeout=deadline)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"error": "%s leg did not finish within %d s" % (key, deadline)}
    for ln in so.splitlines():
        if ln.startswith("{") and ('"%s"' % key) in ln:
            return json.loads(ln)[key]
    return {"error": "%s child exited with code %s: %s" % (key, proc.returncode, se[-400:])}


def self_launch(args, argv):
    """`python bench.py --gpus N` with no launcher: start N ranks as a child `torch.distributed.run` job and relay its output.
    This parent never initialises the GPU (no torch.cuda call happens before this point), and nothing is exec'ed from a
    process that has: the ranks are fresh child processes (scripts/continual/slurm_subspace_reg.sh:7-8 starts one process
    per GPU the same way, through the SLURM array)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    # dmabuf IPC (the image exports this already): the pool's host driver has no legacy IPC handles, and without it RCCL /
    # cross-process device-tensor sharing fails with `hipIpcGetMemHandle: invalid argument`
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if line is not None or proc.returncode else 1


def run_sweep(args, rank, world, dev, host_only):
    """The 10-seed x 8-session sweep (BASELINE.json configs[3]) on this process group: plan = sweep.plan_sweep.
    Returns the dict reported under "sweep" (rank 0's copy is printed)."""
    import torch.distributed as dist
    from subreg_hip import sweep
    plan = sweep.plan_sweep(range(1, args.sweep_seeds + 1), world)
    # every rank creates every group, in the same order (torch.distributed.new_group is collective over the world)
    groups = {tuple(ranks): (dist.new_group(ranks) if world > 1 and len(ranks) > 1 else None)
              for rnd in plan for _seed, ranks in rnd}
    if not host_only:
        from subreg_hip.incremental import IncrementalRunner
        torch.cuda.synchronize()
    sweep.barrier()
    t0 = time.perf_counter()
    done, busy = [], 0.0        # busy: this rank's seconds inside seed runs (set-up of the synthetic backbone / inputs excluded)
    mine = [(sd, rk) for rnd in plan for sd, rk in rnd if rank in rk]
    # a real sweep reads the next seed's checkpoint while this one runs; here: its random initialisation, on a host thread
    prefetch, pool = {}, None
    if not host_only:
        from concurrent.futures import ThreadPoolExecutor
        from subreg_hip import synthetic as syn
        pool = ThreadPoolExecutor(max_workers=1)

        def want(i):
            if i < len(mine) and mine[i][0] not in prefetch:
                prefetch[mine[i][0]] = pool.submit(syn.make_state_dict, mine[i][0], randomize_bn=False)
        want(0)
    pos = -1
    for rnd in plan:
        for seed, ranks in [(sd, rk) for sd, rk in rnd if rank in rk]:
            pos += 1
            group = groups[tuple(ranks)]
            if host_only:                                   # control-path self-test: the plan, groups and collectives, no GPU work
                t = torch.tensor([float(seed)])
                if group is not None:
                    dist.broadcast(t, ranks[0], group=group)
                time.sleep(0.01)
            else:
                sd_host = prefetch.pop(seed).result()
                want(pos + 1)
                net, opt = make_net(args, seed, dev, sd_host)   # same seed -> same synthetic backbone on every rank of the group ...
                if group is not None:
                    sweep.broadcast_module(net, ranks[0], group)   # ... a real sweep loads it on the leader only
                meta, base = make_run_inputs(seed, dev, args.base_batch)
                shard = sweep.RowShard(group) if group is not None else None
                torch.cuda.synchronize()
                t_run = time.perf_counter()                     # (runs-only clock: starts behind the group broadcast and the set-up)
                r = IncrementalRunner(net, meta, base, opt, None, None, None, args.epochs_per_sync or args.epochs, False,
                                      verbose=False, row_shard=shard).start()
                for idx in range(r.iter_num):
                    r.run_session(idx)
                r.finish()
                torch.cuda.synchronize()
                busy += time.perf_counter() - t_run
                del r, net, meta, base
            if rank == ranks[0]:
                done.append(seed)
    if not host_only:
        torch.cuda.synchronize()
        pool.shutdown(wait=False)
    sweep.barrier()
    dt = sweep.max_over_ranks(time.perf_counter() - t0, None if host_only else dev)
    # makespan of the seed runs themselves: every rank's runs are serial and a shared seed's ranks wait for each other inside its
    # collectives, so the slowest rank's busy time is the sweep's run time.  `value` is on that (the same quantity as the
    # headline's timed region: runner start -> finish); the wall time incl. building each seed's synthetic backbone (kaiming
    # init on the host, 10 BN warm-up batches) and inputs is kept as `seconds_with_setup`.
    dt_run = dt if host_only else sweep.max_over_ranks(busy, dev)
    seen = sorted(x for r in sweep.gather_results(done) for x in r)
    # `value` is on the WALL clock of the whole leg (barrier to barrier: per-seed set-up, the group broadcasts and the runs), the
    # same basis at every rank count, so that value(N) / value(1) compares like with like; the runs-only figure is beside it
    return {"workload": "%d seeds x 8 sessions (BASELINE.json configs[3])" % args.sweep_seeds, "value": args.sweep_seeds * 8 / dt,
            "unit": "episodes/s", "seconds": dt, "value_runs_only": args.sweep_seeds * 8 / dt_run, "seconds_runs_only": dt_run,
            "scaling": "strong", "seeds_done": seen,
            "plan": [[[sd, len(rk)] for sd, rk in rnd] for rnd in plan],
            "n_ranks_seen": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
            "model_speedup_over_1_gpu": sweep.sweep_speedup(args.sweep_seeds, world)}


def sweep_in_child(args):
    """One-rank sweep leg in a child process (started, never exec'ed, from this GPU-holding process), killed by PID at the deadline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--sweep-only", "--sweep-seeds", str(args.sweep_seeds), "--epochs", str(args.epochs),
           "--dtype", args.dtype, "--base-batch", str(args.base_batch), "--epochs-per-sync", str(args.epochs_per_sync)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        so, se = proc.communicate(timeout=args.sweep_deadline)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"error": "sweep leg did not finish within %d s" % args.sweep_deadline}
    for ln in so.splitlines():
        if ln.startswith("{") and '"sweep"' in ln:
            return json.loads(ln)["sweep"]
    return {"error": "sweep child exited with code %s: %s" % (proc.returncode, se[-400:])}


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
    ap.add_argument("--sweep-seeds", type=int, default=10,
                    help="also run the S-seed x 8-session sweep (configs[3]) after the timed region; 0 = skip")
    ap.add_argument("--sweep-deadline", type=int, default=900, help="seconds after which the sweep leg is abandoned")
    ap.add_argument("--sweep-only", action="store_true", help="(internal) run only the one-rank sweep leg and print {\"sweep\": ...}")
    ap.add_argument("--route-a-only", action="store_true", help="(measurements) run only the route-A leg and print {\"route_a\": ...}")
    ap.add_argument("--pretrain-only", action="store_true", help="(internal) run only the pretraining-step leg and print {\"pretrain\": ...}")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the pretraining-step, route-A and sweep-model legs")
    ap.add_argument("--reuse-features", action="store_true",
                    help="NOT the headline: opt-in frozen-feature reuse (reported in DESIGN.md only)")
    ap.add_argument("--selftest-one-gpu", action="store_true",
                    help="run all ranks on cuda:0 over gloo (the pool's boxes have one GPU and RCCL does not put two ranks on one "
                         "device): exercises the real multi-rank code path - barriers, sweep plan, group broadcast, row-sharded "
                         "forwards, feature gather - with real kernels; throughput numbers of such a run mean nothing")
    ap.add_argument("--selftest-host", action="store_true",
                    help="control-path self-test WITHOUT a GPU (gloo): launch, rendezvous, sweep plan, collectives, JSON relay; "
                         "computes nothing and reports value 0")
    args = ap.parse_args()

    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    import torch.distributed as dist
    from subreg_hip import sweep
    if args.selftest_host:
        if world > 1:
            dist.init_process_group("gloo")
        sw = run_sweep(args, rank, world, None, True) if args.sweep_seeds > 0 else None
        if rank == 0:
            print(json.dumps({"metric": "incremental episodes/sec, ResNet18 miniImageNet 5w5s", "value": 0.0, "unit": "episodes/s",
                              "n_gpus": world, "steps": 0, "warmup": 0, "data": "none (host control-path self-test)", "sweep": sw}),
                  flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.selftest_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.selftest_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # RCCL; the timed region uses it only for the barrier and the max-over-ranks time

    if args.sweep_only:
        assert world == 1
        print(json.dumps({"sweep": run_sweep(args, 0, 1, dev, False)}), flush=True)
        return
    if args.pretrain_only:
        assert world == 1
        print(json.dumps({"pretrain": pretrain_leg(args)}), flush=True)
        return
    from subreg_hip.incremental import IncrementalRunner
    seed = rank + 1                                           # one seed per GPU, like the SLURM array
    net, opt = make_net(args, seed, dev)
    meta, base = make_run_inputs(seed, dev, args.base_batch)
    eps = args.epochs_per_sync or args.epochs
    if args.route_a_only:
        assert world == 1
        print(json.dumps({"route_a": route_a(args, net, opt, meta, dev, n_epochs=16)}), flush=True)
        return

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
    session_mix, session_sec = [0] * 8, {}
    for i in range(args.steps):
        if i % 8 == 0:
            r = new_runner(True)
            runners.append(r)
        ts = time.perf_counter()
        r.run_session(i % 8)                                  # (returns with the session's results on the host: synchronised)
        session_sec.setdefault(i % 8, []).append(time.perf_counter() - ts)
        session_mix[i % 8] += 1
    torch.cuda.synchronize()
    sweep.barrier()
    torch.cuda.synchronize()
    dt = sweep.max_over_ranks(time.perf_counter() - t0, dev)

    imgs = sum(rr.images_forwarded for rr in runners)
    fwd_ms = sum(e0.elapsed_time(e1) for rr in runners for (e0, e1, _n) in rr.fwd_events)
    n_fwd = sum(len(rr.fwd_events) for rr in runners)
    achieved = imgs * FLOP_PER_IMAGE / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
    traffic, traffic_head = None, None
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath)).get(args.dtype) or {}
        traffic = tj.get("bytes_per_image")                        # HBM bytes per image forwarded (FETCH_SIZE / WRITE_SIZE passes)
        traffic_head = tj.get("head")                              # the commit those counter passes were recorded on
    # `value` = K episodes / time, as the contract says; step i is session i mod 8, so a K that is not a multiple of 8 over-weights
    # the cheap early sessions (session s forwards 125 (s + 2) images per epoch).  The session-balanced figure beside it is
    # images/s over the mean image count of the 8 sessions of a run (incl. each run's initial base evaluation): it does not move
    # with K, and equals `value` when K is a multiple of 8.
    mean_imgs = float(np.mean([images_per_episode(s_, args.epochs, args.base_batch) for s_ in range(8)])) + args.base_batch / 8.0
    out = {
        "metric": "incremental episodes/sec, ResNet18 miniImageNet 5w5s", "value": args.steps * world / dt,
        "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic" if not args.selftest_one_gpu else "synthetic (SELF-TEST: all ranks share one GPU; not a measurement)",
        "config": {"workload": "8-session FSCIL, subspace regularizer, -M (BASELINE.json configs[1]); one seed per GPU",
                   "epochs_per_episode": args.epochs, "images_per_gpu": imgs, "base_batch": args.base_batch,
                   "session_mix": session_mix, "images_per_episode_session_balanced": mean_imgs,
                   "feature_reuse": bool(args.reuse_features), "backbone": "ResNet18 (RFS ResNet-12 family, 8.1219 GFLOP/img)",
                   "n_ranks_seen": dist.get_world_size() if world > 1 else 1},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                     "frac": achieved / PEAK_TFLOPS[args.dtype], "traffic": traffic, "traffic_recorded_at": traffic_head,
                     "kernel": "conv_fwd_kernel family over %d backbone forwards (%.1f ms each)" % (n_fwd, fwd_ms / max(n_fwd, 1))},
        "images_per_s": imgs * world / dt,
        "episodes_per_s_balanced": imgs * world / dt / mean_imgs,   # session-balanced: invariant in --steps
        "epochs_per_s": args.steps * world * args.epochs / dt,     # fine-tune epochs (forward + step + validation) per second, SURVEY.md 8d
    }
    # context, RECORDED (not measured by this run): the reference's own route on this GPU - the same network as plain PyTorch-ROCm
    # modules through MIOpen, tools/torch_rocm_baseline.py.  `vs_baseline` stays null: BASELINE.md holds no published number.
    rpath = os.path.join(REPO, "profiles", "torch_rocm_baseline.json")
    if os.path.exists(rpath) and args.dtype == "bf16":
        try:
            rj = json.load(open(rpath))
            out["reference_route_on_this_gpu"] = {
                "recorded": True, "recorded_at": rj.get("recorded_at"), "file": rj.get("file"),
                "episodes_per_s_fp32": rj.get("episodes_per_s_fp32"), "episodes_per_s_bf16_autocast": rj.get("episodes_per_s_bf16_autocast"),
                "this_line_over_fp32": out["episodes_per_s_balanced"] / world / rj["episodes_per_s_fp32"],
                "this_line_over_bf16_autocast": out["episodes_per_s_balanced"] / world / rj["episodes_per_s_bf16_autocast"],
                "what": rj.get("what")}
        except Exception:                                          # noqa: BLE001
            pass
    session_seconds = {s_: float(np.mean(v)) for s_, v in session_sec.items()}
    del runners, r
    if rank == 0:                                              # the headline is on record before any extra leg starts
        print("[bench.py headline, extra legs follow] " + json.dumps(out), file=sys.stderr, flush=True)
    # ---- after the timed region: the 10-seed sweep (strong scaling).  This extra leg can not cost the headline line:
    #      one rank  -> it runs in a CHILD process (its own HIP context) that is killed at the deadline; a crash or hang there
    #                   is reported under "sweep", the parent prints the line regardless;
    #      N ranks   -> it needs this process group, so it runs here under a deadline thread that prints the line without it.
    if world == 1 and not args.no_extra_legs:
        # the feature-reuse figure and the measured inputs of the sweep model: in this process (they need the backbone), each wrapped -
        # nothing here can cost the line
        # SURVEY.md section 8d "report both": the same 8-session workload with the frozen backbone's (constant) eval-mode features
        # computed ONCE per session and reused by every later epoch - results identical to the headline's recomputation
        # (tests/test_hip_loop.py::test_feature_reuse_is_results_identical; eval/language_eval.py:252,321-326 feed constant inputs
        # through frozen weights).  NOT the headline: `value` above recomputes every forward, as the reference does.
        try:
            def reuse_run():
                with torch.no_grad():
                    net.classifier.weight = torch.nn.Parameter(net.classifier.weight.detach()[:60].clone())
                rr = IncrementalRunner(net, meta, base, opt, None, None, None, eps, True, verbose=False, profile=False).start()
                for s_ in range(8):
                    rr.run_session(s_)
                return rr
            reuse_run()
            torch.cuda.synchronize()
            tr = time.perf_counter()
            rr = reuse_run()
            torch.cuda.synchronize()
            tr = time.perf_counter() - tr
            out["feature_reuse"] = {"episodes_per_s": 8.0 / tr, "seconds_per_8_sessions": tr, "images_forwarded": int(rr.images_forwarded),
                                    "images_forwarded_headline_per_8_sessions": int(round(mean_imgs * 8)),
                                    "epochs_per_episode": args.epochs, "identical_to_headline": True,
                                    "speedup_over_headline": (8.0 / tr) / (out["episodes_per_s_balanced"] / world),
                                    "what": "same 8 sessions, E epochs each, classifier step + validation every epoch; backbone forward of "
                                            "every support / query / base-evaluation image once per session instead of once per epoch"}
            del rr
        except Exception as exc:                                   # noqa: BLE001
            out["feature_reuse"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        try:
            out["sweep_model"] = sweep_model(args, net, dev, session_seconds if len(session_seconds) == 8 else None)
        except Exception as exc:                                   # noqa: BLE001
            out["sweep_model"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if args.sweep_seeds > 0 and world == 1:
        del net, meta, base
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        out["sweep"] = sweep_in_child(args)
        if "sweep_model" in out and "speedup_at_8_gpus" in out["sweep_model"] and isinstance(out["sweep"], dict):
            out["sweep"]["model_speedup_at_8_gpus_measured_inputs"] = out["sweep_model"]["speedup_at_8_gpus"]
            try:                                       # the same model on the WALL clock: per-seed set-up and the group broadcast in
                out["sweep_model"].update(sweep_model_wall(args, out["sweep_model"], out["sweep"]))
            except Exception as exc:                   # noqa: BLE001 - reported in the line
                out["sweep_model"]["wall_error"] = "%s: %s" % (type(exc).__name__, exc)
    elif args.sweep_seeds > 0:
        import threading

        def give_up():
            if rank == 0:
                out["sweep"] = {"error": "sweep leg did not finish within %d s" % args.sweep_deadline}
                print(json.dumps(out), flush=True)
            os._exit(0 if rank == 0 else 1)
        timer = threading.Timer(args.sweep_deadline, give_up)
        timer.daemon = True
        timer.start()
        try:
            out["sweep"] = run_sweep(args, rank, world, dev, False)
        except Exception as exc:                                   # noqa: BLE001 - reported in the line, never silently dropped
            out["sweep"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        timer.cancel()
    if rank == 0 and world == 1 and not args.no_extra_legs:
        # BASELINE.json configs[2] (pretraining step) in a child process of its own, like the one-rank sweep leg
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["pretrain"] = leg_in_child(args, "--pretrain-only", "pretrain", 300)
        # route A (the reference's unchanged loop statements over the drop-in modules): a child process too - the leg as a user's script
        # runs it, in a process of its own.  (Inside THIS process, behind the fused-loop run and the streams it created, the same leg
        # measured 100-106 epochs/s where the fresh process gives 118-123: DESIGN.md section 4.8.)
        out["route_a"] = leg_in_child(args, "--route-a-only", "route_a", 300)
        if isinstance(out["route_a"], dict) and "error" not in out["route_a"]:
            out["route_a"]["process"] = "child of bench.py (fresh process)"
            if 7 in session_seconds:
                out["route_a"]["fused_loop_epochs_per_s_same_shape"] = args.epochs / (session_seconds[7] - 0.0)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.base_batch)
            except Exception as exc:                               # noqa: BLE001 - reported in the line
                out["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
