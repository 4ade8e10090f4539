#!/usr/bin/env python3
"""Data-parallel pretraining check on ONE GPU: two processes (gloo; on an 8-GPU node the same code runs with backend nccl =
RCCL over xGMI, one GPU per rank) each take their share of a global batch, back-propagate loss * n_local / n_global through
the STAGED backward (GradientSync.stage_ready all-reduces each stage's gradient range while the earlier blocks' backward is
still queued), finish with the classifier, and step the fused SGD.  The result must equal - up to the run-to-run noise of the float atomics in
the 1x1 convolutions' weight gradients, which the tool measures - a single process that does what nn.DataParallel does (train_supervised.py:141-142): the same two shards through the same replica code (BatchNorm
statistics per replica), gradients summed, one optimiser step.  Uneven shards (7 = 4 + 3) are part of the check.

  python tools/dp_pretrain_check.py        (the parent never touches the GPU; it only starts the workers)
"""
import os
import socket
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd"), os.path.join(REPO, "tests")]

N_GLOBAL, HW, STEPS = 7, 32, 2


def make(dtype="bf16"):
    import numpy as np
    import torch
    from types import SimpleNamespace
    from subreg_hip import synthetic as syn
    from subreg_hip.resnet_language import create_model
    net = create_model("resnet18", 60, SimpleNamespace(no_dropblock=True, linear_bias=False, hip_dtype=dtype))
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in syn.make_state_dict(1, randomize_bn=False).items()})
    return net.cuda().train()


def batches():
    import numpy as np
    import torch
    from subreg_hip import synthetic as syn
    return [(torch.from_numpy(syn.make_images(10 + s, N_GLOBAL, HW)), torch.from_numpy(np.random.RandomState(20 + s).randint(0, 60, N_GLOBAL)))
            for s in range(STEPS)]


def worker(rank, world, port, q):
    import datetime
    import torch
    import torch.distributed as dist
    from oracle.resnet_ref import MaskSource
    from subreg_hip import pretrain as pt, sweep
    from subreg_hip.train import SGD
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    crit = torch.nn.CrossEntropyLoss()
    # ---- two ranks
    net = make()
    if rank != 0:
        with torch.no_grad():
            for t in net.state_dict().values():               # helpers start from garbage: the broadcast must fix it
                if t.is_floating_point():
                    t.mul_(0.5)
    sweep.broadcast_module(net, 0)
    sync = pt.GradientSync()
    net.hip_backbone().grad_stage_hook = sync.stage_ready
    opt = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    for s, (x, y) in enumerate(batches()):
        xs, ys = pt.shard_batch(x, y, rank, world)
        net.mask_source = MaskSource(100 * s + rank)
        loss = crit(net(xs.cuda()), ys.cuda())
        opt.zero_grad()
        (loss * (float(xs.shape[0]) / N_GLOBAL)).backward()
        sync.finish(opt.params)
        opt.step()
    torch.cuda.synchronize()
    dp = {k: v.detach().cpu() for k, v in net.named_parameters()}
    calls = sync.calls
    dist.barrier()
    # ---- rank 0 alone: both shards through one replica, gradients summed (what DataParallel computes).  Twice: the 1x1
    #      shortcut convolutions' weight gradients are accumulated with float atomics (order-dependent rounding), so two
    #      runs of the SAME single-process code differ in the last bit - the yardstick for the two-rank result
    if rank == 0:
        def emulate():
            ref = make()
            opt = SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
            for s, (x, y) in enumerate(batches()):
                total = None
                for r in range(world):
                    xs, ys = pt.shard_batch(x, y, r, world)
                    ref.mask_source = MaskSource(100 * s + r)
                    loss = crit(ref(xs.cuda()), ys.cuda())
                    opt.zero_grad()
                    (loss * (float(xs.shape[0]) / N_GLOBAL)).backward()
                    grads = [p.grad.clone() for p in ref.parameters()]
                    total = grads if total is None else [a + b for a, b in zip(total, grads)]
                flat = ref.hip_backbone()._train_stash.flat_grads      # hand the sums back as views of the flat buffer (fused step)
                for p, g in zip(ref.parameters(), total):
                    if p.grad._base is flat:
                        p.grad.copy_(g)
                    else:
                        p.grad = g
                opt.step()
            torch.cuda.synchronize()
            return {k: v.detach().cpu() for k, v in ref.named_parameters()}
        r1, r2 = emulate(), emulate()
        scale = max(float(v.abs().max()) for v in r1.values())
        d_dp = max(float((dp[k] - r1[k]).abs().max()) for k in r1)
        d_self = max(float((r2[k] - r1[k]).abs().max()) for k in r1)
        q.put((d_dp, d_self, scale, calls))
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    d_dp, d_self, scale, calls = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    ok = d_dp <= max(4 * d_self, 1e-6 * scale)
    print("data-parallel pretraining, 2 ranks x 1 GPU (gloo), global batch %d = 4 + 3, %d steps, staged + overlapped all-reduce "
          "(%d collectives): max |parameter difference| to the single-process DataParallel emulation %.3g; two runs of that "
          "emulation differ by %.3g (float atomics in the 1x1 weight gradients); parameter scale %.3g -> %s" %
          (N_GLOBAL, STEPS, calls, d_dp, d_self, scale, "EQUIVALENT" if ok else "MISMATCH"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
