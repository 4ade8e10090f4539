#!/usr/bin/env python3
"""Intra-seed data parallelism check on ONE GPU: two processes (gloo, host-staged gather) share a session's forwards and
must reproduce the single-process run bit for bit (same losses, accuracies, classifier rows).

  python tools/dp_check.py [golden tag]     (the parent never touches the GPU; it only starts the workers)
With a freeze_backbone_at > 1 golden (hw32_freeze3) the pre-freeze epochs run on every rank and the leader's network is broadcast
after each step; the single-process run is itself not bit-reproducible there (float atomics in the weight-gradient kernels), so
the comparison is epochs and accuracies equal, losses and classifier rows within 1e-4.
On an 8-GPU node the same code path runs with backend nccl (RCCL over xGMI) and one GPU per rank."""
import os
import socket
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd"), os.path.join(REPO, "tests")]


def worker(rank, world, port, q, tag):
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from test_hip_loop import build_case
    from subreg_hip import sweep
    from subreg_hip.incremental import IncrementalRunner
    g = np.load(os.path.join(REPO, "tests", "golden", "loop_%s.npz" % tag))
    exact = int(g["opt.freeze_backbone_at"]) == 1 if "opt.freeze_backbone_at" in g.files else True

    def run(shard):
        net, opt, meta, base_loader, bsl, inits, picks = build_case(g, "f32")
        if shard is not None:
            with torch.no_grad():                      # helpers start from garbage: the broadcast must fix it
                if rank != 0:
                    for t in net.state_dict().values():
                        if t.is_floating_point():
                            t.mul_(0.5)
            sweep.broadcast_module(net, 0)
        r = IncrementalRunner(net, meta, base_loader, opt, bsl, inits, picks, 4, False, verbose=False, row_shard=shard).start()
        for idx in range(r.iter_num):
            r.run_session(idx)
        r.finish()
        return net.last_run, r.images_forwarded
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    dp_run, dp_imgs = run(sweep.RowShard())
    ref_run, ref_imgs = run(None) if rank == 0 else (None, None)
    if rank == 0:
        close = np.array_equal if exact else (lambda a, b: np.allclose(a, b, rtol=1e-4, atol=1e-5))
        same = (dp_run["epochs"] == ref_run["epochs"] and dp_run["test_acc"] == ref_run["test_acc"] and
                all(close(np.asarray(a), np.asarray(b)) for a, b in zip(dp_run["loss"], ref_run["loss"])) and
                close(dp_run["classifier_weight"], ref_run["classifier_weight"]))
        q.put((same, dp_imgs, ref_imgs, dp_run["epochs"]))
    dist.barrier()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tag = sys.argv[1] if len(sys.argv) > 1 else "hw32_M"
    procs = [ctx.Process(target=worker, args=(r, 2, port, q, tag)) for r in range(2)]
    for p in procs:
        p.start()
    same, dp_imgs, ref_imgs, epochs = q.get(timeout=400)
    for p in procs:
        p.join(timeout=120)
    print("%s: 2-rank data-parallel run == single-process run: %s (epochs %s; images forwarded by rank 0: %d of %d)" %
          (tag, same, epochs, dp_imgs, ref_imgs))
    sys.exit(0 if same and all(p.exitcode == 0 for p in procs) else 1)


if __name__ == "__main__":
    main()
