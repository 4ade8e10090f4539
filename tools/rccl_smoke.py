#!/usr/bin/env python3
"""RCCL smoke on ONE GPU: backend "nccl" (= RCCL) with a world of one rank, driving every collective the build issues with the
tensors it issues them on - the benchmark's barrier / max-over-ranks, the sweep's backbone broadcast (fp32 parameters and the
int64 BatchNorm counters), the feature all-gather of the row-sharded forward, the per-seed result gather, a sub-group, and the
staged asynchronous gradient all-reduce of the pretraining step on views of the flat gradient buffer, overlapped with the
HIP backward.  With one rank every collective is the identity, so results are checked exactly; what this proves is that
RCCL initialises on the device, accepts these dtypes / views / async handles and orders them against the compute stream -
the pool has no multi-GPU box, the 2-rank runs of the same code go over gloo (tools/dp_check.py, dp_pretrain_check.py).

  python tools/rccl_smoke.py      (sets MASTER_ADDR / RANK / WORLD_SIZE itself)
"""
import os
import socket
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd")]


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    # dmabuf IPC: this pool's host driver supports no legacy IPC handles - without this RCCL's buffer registration fails with
    # `hipIpcGetMemHandle: invalid argument` (the image exports it already; kept for environments built by hand)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from types import SimpleNamespace
    from subreg_hip import pretrain as pt, sweep, synthetic as syn
    from subreg_hip.resnet_language import create_model
    from subreg_hip.train import SGD
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    # --- benchmark contract
    sweep.barrier()
    assert sweep.max_over_ranks(1.25, dev) == 1.25                       # fp64 MAX all-reduce on the device
    assert sweep.gather_results([1, 2]) == [[1, 2]]
    # --- sweep: sub-group, backbone broadcast (every state_dict tensor incl. int64 counters), feature all-gather
    grp = dist.new_group([0])
    net = create_model("resnet18", 60, SimpleNamespace(no_dropblock=True, linear_bias=False, hip_dtype="bf16"))
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in syn.make_state_dict(3, randomize_bn=False).items()})
    net = net.to(dev)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    n_b = sweep.broadcast_module(net, 0, grp, force=True)                # ONE flat RCCL broadcast per dtype (fp32 + the int64 counters)
    assert all(torch.equal(before[k], v) for k, v in net.state_dict().items())
    local = torch.randn(350, 640, device=dev)
    out = torch.empty(350, 640, device=dev)
    dist.all_gather_into_tensor(out, local.contiguous(), group=grp)      # RowShard.gather's collective
    assert torch.equal(out, local)
    # --- pretraining: staged asynchronous SUM all-reduce on views of the flat gradient buffer, behind the HIP backward
    net.train()
    sync = pt.GradientSync(grp)
    sync.world = 2                                                        # take the multi-rank code path with the 1-rank group
    net.hip_backbone().grad_stage_hook = sync.stage_ready
    opt = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    x = torch.from_numpy(syn.make_images(5, 6, 32)).to(dev)
    y = torch.randint(0, 60, (6,), device=dev)
    crit = torch.nn.CrossEntropyLoss()
    loss = crit(net(x), y)
    opt.zero_grad()
    loss.backward()
    g_before = [p.grad.clone() for p in net.parameters()]
    sync.finish(opt.params)
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.grad) for a, p in zip(g_before, net.parameters()))
    opt.step()
    torch.cuda.synchronize()
    calls = sync.calls
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL (backend nccl, 1 rank on %s): barrier, fp64 MAX all-reduce, object gather, sub-group, %d parameter / counter "
          "broadcasts, [350, 640] feature all-gather, %d asynchronous gradient all-reduces on flat-buffer views behind the staged "
          "backward + SGD step: all identities reproduced exactly -> OK" % (torch.cuda.get_device_name(0), n_b, calls))


if __name__ == "__main__":
    main()
