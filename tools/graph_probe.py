#!/usr/bin/env python3
"""Diagnostic: which part of the training step captures as a hipGraph?  Each configuration in a child process (a failing
hipStreamEndCapture takes the process down).  modes: step (train.GraphedStep), fwd (train-mode forward + loss), fwdbwd (+ backward),
evalstep (model.eval(): running statistics, no masks), opt (optimiser step alone)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd")]


def child(mode, B, hw, dtype):
    import faulthandler
    faulthandler.enable()
    import numpy as np
    import torch
    from types import SimpleNamespace
    from subreg_hip import synthetic as syn
    from subreg_hip.resnet_language import create_model
    from subreg_hip.train import SGD, GraphedStep
    net = create_model("resnet18", 60, SimpleNamespace(no_dropblock=True, linear_bias=False, hip_dtype=dtype))
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in syn.make_state_dict(1).items()})
    net = net.cuda().train()
    if mode == "evalstep":
        net.eval()
    opt = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    crit = torch.nn.CrossEntropyLoss()
    x = torch.randn(B, 3, hw, hw, device="cuda")
    y = torch.randint(0, 60, (B,), device="cuda")
    if mode in ("step", "evalstep"):
        st = GraphedStep(net, opt, lambda a, b: crit(net(a), b))
        hold = os.environ.get("PROBE_HOLD", "1") == "1"
        for i in range(5):
            print("call", i, file=sys.stderr, flush=True)
            if hold:
                loss = st(x, y)
            else:
                st(x, y)
        torch.cuda.synchronize()
        print("OK %s B=%d hw=%d %s replays=%d failed=%s hold=%s" % (mode, B, hw, dtype, st.replays, [e["failed"] for e in st.entries.values()], hold), flush=True)
        return
    hb = net.hip_backbone()

    def body():
        if mode == "opt":
            opt.step()
            return None
        loss = crit(net(x), y)
        if mode == "fwdbwd":
            opt.zero_grad()
            loss.backward()
        return loss
    for _ in range(2):
        loss = crit(net(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    hb.use_mask_params = True
    hb.set_mask_params(hw, hw)
    torch.cuda.synchronize()
    if mode == "opt":
        loss = crit(net(x), y)
        opt.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capture begin", file=sys.stderr, flush=True)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = body()
    print("capture end", file=sys.stderr, flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("OK %s B=%d hw=%d %s" % (mode, B, hw, dtype), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    for env in ({}, {"SUBREG_TRAIN_ONE_STREAM": "1"}):
        for mode in ("fwd", "fwdbwd", "opt", "evalstep", "step"):
            for B, hw, dtype in ((8, 32, "bf16"), (64, 84, "bf16")):
                e = dict(os.environ)
                e.update(env)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", mode, str(B), str(hw), dtype], env=e, capture_output=True, text=True)
                lines = (r.stdout + r.stderr).splitlines()
                tail = [ln for ln in lines if ln.startswith("OK") or "Error" in ln or "error" in ln or "Fatal" in ln or ln.startswith("call") or ln.startswith("capture")]
                where = [ln for ln in lines if ln.startswith("  File") and ("subreg_hip" in ln or "graphs.py" in ln)][:4]
                print(env, mode, (B, hw, dtype), "rc=%d" % r.returncode, tail[-3:], where, flush=True)
