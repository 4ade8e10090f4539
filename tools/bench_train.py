#!/usr/bin/env python3
"""Pretraining-step benchmark (BASELINE.json configs[2]: train_supervised.py step, ResNet18, batch 64, bf16, 1 GPU):
train-mode forward with stash + backward + SGD(0.05, 0.9, 5e-4) on all 26.29 M parameters, synthetic 84x84 batch.
Algorithmic work: 24.339 GFLOP per image (fwd + dgrad + wgrad)."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd")]
import numpy as np   # noqa: E402
import torch         # noqa: E402

from subreg_hip import synthetic as syn                 # noqa: E402
from subreg_hip.resnet_language import create_model     # noqa: E402
from subreg_hip.train import SGD, GraphedStep           # noqa: E402
from types import SimpleNamespace                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--host-time", action="store_true", help="also print the host-side enqueue time of a step")
    ap.add_argument("--graph", action="store_true", help="run the step as one replayed hipGraph (train.GraphedStep)")
    ap.add_argument("--dropblock", action="store_true", help="DropBlock with block_size 5 (no --no_dropblock)")
    ap.add_argument("--randomize-bn", action="store_true", help="random BatchNorm parameters / running statistics")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    net = create_model("resnet18", 60, SimpleNamespace(no_dropblock=not a.dropblock, linear_bias=False, hip_dtype=a.dtype))
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in syn.make_state_dict(1, randomize_bn=a.randomize_bn).items()})
    net = net.to(dev).train()
    opt = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    crit = torch.nn.CrossEntropyLoss()
    x = torch.randn(a.batch, 3, 84, 84, device=dev)
    y = torch.randint(0, 60, (a.batch,), device=dev)

    def step():
        out = net(x)
        loss = crit(out, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    if a.graph:
        graphed = GraphedStep(net, opt, lambda xx, yy: crit(net(xx), yy))
        eager = step

        def step():                                                        # noqa: F811
            return graphed(x, y)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    if a.graph:
        print("graph replays so far: %d" % graphed.replays)
    if a.host_time:
        # host enqueue time of one step: the device is idle and its queues empty when the step's calls start, so what is timed
        # is Python + launch overhead alone (the step is host-bound wherever this exceeds the device time)
        ht = []
        for _ in range(a.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            ht.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        print("host enqueue time per step B=%d: median %.2f ms, min %.2f ms" % (a.batch, 1e3 * float(np.median(ht)), 1e3 * min(ht)))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("train step B=%d %s: %.2f ms/step, %.0f img/s, %.1f TFLOP/s algorithmic (24.339 GFLOP/img), loss %.3f" %
          (a.batch, a.dtype, dt * 1e3, a.batch / dt, a.batch * 24.339e9 / dt / 1e12, loss.item()))


if __name__ == "__main__":
    main()
