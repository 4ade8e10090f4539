#!/usr/bin/env python3
"""Whole eval-mode backbone forward (22 convs, pack, avgpool) at the bench's batch sizes, as a hipGraph replay, for
1..N eval lanes (HipBackbone.EVAL_LANES: sub-batches on concurrent streams).  A/B within one process (interleaved rounds).

  python tools/bench_forward.py [--batches 250,500,750,1125] [--lanes 1,2,3,4] [--rounds 5]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd")]
import numpy as np            # noqa: E402
import torch                  # noqa: E402

from subreg_hip import synthetic as syn            # noqa: E402
from subreg_hip.backbone import HipBackbone        # noqa: E402

FLOP = 8.1219e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="250,500,750,1125")
    ap.add_argument("--lanes", default="1,2,3,4")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = syn.make_state_dict(1, randomize_bn=False)
    params = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sd.items() if v.dtype != np.int64}
    hb = HipBackbone(params, (1, 1, 2, 2), a.dtype)
    lanes = [int(v) for v in a.lanes.split(",")]
    for B in [int(v) for v in a.batches.split(",")]:
        x = torch.randn(B, 3, 84, 84, device=dev)
        graphs, outs = {}, {}
        for nl in lanes:
            hb.EVAL_LANES = nl
            outs[nl] = torch.empty(B, 640, device=dev)
            hb.forward(x, out=outs[nl])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                hb.forward(x, out=outs[nl], check_params=False)
            graphs[nl] = g
        ref = outs[lanes[0]].clone()
        times = {nl: [] for nl in lanes}
        for _ in range(a.rounds):
            for nl in lanes:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _i in range(a.iters):
                    graphs[nl].replay()
                e1.record()
                torch.cuda.synchronize()
                times[nl].append(e0.elapsed_time(e1) / a.iters)
        for nl in lanes:
            same = bool(torch.equal(outs[nl], ref))
            ms = float(np.median(times[nl]))
            print("B=%5d lanes=%d  %7.3f ms (min %.3f)  %7.1f TFLOP/s  %4.1f%% of peak  identical=%s" %
                  (B, nl, ms, min(times[nl]), B * FLOP / ms * 1e-9, B * FLOP / ms * 1e-9 / 25.0, same), flush=True)


if __name__ == "__main__":
    main()
