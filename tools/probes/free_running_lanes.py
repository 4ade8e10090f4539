#!/usr/bin/env python3
"""Probe (round 6): would the two eval lanes gain from NOT joining after every forward?

The fused loop's backbone is frozen and its inputs are constant, so the forward of epoch e + 1 does not depend on the classifier step of epoch
e: the two lanes (half the batch each, own workspaces, own streams) could run free of each other, one up to an epoch ahead, the step waiting
for both halves of ITS epoch.  Lanes that drift apart pair a layer-1 kernel (LDS / issue bound) with a wide-layer kernel (MFMA bound)
instead of running the same layer side by side.  This script times, on one box, interleaved:
  joined     hb.forward(x) with EVAL_LANES = 2 (fork / join inside every forward: what ships)
  free       the two halves on two streams, N forwards each, no join (optionally the second lane started `stagger` of a forward late)
Usage: free_running_lanes.py [B ...]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from subreg_hip import synthetic as syn
    from subreg_hip.backbone import HipBackbone
    batches = [int(a) for a in sys.argv[1:]] or [250, 500, 700, 1000, 1125]
    sd = syn.make_state_dict(3)
    params = {k: torch.from_numpy(np.array(v)).cuda() for k, v in sd.items() if v.dtype != np.int64}
    hb = HipBackbone(params, (1, 1, 2, 2), "bf16")
    N = 24
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    print("%6s %12s %12s %12s %12s   (ms per forward of B images; joined = what ships)" % ("B", "joined", "free", "free+0.5", "one lane"))
    for B in batches:
        x = torch.randn(B, 3, 84, 84, device="cuda")
        h = (B + 1) // 2
        xa, xb = x[:h].contiguous(), x[h:].contiguous()
        out = torch.empty(B, hb.out_dim, device="cuda")
        fa, fb = torch.empty(h, hb.out_dim, device="cuda"), torch.empty(B - h, hb.out_dim, device="cuda")

        def joined(lanes):
            hb.EVAL_LANES = lanes
            for _ in range(3):
                hb.forward(x, out=out, check_params=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(N):
                hb.forward(x, out=out, check_params=False)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / N * 1e3

        def free(stagger):
            hb.EVAL_LANES = 1
            cur = torch.cuda.current_stream()
            for _ in range(2):
                with torch.cuda.stream(s1):
                    hb.forward(xa, out=fa, check_params=False)
                with torch.cuda.stream(s2):
                    hb.forward(xb, out=fb, check_params=False, lane_base=1)
            torch.cuda.synchronize()
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            t0 = time.perf_counter()
            if stagger:
                with torch.cuda.stream(s2):
                    torch.cuda._sleep(int(stagger))
            for _ in range(N):
                with torch.cuda.stream(s1):
                    hb.forward(xa, out=fa, check_params=False)
                with torch.cuda.stream(s2):
                    hb.forward(xb, out=fb, check_params=False, lane_base=1)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / N * 1e3

        res = {"joined": [], "free": [], "free+0.5": [], "one": []}
        for _ in range(2):
            res["joined"].append(joined(2))
            res["free"].append(free(0))
            # half a lane-forward late: cycles of the 100 MHz counter torch.cuda._sleep spins on are not clock cycles - calibrate from `joined`
            res["free+0.5"].append(free(res["joined"][-1] * 0.5e-3 * 2.0e9))
            res["one"].append(joined(1))
        hb.EVAL_LANES = 2
        print("%6d %12s %12s %12s %12s" % (B, *("%.3f/%.3f" % tuple(res[k]) for k in ("joined", "free", "free+0.5", "one"))))


if __name__ == "__main__":
    main()
