#!/usr/bin/env python3
"""Input-sequence prefetch of HipBackbone.forward_graphed (route A's forwards): nine 125-image forwards per round, every result read on
the host before the next call (as eval/language_eval.py:36-43 does).  Run with SUBREG_EVAL_PREFETCH=0 / 1 / 2 / 3."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd")]
import numpy as np   # noqa: E402
import torch         # noqa: E402

from subreg_hip import synthetic as syn            # noqa: E402
from subreg_hip.backbone import HipBackbone        # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 125
    # argv[2]: streams created (and used once) BEFORE the backbone makes its own - HIP deals streams onto its hardware queues in creation
    # order, so this moves the prefetch streams onto other queues (bench.py's route-A leg runs after a fused-loop run that made many)
    dummies = [torch.cuda.Stream() for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 0)]
    for st in dummies:
        with torch.cuda.stream(st):
            torch.zeros(1, device="cuda").add_(1)
    sd = syn.make_state_dict(3)
    params = {k: torch.from_numpy(np.array(v)).cuda() for k, v in sd.items() if v.dtype != np.int64}
    hb = HipBackbone(params, (1, 1, 2, 2), "bf16")
    xs = [torch.randn(B, 3, 84, 84, device="cuda") for _ in range(9)]
    for _ in range(4):
        for x in xs:
            hb.forward_graphed(x).sum().item()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        for x in xs:
            hb.forward_graphed(x).sum().item()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("prefetch depth %d, %d streams made before: %d forwards of %d images, result read on the host after each: %.2f ms per round (%.3f ms per forward), hits %d"
          % (hb.EVAL_PREFETCH, len(dummies), len(xs), B, dt * 1e3, dt * 1e3 / len(xs), hb.prefetch_hits))


if __name__ == "__main__":
    main()
