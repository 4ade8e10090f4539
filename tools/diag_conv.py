#!/usr/bin/env python3
"""Per-wave cycle breakdown of the conv main loop (needs a library built with -DSUBREG_DIAG=3; GPU only).

  python tools/diag_conv.py [--batch 256] [--only L3.0.conv2]
For every layer of tools/bench_conv.py prints the median over waves of: prologue, loop, and inside the loop the cycles
spent issuing DMAs, in LDS reads + MFMAs, in the end-of-step vmcnt wait and in the barrier, the shader clock
(s_memtime / s_memrealtime x 100 MHz) and the epilogue (end of loop -> last store issued).
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))
sys.path.insert(0, os.path.join(REPO, "tools"))

import torch  # noqa: E402

from subreg_hip import _lib  # noqa: E402
from bench_conv import LAYERS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--only", default="")
    ap.add_argument("--kernel", default="general", choices=["general", "wide"])
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    dt, td = _lib.dtype_code("bf16"), torch.bfloat16
    B = a.batch
    print("%-24s %8s %8s | %7s %7s %7s %7s | %5s %5s" % ("layer", "prolog", "loop", "issue", "mma", "wait", "barrier", "GHz", "epilog"))
    for name, H, Cin, Cout, k, pool, cin2, count in LAYERS:
        if a.only and a.only not in name:
            continue
        if Cin in (0, 3, -3) or cin2 == 3:          # layer 1 runs on its own kernels (conv_first / conv64_resident): no stamps there
            continue
        npix = B * H * H
        x = torch.randn(npix, Cin, device=dev).to(td)
        w = (torch.randn(Cout, k * k, Cin, device=dev) / (Cin * k * k) ** 0.5).to(td)
        shift = torch.randn(Cout, device=dev)
        Ho = H // 2 if pool else H
        y = torch.empty(B * Ho * Ho, Cout, device=dev, dtype=td)
        x2 = w2 = None
        c2 = 0
        if cin2 >= 0:
            c2 = Cout if cin2 == 0 else cin2
            x2 = torch.randn(npix, c2, device=dev).to(td)
            w2 = (torch.eye(Cout, device=dev) if cin2 == 0 else torch.randn(Cout, c2, device=dev) / c2 ** 0.5).to(td).contiguous()
        flags = _lib.CONV_LRELU | (_lib.CONV_POOL2 if pool else 0) | (_lib.CONV_KERNEL_WIDE if a.kernel == "wide" else _lib.CONV_KERNEL_GENERAL)
        stamps = torch.zeros(((npix + 63) // 64) * ((Cout + 63) // 64) * 8 * 8 + 1024, device=dev)

        def run():
            _lib.check(lib.subreg_conv_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), None, _lib.ptr(shift), None, _lib.ptr(stamps),
                                           _lib.ptr(x2), _lib.ptr(w2), c2, B, H, H, Cin, Cout, k, flags, dt, _lib.stream_ptr()))
        for _ in range(30):
            run()
        torch.cuda.synchronize()
        st = stamps.view(-1, 8)
        st = st[st[:, 1] > 0]
        med = st.median(dim=0).values.tolist()
        ghz = (st[:, 0] + st[:, 1]).sum().item() / max(st[:, 6].sum().item(), 1.0) * 0.1
        print("%-24s %8.0f %8.0f | %7.0f %7.0f %7.0f %7.0f | %5.2f %5.0f" % (name[:24], med[0], med[1], med[2], med[3], med[4], med[5], ghz, med[7]))


if __name__ == "__main__":
    main()
