#!/usr/bin/env python3
"""Weight-gradient (dW) kernels of the pretraining step through the C ABI, per 3x3 layer shape at batch B: time of
subreg_conv_wgrad (pad copies + dW kernel) and of subreg_unpack_wgrad, HIP events, random bf16 data.  Algorithmic work:
2 * B*H*W * 9*Cin * Cout FLOP per layer.

  python tools/bench_wgrad.py [--batch 64] [--iters 20]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))
import torch  # noqa: E402

from subreg_hip import _lib  # noqa: E402

LAYERS = [("L1.conv2/3", 84, 64, 64), ("L2.conv1", 42, 64, 160), ("L2.conv2/3", 42, 160, 160), ("L3.0.conv1", 21, 160, 320),
          ("L3.0.conv2/3", 21, 320, 320), ("L3.1.conv", 10, 320, 320), ("L4.0.conv1", 10, 320, 640), ("L4.0.conv2/3", 10, 640, 640),
          ("L4.1.conv", 5, 640, 640)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    lib, dev, B, dt = _lib.load(), torch.device("cuda:0"), a.batch, _lib.BF16
    s = _lib.stream_ptr
    tot = 0.0
    for name, H, Cin, Cout in LAYERS:
        x = torch.randn(B * H * H * Cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(B * H * H * Cout, device=dev).to(torch.bfloat16)
        ns = lib.subreg_conv_wgrad_splits(B, H, H, Cin, Cout, 3, dt)
        gw = torch.empty(ns * Cout * 9 * Cin, dtype=torch.float32, device=dev)
        grad = torch.empty(Cout * Cin * 9, dtype=torch.float32, device=dev)
        pads = [torch.empty(B * (H + 2) * (H + 2) * c, dtype=torch.bfloat16, device=dev) for c in (Cin, Cout)]

        def t_of(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / a.iters
        tw = t_of(lambda: _lib.check(lib.subreg_conv_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(gw), _lib.ptr(pads[0]), _lib.ptr(pads[1]),
                                                            B, H, H, Cin, Cout, 3, dt, s())))
        tu = t_of(lambda: _lib.check(lib.subreg_unpack_wgrad(_lib.ptr(gw), _lib.ptr(grad), Cout, Cin, 3, 0, ns, s())))
        fl = 2.0 * B * H * H * 9 * Cin * Cout
        tot += tw + tu
        print("%-14s M=%7d Cin=%3d Cout=%3d splits=%3d   pad+dW %7.1f us  (%6.1f TFLOP/s)   unpack %6.1f us" %
              (name, B * H * H, Cin, Cout, ns, tw, fl / tw * 1e-6, tu))
    print("sum over the nine shapes: %.1f us" % tot)


if __name__ == "__main__":
    main()
