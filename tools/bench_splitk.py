#!/usr/bin/env python3
"""Eval-mode 3x3 convolutions of the small maps with and without the K split over workgroups (subreg_conv_fwd_ws against
subreg_conv_fwd), at the batches of the reference's own 125-image forwards and of the sweep's row-sharded helpers.
  python tools/bench_splitk.py [batch ...]         (SUBREG_SPLITK_MAXBLOCKS / SUBREG_SPLITK_SLOTS move the rule)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))
import torch   # noqa: E402

from subreg_hip import _lib   # noqa: E402

LAYERS = [("L3.1.conv1/2", 10, 320, 320), ("L4.0.conv2", 5, 640, 640), ("L4.1.conv1/2", 5, 640, 640)]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    batches = [int(v) for v in sys.argv[1:]] or [63, 125, 250]
    for B in batches:
        for name, H, Cin, Cout in LAYERS:
            if name == "L4.0.conv2":
                H = 10
            npix = B * H * H
            x = torch.randn(npix, Cin, device=dev).to(torch.bfloat16)
            w = (torch.randn(Cout, 9, Cin, device=dev) / (Cin * 9) ** 0.5).to(torch.bfloat16)
            shift = torch.randn(Cout, device=dev)
            y0 = torch.empty(npix, Cout, device=dev, dtype=torch.bfloat16)
            y1 = torch.empty_like(y0)
            need = lib.subreg_conv_splitk_floats(B, H, H, Cin, Cout, 3, _lib.BF16)
            ws = torch.empty(max(int(need), 1), device=dev, dtype=torch.float32)

            def plain():
                _lib.check(lib.subreg_conv_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y0), None, _lib.ptr(shift), None, None, None, None, 0,
                                               B, H, H, Cin, Cout, 3, _lib.CONV_LRELU, _lib.BF16, _lib.stream_ptr()))

            def split():
                _lib.check(lib.subreg_conv_fwd_ws(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y1), None, _lib.ptr(shift), None, None, None, None, 0,
                                                  B, H, H, Cin, Cout, 3, _lib.CONV_LRELU, _lib.BF16, _lib.ptr(ws), int(need), _lib.stream_ptr()))
            res = []
            for fn in (plain, split):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 20)
            d = (y0.float() - y1.float()).abs().max().item()
            print("B=%4d %-14s M=%6d  ksplit %d  plain %6.1f us  split %6.1f us  (%.2fx)  max |diff| %.3g" %
                  (B, name, npix, int(need) // (npix * Cout) if need else 1, res[0], res[1], res[0] / res[1], d))


if __name__ == "__main__":
    main()
