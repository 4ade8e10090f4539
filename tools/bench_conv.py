#!/usr/bin/env python3
"""Per-layer micro-benchmark of subreg_conv_fwd as the eval-mode backbone calls it (folded scale, fused shortcut).

  python tools/bench_conv.py [--batch 256] [--dtype bf16] [--iters 20]
Prints one line per distinct conv of ResNet18 at 84x84: time (HIP events on the launch stream, random data),
algorithmic TFLOP/s and the fraction of the dense MFMA peak.  GPU only.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))

import torch  # noqa: E402

from subreg_hip import _lib  # noqa: E402

# (name, H, Cin, Cout, ksize, pool, Cin2 (fused shortcut K; -1 none; 0 identity), count per forward)
# Cin = 3 marks the production first layer: conv1 straight from the fp32 NCHW image (conv_first.hip); Cin2 = 3 marks conv3 of
# layer 1 with its 1x1 shortcut fed from the image (conv64_resident.hip IMG kernels).  `--im2col` benches the K = 32 im2col
# route they replaced (plus its pack_input launch) instead.
# (Cin = -3: conv1 + conv2 of layer 1 as the ONE fused launch the backbone runs, conv64_fused_first_kernel; `--unfused` benches
# the two launches it replaced.)
LAYERS = [
    ("L1.conv1+conv2 (fp32 image)", 84, -3, 64, 3, False, -1, 1),
    ("L1.conv3+ds(image)+pool", 84, 64, 64, 3, True, 3, 1),
    ("L2.conv1", 42, 64, 160, 3, False, -1, 1),
    ("L2.conv2", 42, 160, 160, 3, False, -1, 1),
    ("L2.conv3+ds+pool", 42, 160, 160, 3, True, 64, 1),
    ("L3.0.conv1", 21, 160, 320, 3, False, -1, 1),
    ("L3.0.conv2", 21, 320, 320, 3, False, -1, 1),
    ("L3.0.conv3+ds+pool", 21, 320, 320, 3, True, 160, 1),
    ("L3.1.conv1/2", 10, 320, 320, 3, False, -1, 2),
    ("L3.1.conv3+id", 10, 320, 320, 3, False, 0, 1),
    ("L4.0.conv1", 10, 320, 640, 3, False, -1, 1),
    ("L4.0.conv2", 10, 640, 640, 3, False, -1, 1),
    ("L4.0.conv3+ds+pool", 10, 640, 640, 3, True, 320, 1),
    ("L4.1.conv1/2", 5, 640, 640, 3, False, -1, 2),
    ("L4.1.conv3+id", 5, 640, 640, 3, False, 0, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--im2col", action="store_true", help="layer 1 over the K = 32 im2col buffer (the round-2 route) + pack_input")
    ap.add_argument("--unfused", action="store_true", help="conv1 (from the image) and conv2 of layer 1 as two launches")
    ap.add_argument("--kernel", default="auto", choices=["auto", "general", "wide", "wide_alt", "wide128", "wide256"],
                    help="wide layers (Cout % 160 == 0): the dispatcher's rule, conv_fwd.hip forced, conv_wide.hip forced")
    ap.add_argument("--data", default="normal", choices=["normal", "zeros", "narrow", "half"],
                    help="operand values of the general layers (power experiment): N(0, 1) activations and N(0, 1/K) weights; all zeros; "
                         "the probes' distribution (random sign, magnitude uniform in [1, 2)); normal with a random half of the activations zero")
    a = ap.parse_args()

    def shaped(t):
        if a.data == "zeros":
            return torch.zeros_like(t)
        if a.data == "narrow":
            return ((torch.rand_like(t.float()) + 1.0) * torch.where(torch.rand_like(t.float()) < 0.5, -1.0, 1.0)).to(t.dtype)
        return t
    lib = _lib.load()
    dev = torch.device("cuda:0")
    dt = _lib.dtype_code(a.dtype)
    td = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    peak = 2500.0 if a.dtype == "bf16" else 157.3
    B = a.batch
    tot_t, tot_f = 0.0, 0.0
    layers = list(LAYERS)
    if a.im2col or a.dtype != "bf16":
        layers[0:2] = [("pack_input (im2col rows)", 84, 0, 32, 0, False, -1, 1), ("L1.conv1 (K=32 im2col)", 84, 32, 64, 1, False, -1, 1),
                       ("L1.conv2", 84, 64, 64, 3, False, -1, 1), ("L1.conv3+ds+pool", 84, 64, 64, 3, True, 32, 1)]
    elif a.unfused:
        layers[0:1] = [("L1.conv1 (fp32 image)", 84, 3, 64, 3, False, -1, 1), ("L1.conv2", 84, 64, 64, 3, False, -1, 1)]
    for name, H, Cin, Cout, k, pool, cin2, count in layers:
        if a.only and a.only not in name:
            continue
        npix = B * H * H
        if Cin == 0 or Cin == 3 or Cin == -3 or cin2 == 3:          # the first layer's special kernels
            img = torch.randn(B, 3, H, H, device=dev)
            w1 = (torch.randn(64, 32, device=dev) / 27 ** 0.5).to(td)
            shift = torch.randn(64, device=dev)
            if Cin == 0:
                out = torch.empty(npix, 32, device=dev, dtype=td)
                run = lambda: _lib.check(lib.subreg_pack_input(_lib.ptr(img), _lib.ptr(out), B, H, H, dt, _lib.stream_ptr()))   # noqa: E731
                flops, mrows, kk = 0.0, npix, 0
            elif Cin == -3:
                w2 = (torch.randn(64, 9, 64, device=dev) / 576 ** 0.5).to(td)
                sh2 = torch.randn(64, device=dev)
                out = torch.empty(npix, 64, device=dev, dtype=td)
                run = lambda: _lib.check(lib.subreg_conv12_first_fused(_lib.ptr(img), _lib.ptr(w1), _lib.ptr(shift), _lib.ptr(w2), _lib.ptr(sh2),   # noqa: E731
                                                                       _lib.ptr(out), B, H, H,
                                                                       _lib.CONV_LRELU | (_lib.CONV_KERNEL_WIDE if a.kernel == "wide" else 0), dt, _lib.stream_ptr()))
                flops, mrows, kk = 2.0 * npix * 64 * (27 + 576), npix, 603
            elif Cin == 3:
                out = torch.empty(npix, 64, device=dev, dtype=td)
                run = lambda: _lib.check(lib.subreg_conv_first_fwd(_lib.ptr(img), _lib.ptr(w1), _lib.ptr(out), _lib.ptr(shift), B, H, H, 64,   # noqa: E731
                                                                   _lib.CONV_LRELU, dt, _lib.stream_ptr()))
                flops, mrows, kk = 2.0 * npix * 64 * 27, npix, 27
            else:
                x3 = torch.randn(npix, 64, device=dev).to(td)
                w3 = (torch.randn(64, 9, 64, device=dev) / 576 ** 0.5).to(td)
                out = torch.empty(B * (H // 2) ** 2, 64, device=dev, dtype=td)
                run = lambda: _lib.check(lib.subreg_conv_fwd_image_shortcut(_lib.ptr(x3), _lib.ptr(w3), _lib.ptr(out), _lib.ptr(shift), _lib.ptr(img),   # noqa: E731
                                                                            _lib.ptr(w1), B, H, H, 64, 64, _lib.CONV_LRELU | _lib.CONV_POOL2, dt,
                                                                            _lib.stream_ptr()))
                flops, mrows, kk = 2.0 * npix * 64 * 576 + 2.0 * npix * 64 * 3, npix, 576
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            tf = flops / us * 1e-6
            tot_t += us * count
            tot_f += flops * count
            print("%-24s M=%8d K=%5d N=%4d  %8.1f us  %7.1f TFLOP/s  %5.1f%% of peak" % (name, mrows, kk, Cout, us, tf, 100 * tf / peak))
            continue
        x = shaped(torch.randn(npix, Cin, device=dev).to(td))
        if a.data == "half":
            x = x * (torch.rand(npix, Cin, device=dev) < 0.5).to(td)
        w = shaped((torch.randn(Cout, k * k, Cin, device=dev) / (Cin * k * k) ** 0.5).to(td))
        shift = torch.randn(Cout, device=dev)
        Ho = H // 2 if pool else H
        y = torch.empty(B * Ho * Ho, Cout, device=dev, dtype=td)
        x2 = w2 = None
        c2 = 0
        if cin2 >= 0:
            c2 = Cout if cin2 == 0 else cin2
            x2 = shaped(torch.randn(npix, c2, device=dev).to(td))
            w2 = shaped((torch.eye(Cout, device=dev) if cin2 == 0 else torch.randn(Cout, c2, device=dev) / c2 ** 0.5).to(td)).contiguous()
        flags = _lib.CONV_LRELU | (_lib.CONV_POOL2 if pool else 0)
        if Cout % 160 == 0 and a.dtype == "bf16":
            flags |= {"auto": 0, "general": _lib.CONV_KERNEL_GENERAL, "wide": _lib.CONV_KERNEL_WIDE,
                      "wide_alt": _lib.CONV_KERNEL_WIDE | _lib.CONV_KERNEL_WIDE_ALT, "wide128": _lib.CONV_KERNEL_WIDE | _lib.CONV_KERNEL_WIDE_128,
                      "wide256": _lib.CONV_KERNEL_WIDE | _lib.CONV_KERNEL_WIDE_256}[a.kernel]

        def run():
            _lib.check(lib.subreg_conv_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), None, _lib.ptr(shift), None, None,
                                           _lib.ptr(x2), _lib.ptr(w2), c2, B, H, H, Cin, Cout, k, flags, dt, _lib.stream_ptr()))
        if lib.subreg_conv_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), None, _lib.ptr(shift), None, None, _lib.ptr(x2), _lib.ptr(w2), c2, B, H, H,
                               Cin, Cout, k, flags, dt, _lib.stream_ptr()) == -2:            # SUBREG_EUNSUPPORTED: a forced kernel that does not take the layer
            print("%-24s M=%8d K=%5d N=%4d  (the forced kernel does not take this problem)" % (name, npix, Cin * k * k, Cout))
            continue
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        mrows = B * Ho * Ho * 4 if pool else npix                 # rows actually computed (floor pooling skips the odd edge)
        flops = 2.0 * npix * Cout * Cin * k * k + (2.0 * npix * Cout * c2 if (cin2 > 0) else 0.0)   # algorithmic (identity adds none)
        tf = flops / us * 1e-6
        tot_t += us * count
        tot_f += flops * count
        print("%-24s M=%8d K=%5d N=%4d  %8.1f us  %7.1f TFLOP/s  %5.1f%% of peak" % (name, mrows, Cin * k * k, Cout, us, tf, 100 * tf / peak))
    if not a.only:
        print("conv stack, B=%d: %.1f us, %.1f TFLOP/s algorithmic (%.1f%% of %s peak)" % (B, tot_t, tot_f / tot_t * 1e-6, 100 * tot_f / tot_t * 1e-6 / peak, a.dtype))


if __name__ == "__main__":
    main()
