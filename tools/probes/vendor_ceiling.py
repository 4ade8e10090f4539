#!/usr/bin/env python3
"""What the vendor's own libraries reach on this box (context for the roofline fraction; NOT part of the product path): a large bf16
GEMM through torch.matmul (hipBLASLt / rocBLAS) on N(0, 1) data and on zeros, and MIOpen's convolution on the shapes of the
backbone's widest layers (torch.nn.functional.conv2d, channels_last, bf16, eval-mode 3x3, stride 1, pad 1), timed with HIP
events over 20 launches after 5 of warm-up.  Peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")


def timed(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    print("torch", torch.__version__, torch.cuda.get_device_name(0))
    for (m, n, k) in ((8192, 8192, 8192), (16384, 8192, 4096), (308700, 320, 2880)):
        for kind in ("normal", "zeros"):
            a = torch.randn(m, k, device=dev).to(torch.bfloat16) if kind == "normal" else torch.zeros(m, k, device=dev, dtype=torch.bfloat16)
            b = torch.randn(k, n, device=dev).to(torch.bfloat16) if kind == "normal" else torch.zeros(k, n, device=dev, dtype=torch.bfloat16)
            t = timed(lambda: torch.matmul(a, b))
            tf = 2.0 * m * n * k / t * 1e-12
            print("matmul bf16 %6d x %5d x %5d  %-6s  %8.1f us  %7.1f TFLOP/s  %.3f of 2500" % (m, n, k, kind, t * 1e6, tf, tf / 2500))
            del a, b
    for name, B, C, H, K in (("L2.conv2", 700, 160, 42, 160), ("L3.0.conv2", 700, 320, 21, 320), ("L4.0.conv2", 700, 640, 10, 640), ("L4.1.conv1/2", 700, 640, 5, 640)):
        x = torch.randn(B, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        try:
            t = timed(lambda: F.conv2d(x, w, None, 1, 1))
            tf = 2.0 * B * H * H * K * C * 9 / t * 1e-12
            print("MIOpen conv2d bf16 NHWC %-12s B=%d  %8.1f us  %7.1f TFLOP/s  %.3f of 2500" % (name, B, t * 1e6, tf, tf / 2500))
        except Exception as e:      # noqa: BLE001
            print("MIOpen conv2d %s failed: %s" % (name, str(e)[:200]))


if __name__ == "__main__":
    main()
