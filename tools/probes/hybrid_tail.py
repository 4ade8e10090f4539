#!/usr/bin/env python3
"""Probe (round 6): what would running the LAST, partial round of a wide conv's tiles as 128-row tiles buy?

A launch of T 256-row tiles on 512 slots (256 CUs x 2 workgroups) ends with T % 512 tiles on a chip that is otherwise idle.  Emulated here with
two launches side by side on two streams: the images that fill whole rounds on the 256-row tiling, the remaining images on the 128-row tiling
(conv_wide16_kernel<NA = 2>), against ONE launch of all images on the 256-row tiling and against the dispatcher's own choice.
Usage: hybrid_tail.py [H Cin Cout]   (default: layer 4.0 conv2, 10x10, 640 -> 640)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))
import torch  # noqa: E402

from subreg_hip import _lib  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    H, Cin, Cout, k = 10, 640, 640, 3
    if len(sys.argv) >= 4:
        H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dt, td = _lib.dtype_code("bf16"), torch.bfloat16
    ntn = Cout // 160
    w = (torch.randn(Cout, k * k, Cin, device=dev) / (Cin * k * k) ** 0.5).to(td)
    shift = torch.randn(Cout, device=dev)
    s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    W256, W128, AUTO = _lib.CONV_KERNEL_WIDE | _lib.CONV_KERNEL_WIDE_256, _lib.CONV_KERNEL_WIDE | _lib.CONV_KERNEL_WIDE_128, 0

    def conv(x, y, B, flags, stream):
        with torch.cuda.stream(stream):
            rc = lib.subreg_conv_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), None, _lib.ptr(shift), None, None, None, None, 0, B, H, H,
                                     Cin, Cout, k, _lib.CONV_LRELU | flags, dt, _lib.stream_ptr())
        assert rc == 0, rc

    print("# %dx%d maps, %d -> %d channels; ms per launch (20 launches); tiles = 256-row tiles of the whole batch" % (H, H, Cin, Cout))
    print("%6s %7s %10s %10s %22s" % ("B", "tiles", "auto", "wide256", "rounds@256 + tail@128"))
    for B in (175, 250, 350, 500, 563, 700, 1000, 1125):
        npix = B * H * H
        x = torch.randn(npix, Cin, device=dev).to(td)
        y = torch.empty(npix, Cout, device=dev, dtype=td)
        mt = (npix + 255) // 256
        tiles = mt * ntn
        full_mt = (tiles // 512) * 512 // ntn                   # m-tiles of the whole rounds
        B1 = min(B, (full_mt * 256) // (H * H))                 # images whose rows fill them (rounded down to whole images)
        B2 = B - B1

        def timed(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20.0

        cur = torch.cuda.current_stream()

        def hybrid():
            s_main.wait_stream(cur)
            s_side.wait_stream(cur)
            if B1 > 0:
                conv(x[:B1 * H * H], y[:B1 * H * H], B1, W256, s_main)
            if B2 > 0:
                conv(x[B1 * H * H:], y[B1 * H * H:], B2, W128 if B1 > 0 else W256, s_side)
            cur.wait_stream(s_main)
            cur.wait_stream(s_side)

        t_auto = timed(lambda: conv(x, y, B, AUTO, cur))
        t_256 = timed(lambda: conv(x, y, B, W256, cur))
        t_h = timed(hybrid)
        print("%6d %7d %10.4f %10.4f %12.4f  (%d + %d images)" % (B, tiles, t_auto, t_256, t_h, B1, B2))


if __name__ == "__main__":
    main()
