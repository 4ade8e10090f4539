#!/usr/bin/env python3
"""Where does a conv launch's time step up as the tile count grows?  One workgroup per 128 x 160 (or 256 x 160) tile: a step right above
512 tiles means two workgroups per CU are resident, above 768 three.  Layer 4.1's shape (5x5, 640 -> 640) or `H Cin Cout` from the command
line (batches scaled to the same tile counts), kernels forced.  Result: profiles/r06_occupancy_steps.txt."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))
import torch  # noqa: E402

from subreg_hip import _lib  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    H, Cin, Cout, k = 5, 640, 640, 3
    if len(sys.argv) >= 4:
        H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    scale = (25 * 640) / (H * H * Cout)                       # same tile counts as the default shape
    dt, td = _lib.dtype_code("bf16"), torch.bfloat16
    kernels = {"general": _lib.CONV_KERNEL_GENERAL, "wide128": _lib.CONV_KERNEL_WIDE | _lib.CONV_KERNEL_WIDE_128,
               "wide256": _lib.CONV_KERNEL_WIDE | _lib.CONV_KERNEL_WIDE_256}
    print("# %dx%d maps, %d -> %d channels" % (H, H, Cin, Cout))
    print("%6s %8s" % ("B", "M") + "".join("%22s" % n for n in kernels))
    for B0 in (160, 320, 480, 600, 640, 655, 660, 700, 800, 900, 975, 985, 1000, 1100, 1300):
        B = max(1, int(B0 * scale))
        npix = B * H * H
        x = torch.randn(npix, Cin, device=dev).to(td)
        w = (torch.randn(Cout, k * k, Cin, device=dev) / (Cin * k * k) ** 0.5).to(td)
        shift = torch.randn(Cout, device=dev)
        y = torch.empty(npix, Cout, device=dev, dtype=td)
        row = "%6d %8d" % (B, npix)
        for name, fl in kernels.items():
            flags = _lib.CONV_LRELU | fl

            def run():
                return lib.subreg_conv_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), None, _lib.ptr(shift), None, None, None, None, 0, B, H, H,
                                           Cin, Cout, k, flags, dt, _lib.stream_ptr())
            if run() != 0:
                row += "%22s" % "-"
                continue
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            tm = 128 if name != "wide256" else 256
            tiles = ((npix + tm - 1) // tm) * (Cout // 160)
            row += "%12.1f us %4d t" % (e0.elapsed_time(e1) * 50.0, tiles)
        print(row)


if __name__ == "__main__":
    main()
