#!/usr/bin/env python3
"""In-kernel phase breakdown of conv64_resident (a -DR64_DIAG=1 build of the library: SUBREG_LIB=...).
Median cycles per tile and wave in: address arithmetic, chunk 0, mid barrier, chunk 1, epilogue, DMA wait, end barrier."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from subreg_hip import _lib   # noqa: E402


def main():
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for pool, sc in ((False, False), (True, True)):
        H = 84
        x = torch.randn(B * H * H, 64, device=dev).to(torch.bfloat16)
        w = (torch.randn(64, 9, 64, device=dev) / 24).to(torch.bfloat16)
        shift = torch.randn(64, device=dev)
        Ho = H // 2 if pool else H
        y = torch.empty(B * Ho * Ho, 64, device=dev, dtype=torch.bfloat16)
        x2 = torch.randn(B * H * H, 32, device=dev).to(torch.bfloat16) if sc else None
        w2 = (torch.randn(64, 32, device=dev) / 6).to(torch.bfloat16) if sc else None
        flags = _lib.CONV_LRELU | (_lib.CONV_POOL2 if pool else 0)
        for _ in range(3):
            _lib.check(lib.subreg_conv_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), None, _lib.ptr(shift), None, None, _lib.ptr(x2),
                                           _lib.ptr(w2), 32 if sc else 0, B, H, H, 64, 64, 3, flags, _lib.BF16, _lib.stream_ptr()))
        torch.cuda.synchronize()
        out = np.zeros(4096 * 8, np.float32)
        assert raw.subreg_r64_diag_read(out.ctypes.data_as(C.c_void_p), out.size) == 0
        d = out.reshape(-1, 8)
        d = d[d[:, 0] > 0]
        per = d[:, 1:] / d[:, :1]
        names = ["addr", "chunk0", "bar1", "chunk1", "epilogue", "dma wait", "bar2"]
        med = np.median(per, axis=0)
        print("pool=%d sc=%d B=%d: waves %d, tiles/wave %.1f, cycles per tile %.0f = " % (pool, sc, B, len(d), np.median(d[:, 0]), med.sum()) +
              ", ".join("%s %.0f" % (n, v) for n, v in zip(names, med)))


if __name__ == "__main__":
    main()
