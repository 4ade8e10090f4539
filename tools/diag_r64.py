#!/usr/bin/env python3
"""In-kernel phase breakdown of conv64_resident (a -DR64_DIAG=1 build of the library: SUBREG_LIB=...).
Median cycles per tile and wave in: address arithmetic, chunk 0, mid barrier, chunk 1, epilogue, DMA wait, end barrier
(8-wave kernel; SUBREG_NO_WIDE64=1 sends the un-pooled case there too) or the two half-phases of conv64_wide_kernel."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from subreg_hip import _lib   # noqa: E402


def fused(lib, raw, dev, B):
    """conv64_fused_first_kernel (layer 1 conv1 -> conv2): cycles per tile and wave by phase, waves 0-3 and 4-7 apart (the two groups run
    conv1 of the next tile at opposite ends of the tile)."""
    H = 84
    img = torch.randn(B, 3, H, H, device=dev)
    w1 = (torch.randn(64, 32, device=dev) / 5).to(torch.bfloat16)
    w2 = (torch.randn(64, 9, 64, device=dev) / 24).to(torch.bfloat16)
    sh1, sh2 = torch.randn(64, device=dev), torch.randn(64, device=dev)
    y = torch.empty(B * H * H, 64, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        _lib.check(lib.subreg_conv12_first_fused(_lib.ptr(img), _lib.ptr(w1), _lib.ptr(sh1), _lib.ptr(w2), _lib.ptr(sh2), _lib.ptr(y), B, H, H,
                                                 _lib.CONV_LRELU, _lib.BF16, _lib.stream_ptr()))
    torch.cuda.synchronize()
    out = np.zeros(4096 * 12, np.float32)
    assert raw.subreg_r64_diag_read(out.ctypes.data_as(C.c_void_p), out.size) == 0
    d = out.reshape(-1, 8, 12)
    d = d[d[:, 0, 0] > 0]
    names = ["conv1 (first)", "chunk 0", "chunk 1", "epilogue", "conv1 (after)", "DMA wait + barrier", "convert + barrier"]
    if os.environ.get("SUBREG_L1_WIDE_FUSED") == "1":       # conv64_wide_fused_kernel: four waves, its own phases
        g = d[:, 0:4, :].reshape(-1, 12)
        per = np.median(g[:, 1:6] / g[:, :1], axis=0)
        wn = ["roll copy + tile set-up", "phase 0 (row tile 0 + conv1 group)", "phase 1", "LDS / DMA wait", "patch conversion + barrier"]
        print("wide fused conv1+conv2 B=%d: tiles/wave %.1f, clock %.2f GHz, cycles per tile %.0f = " % (B, np.median(g[:, 0]), np.median(g[:, 11]), per.sum()) +
              ", ".join("%s %.0f" % (n, v) for n, v in zip(wn, per)))
        return
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        g = d[:, sl, :].reshape(-1, 12)
        per = np.median(g[:, 1:8] / g[:, :1], axis=0)
        print("fused conv1+conv2 B=%d %s: tiles/wave %.1f, clock %.2f GHz, cycles per tile %.0f = " % (B, grp, np.median(g[:, 0]), np.median(g[:, 11]), per.sum()) +
              ", ".join("%s %.0f" % (n, v) for n, v in zip(names, per)))


def conv3(lib, raw, dev, B):
    """conv64_pool_img_kernel (layer 1 conv3 + image shortcut + pool): cycles per tile and wave by phase, waves 0-3 and 4-7 apart."""
    H = 84
    img = torch.randn(B, 3, H, H, device=dev)
    x = torch.randn(B * H * H, 64, device=dev).to(torch.bfloat16)
    w = (torch.randn(64, 9, 64, device=dev) / 24).to(torch.bfloat16)
    w2 = (torch.randn(64, 32, device=dev) / 6).to(torch.bfloat16)
    shift = torch.randn(64, device=dev)
    y = torch.empty(B * (H // 2) * (H // 2), 64, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        _lib.check(lib.subreg_conv_fwd_image_shortcut(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(shift), _lib.ptr(img), _lib.ptr(w2), B, H, H, 64, 64,
                                                      _lib.CONV_LRELU | _lib.CONV_POOL2, _lib.BF16, _lib.stream_ptr()))
    torch.cuda.synchronize()
    out = np.zeros(4096 * 12, np.float32)
    assert raw.subreg_r64_diag_read(out.ctypes.data_as(C.c_void_p), out.size) == 0
    d = out.reshape(-1, 8, 12)
    d = d[d[:, 0, 0] > 0]
    names = ["staging (first) + addresses", "chunk 0", "chunk 1", "staging (after)", "epilogue", "DMA wait", "barrier"]
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        g = d[:, sl, :].reshape(-1, 12)
        per = np.median(g[:, 1:8] / g[:, :1], axis=0)
        print("conv3 + image shortcut + pool B=%d %s: tiles/wave %.1f, clock %.2f GHz, cycles per tile %.0f = " % (B, grp, np.median(g[:, 0]), np.median(g[:, 11]), per.sum()) +
              ", ".join("%s %.0f" % (n, v) for n, v in zip(names, per)))


def main():
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    if "--fused" in sys.argv:
        return fused(lib, raw, dev, B)
    if "--conv3" in sys.argv:
        return conv3(lib, raw, dev, B)
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
        out = np.zeros(4096 * 12, np.float32)
        assert raw.subreg_r64_diag_read(out.ctypes.data_as(C.c_void_p), out.size) == 0
        d = out.reshape(-1, 12)
        nw = int((d[:, 0] > 0).sum())
        tot_all = (d[:nw, 1:7].sum(axis=1) + d[:nw, 8])            # prologue + loop cycles of every wave (workgroup-major, 8 waves each)
        wg = tot_all.reshape(-1, 8).max(axis=1)
        print("  kernel-resident cycles per workgroup: min %.0f median %.0f max %.0f; by XCD (blockIdx %% 8) max: %s" %
              (wg.min(), np.median(wg), wg.max(), " ".join("%.0f" % wg[x::8].max() for x in range(8))))
        d = d[d[:, 0] > 0]
        print("  wave resident time (s_memrealtime): median %.1f us, max %.1f us" % (np.median(d[:, 10]) / 100.0, d[:, 10].max() / 100.0))
        per = d[:, 1:7] / d[:, :1]
        names = ["setup+dma issue", "chunk0", "bar1", "chunk1", "epilogue", "dma wait+bar2"]
        if not pool and os.environ.get("SUBREG_NO_WIDE64") != "1":     # conv64_wide_kernel (4 waves per workgroup) stamps its own phases
            names = ["set-up", "phase A (row tile 0 + fillers)", "-", "phase B (row tile 1 + fillers)", "DMA wait", "barrier"]
        med = np.median(per, axis=0)
        print("pool=%d sc=%d B=%d: waves %d, tiles/wave %.1f, shader clock %.2f GHz, cycles per tile %.0f = " %
              (pool, sc, B, len(d), np.median(d[:, 0]), np.median(d[:, 7]), med.sum()) + ", ".join("%s %.0f" % (n, v) for n, v in zip(names, med)) +
              "; prologue %.0f cycles; workgroup start spread %.1f us" % (np.median(d[:, 8]), (d[:, 9].max() - d[:, 9].min()) / 100.0))


if __name__ == "__main__":
    main()
