#!/usr/bin/env python3
"""Achieved HBM bandwidth of the element-wise / BatchNorm / packing kernels (the part of the path whose roofline is HBM, not
MFMA), through the C ABI, at the shapes the pretraining step (B = 64, train_supervised.py:205-268) and an evaluation forward
use them on.  One line per (kernel call, layer shape): time from HIP events on the launch stream, ALGORITHMIC bytes (every
operand read once, every result written once) and the fraction of the HBM SPEC rate (8 TB/s, MI355X_MICROARCH.md); the box's own
streaming rates (tools/probes/stream_bw.hip: read ~6.3, write ~4.7, copy ~4.8 TB/s) are the practical ceiling of a read + write
pass.  A row whose tensors fit the 256 MiB Infinity Cache between two launches of the timing loop is marked `L3`: its bytes
never reach HBM, so its rate is NOT an HBM figure (such rows can read above the copy rate).

  python tools/bench_elementwise.py [--batch 64] [--iters 30]
"""
import argparse
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))

import torch  # noqa: E402

from subreg_hip import _lib  # noqa: E402

# (layer, H = W of the block's input, C, pooled)
BLOCKS = [("layer1.0", 84, 64, True), ("layer2.0", 42, 160, True), ("layer3.0", 21, 320, True), ("layer3.1", 10, 320, False),
          ("layer4.0", 10, 640, True), ("layer4.1", 5, 640, False)]
SPEC = 8.0    # TB/s, HBM3E spec (the roofline's peak); a read + write stream sustains ~4.8 on these boxes (stream_bw copy)
L3_BYTES = 256 * 2 ** 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, dt, bf = a.batch, _lib.BF16, torch.bfloat16
    s = _lib.stream_ptr

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

    tot = {}

    def line(kind, name, us, nbytes):
        tbs = nbytes / us * 1e-6
        k = tot.setdefault(kind, [0.0, 0.0])
        k[0] += us
        k[1] += nbytes
        print("%-34s %-9s %8.1f us  %8.1f MB  %5.2f TB/s  %3.0f%% of %.0f (spec)%s" %
              (kind, name, us, nbytes * 1e-6, tbs, 100 * tbs / SPEC, SPEC, "  L3" if nbytes < L3_BYTES else ""))

    print("batch %d, bf16; bytes are algorithmic (operands read once, results written once)" % B)
    for name, H, Cc, pool in BLOCKS:
        npix = B * H * H
        n = npix * Cc
        Ho = H // 2 if pool else H
        nout = B * Ho * Ho * Cc
        raw, act, res, dy, dx = (torch.randn(n, device=dev).to(bf) for _ in range(5))
        out = torch.empty(nout, device=dev, dtype=bf)
        gout = torch.randn(nout, device=dev).to(bf)
        keep = (torch.rand(nout, device=dev) > 0.1).to(torch.uint8)
        sc, sh, mean, invstd, gamma = (torch.rand(Cc, device=dev) + 0.5 for _ in range(5))
        dgam, dbet = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        partial = torch.empty(lib.subreg_bn_bwd_slices(npix) * Cc * 2, dtype=torch.float64, device=dev)
        P = _lib.ptr
        # train-mode forward, second pass of bn1 / bn2: act = lrelu(raw * scale + shift)
        us = t_of(lambda: _lib.check(lib.subreg_bn_apply(P(raw), P(sc), P(sh), None, None, None, None, 1.0, None, P(act), B, H, H, Cc,
                                                         _lib.CONV_LRELU, dt, s())))
        line("bn_apply (bn1/bn2 + LeakyReLU)", name, us, 2 * n * 2)
        # block tail: out = keep * pool(lrelu(bn3(raw3) + bn_ds(res)))
        fl = _lib.CONV_LRELU | (_lib.CONV_POOL2 if pool else 0)
        us = t_of(lambda: _lib.check(lib.subreg_bn_apply(P(raw), P(sc), P(sh), P(res), P(sc), P(sh), P(keep), 1.11, None, P(out), B, H, H, Cc,
                                                         fl, dt, s())))
        line("bn_apply (block tail, +res, pool)", name, us, 2 * n * 2 + nout * 3)
        # BN backward (reduce + finalize + apply), with and without the fused LeakyReLU'
        for with_act in (True, False):
            us = t_of(lambda: _lib.check(lib.subreg_bn_bwd(P(dy), P(act) if with_act else None, P(raw), P(mean), P(invstd), P(gamma),
                                                           P(partial), P(dgam), P(dbet), P(dx), npix, Cc, dt, s())))
            nt = 3 if with_act else 2
            line("bn_bwd (reduce+finalize+apply)%s" % (" +act" if with_act else ""), name, us, (2 * nt + 1) * n * 2)
        # block tail backward
        dv = torch.empty(n, device=dev, dtype=bf)
        us = t_of(lambda: _lib.check(lib.subreg_block_tail_bwd(P(gout), P(keep), 1.11, None, P(raw), P(sc), P(sh), P(res), P(sc), P(sh), P(dv), B, H,
                                                               H, Cc, 1 if pool else 0, dt, s())))
        line("block_tail_bwd", name, us, nout * 3 + 3 * n * 2)
    # first-layer packing and the global average pool
    for Bv in sorted({B, 700}):
        img = torch.randn(Bv, 3, 84, 84, device=dev)
        col = torch.empty(Bv * 84 * 84 * 32, device=dev, dtype=bf)
        us = t_of(lambda: _lib.check(lib.subreg_pack_input(_lib.ptr(img), _lib.ptr(col), Bv, 84, 84, dt, s())))
        line("pack_input (im2col route only)", "B=%d" % Bv, us, Bv * 84 * 84 * 76)
        x = torch.randn(Bv * 25 * 640, device=dev).to(bf)
        feat = torch.empty(Bv, 640, device=dev)
        us = t_of(lambda: _lib.check(lib.subreg_avgpool(_lib.ptr(x), _lib.ptr(feat), Bv, 5, 5, 640, dt, s())))
        line("avgpool", "B=%d" % Bv, us, Bv * 25 * 640 * 2 + Bv * 640 * 4)
    print()
    for kind, (us, nb) in tot.items():
        print("%-34s total     %8.1f us  %8.1f MB  %5.2f TB/s" % (kind, us, nb * 1e-6, nb / us * 1e-6))


if __name__ == "__main__":
    main()
