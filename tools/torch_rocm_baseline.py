#!/usr/bin/env python3
"""The reference's route on this GPU: the same network as plain PyTorch-ROCm modules (nn.Conv2d / BatchNorm2d / LeakyReLU / MaxPool2d
through MIOpen and rocBLAS), i.e. what `python eval_incremental.py` / `train_supervised.py` of the reference executes when it is
simply run on an MI355X.  Context for bench.py's numbers, never a target and not part of the product path (nothing here touches
subreg_hip or oracle/).  The architecture is restated from SURVEY.md section 3 (RFS ResNet-12 family "resnet18": widths 64 / 160 / 320 /
640, blocks 1 / 1 / 2 / 2, three bias-free 3x3 convs per block + BN + LeakyReLU(0.1), 1x1 conv + BN shortcut in the first block of a
layer, MaxPool2d(2) after it, global average pool, Linear(640, n_cls, bias=False)); random weights, synthetic 84x84 images.
  python tools/torch_rocm_baseline.py"""
import sys
import time

import torch
import torch.nn as nn


class Block(nn.Module):
    def __init__(self, cin, cout, first):
        super().__init__()
        self.c1, self.b1 = nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)
        self.c2, self.b2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)
        self.c3, self.b3 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)
        self.act = nn.LeakyReLU(0.1)
        self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, 1, bias=False), nn.BatchNorm2d(cout)) if first else None
        self.pool = nn.MaxPool2d(2) if first else None

    def forward(self, x):
        o = self.act(self.b1(self.c1(x)))
        o = self.act(self.b2(self.c2(o)))
        o = self.b3(self.c3(o))
        o = self.act(o + (x if self.down is None else self.down(x)))
        return o if self.pool is None else self.pool(o)


class Net(nn.Module):
    def __init__(self, n_cls=60):
        super().__init__()
        layers, cin = [], 3
        for width, nb in zip((64, 160, 320, 640), (1, 1, 2, 2)):
            for bi in range(nb):
                layers.append(Block(cin if bi == 0 else width, width, bi == 0))
            cin = width
        self.body = nn.Sequential(*layers)
        self.classifier = nn.Linear(640, n_cls, bias=False)

    def features(self, x):
        return self.body(x).mean((2, 3))

    def forward(self, x):
        return self.classifier(self.features(x))


def timed(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda:0")
    # MIOpen in its default immediate mode, or - `--find`, what the reference asks for (eval_incremental.py:114 cudnn.benchmark = True) -
    # in find mode, which on a fresh box without a performance database compiles and times candidate kernels for minutes PER SHAPE
    # (`--quick`: the batch-700 eval forward only; `--train-only`: the 64-image training step only)
    find, quick, train_only = "--find" in sys.argv, "--quick" in sys.argv, "--train-only" in sys.argv
    torch.backends.cudnn.benchmark = find
    print("MIOpen mode:", "find (cudnn.benchmark = True)" if find else "immediate (default)", flush=True)
    print("torch", torch.__version__, "| 8.1219 GFLOP per 84x84 image forward, 24.339 per training image; 69875 image-forwards per episode (bench.py)")
    for mode, dt in (("fp32 (what the reference runs)", None), ("bf16 autocast, channels_last", torch.bfloat16)):
        net = Net().to(dev).eval()
        if dt is not None:
            net = net.to(memory_format=torch.channels_last)
        for B in (() if train_only else (700,) if quick else (125, 700, 1125)):
            x = torch.randn(B, 3, 84, 84, device=dev)
            if dt is not None:
                x = x.contiguous(memory_format=torch.channels_last)

            def fwd():
                with torch.no_grad(), torch.autocast("cuda", dtype=dt, enabled=dt is not None):
                    return net.features(x)
            t0 = time.perf_counter()
            fwd()
            torch.cuda.synchronize()
            print("  (first call, incl. MIOpen's kernel selection: %.0f s)" % (time.perf_counter() - t0), flush=True)
            t = timed(fwd, 10, 4)
            print("eval forward  %-32s B=%4d  %8.2f ms  %8.0f img/s  %7.1f TFLOP/s  -> %.3f episodes/s at 100 epochs" %
                  (mode, B, t * 1e3, B / t, B * 8.1219e9 / t * 1e-12, B / t / 69875.0))
        if quick:
            continue
        net.train()
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
        crit = nn.CrossEntropyLoss()
        for B in ((64,) if train_only else (64, 128)):
            x = torch.randn(B, 3, 84, 84, device=dev)
            if dt is not None:
                x = x.contiguous(memory_format=torch.channels_last)
            y = torch.randint(0, 60, (B,), device=dev)

            def step():
                with torch.autocast("cuda", dtype=dt, enabled=dt is not None):
                    loss = crit(net(x), y)
                opt.zero_grad()
                loss.backward()
                opt.step()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            print("  (first step, incl. MIOpen's kernel selection: %.0f s)" % (time.perf_counter() - t0), flush=True)
            t = timed(step, 10, 4)
            print("train step    %-32s B=%4d  %8.2f ms  %8.0f img/s  %7.1f TFLOP/s" % (mode, B, t * 1e3, B / t, B * 24.339e9 / t * 1e-12))
        del net, opt


if __name__ == "__main__":
    main()
