import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "subspace-reg_amd"), os.path.join(REPO, "tests")]
from test_hip_kernels import _nhwc_dev, _pack_w, _t, _nchw_host, _dev
from oracle import resnet_ref as rr
from subreg_hip import _lib
lib = _lib.load()
for (B, H, W, Cin, Cout, c2, use_res) in [(6, 10, 10, 640, 320, 640, False), (6, 10, 10, 320, 320, 0, True), (6, 10, 10, 320, 320, 0, False)]:
    rs = np.random.RandomState(1)
    x = rs.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rs.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    want = rr.conv_nhwc(rr._nhwc(x).astype(np.float64), w.astype(np.float64))
    xd, wd = _nhwc_dev(x, "f32"), _pack_w(w, "f32")
    x2d = w2d = rd = None
    if c2:
        x2 = rs.standard_normal((B, c2, H, W)).astype(np.float32)
        w2 = (rs.standard_normal((Cout, c2, 1, 1)) / np.sqrt(c2)).astype(np.float32)
        want = want + rr.conv_nhwc(rr._nhwc(x2).astype(np.float64), w2.astype(np.float64))
        x2d, w2d = _nhwc_dev(x2, "f32"), _pack_w(w2, "f32")
    if use_res:
        r = rs.standard_normal((B, Cout, H, W)).astype(np.float32)
        want = want + rr._nhwc(r)
        rd = _nhwc_dev(r, "f32")
    zero = torch.zeros(Cout, device=_dev())
    y = torch.empty(B * H * W * Cout, device=_dev())
    _lib.check(lib.subreg_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), None, _lib.ptr(zero), _lib.ptr(rd), None, _lib.ptr(x2d), _lib.ptr(w2d), c2,
                                   B, H, W, Cin, Cout, 3, 0, 0, _lib.stream_ptr()))
    got = rr._nhwc(_nchw_host(y, B, Cout, H, W, "f32"))
    err = np.abs(got - want)
    print((B, H, W, Cin, Cout, c2, use_res), "max err %.3e  max|want| %.2f  n(err>1e-4) %d" % (err.max(), np.abs(want).max(), (err > 1e-4).sum()))
    bad = np.argwhere(err > 1e-4)
    if len(bad):
        print("  bad rows (b,h,w) sample:", sorted(set((int(a), int(b), int(c)) for a, b, c, d in bad))[:12], " channels:", sorted(set(int(d) for *_, d in bad))[:8])
