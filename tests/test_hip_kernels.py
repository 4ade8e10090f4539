"""GPU parity tests: the hand-written gfx950 kernels, called THROUGH THE C ABI, against the CPU oracle
(same seeded inputs) and against the golden vectors produced by the reference itself.

Tolerances (written here on purpose):
  f32 path  (v_mfma_f32_32x32x2_f32, the parity mode): 2e-4 abs+rel on activations/features, the
            north_star's 1e-4 on classifier weights and accuracies.
  bf16 path (throughput mode): operands are rounded to 8 bits of mantissa => 3e-2 relative to the tensor's
            max; it is gated on argmax/accuracy level elsewhere.
"""
import ctypes as C
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import resnet_ref as rr                       # noqa: E402  (the checker)
from oracle import subspace_ref as sr                     # noqa: E402
from subreg_hip import _lib, synthetic as syn             # noqa: E402

from conftest import GOLDEN                               # noqa: E402


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev(), dtype)


def _cmp(name, got, want, atol, rtol):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = np.abs(got - want)
    bad = err > atol + rtol * np.abs(want)
    if bad.any():
        i = np.unravel_index(np.argmax(err), err.shape)
        raise AssertionError("%s: %d/%d bad, max err %.4g at %s (got %.6g want %.6g), ref max %.4g" %
                             (name, bad.sum(), bad.size, err.max(), i, got[i], want[i], np.abs(want).max()))


def _tol(dtype, scale=1.0, bf16=1e-2):
    """(atol, rtol): f32 2e-4; bf16 `bf16` of the tensor's magnitude (1e-2 one conv, 3e-2 whole backbone)."""
    return (2e-4 * max(scale, 1.0), 2e-4) if dtype == "f32" else (bf16 * scale, bf16)


def _nhwc_dev(x_nchw, dtype):
    """NCHW fp32 numpy -> device NHWC tensor in the compute dtype through the library's own kernel."""
    lib = _lib.load()
    B, Cc, H, W = x_nchw.shape
    src = _t(x_nchw)
    dst = torch.empty(B * H * W * Cc, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, device=_dev())
    _lib.check(lib.subreg_nchw_to_nhwc(_lib.ptr(src), _lib.ptr(dst), B, Cc, H, W, _lib.dtype_code(dtype), _lib.stream_ptr()))
    return dst


def _nchw_host(t, B, Cc, H, W, dtype):
    lib = _lib.load()
    out = torch.empty(B, Cc, H, W, dtype=torch.float32, device=_dev())
    _lib.check(lib.subreg_nhwc_to_nchw(_lib.ptr(t), _lib.ptr(out), B, Cc, H, W, _lib.dtype_code(dtype), _lib.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _pack_w(w_oihw, dtype, fold=None):
    lib = _lib.load()
    O, I, k, _ = w_oihw.shape
    out = torch.empty(O * k * k * I, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, device=_dev())
    fd = _t(fold) if fold is not None else None
    _lib.check(lib.subreg_pack_conv_weight(_lib.ptr(_t(w_oihw)), _lib.ptr(fd), _lib.ptr(out), O, I, k, 0,
                                           _lib.dtype_code(dtype), _lib.stream_ptr()))
    return out


def _round_bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float32).numpy()


# Kernel selection of subreg_conv_fwd for the wide layers (Cout % 160 == 0, bf16, eval mode): "auto" = the dispatcher's measured rule,
# "general" = conv_fwd.hip forced, "wide" = conv_wide.hip forced in its default MFMA shape (16x16x32: conv_wide16_kernel),
# "wide_alt" = conv_wide.hip in the other shape (32x32x16: conv_wide_kernel), "wide128" / "wide256" = conv_wide16_kernel's 128- / 256-row tiling
# forced (128 rows: skipped where the patch does not fit).  The parity tests below run over all four.
KERNELS = ["auto", "general", "wide", "wide_alt", "wide128", "wide256"]


def _kernel_flag(kernel, dtype, Cout, k=3):
    if kernel == "auto":
        return 0
    if dtype != "bf16" or Cout % 160 != 0 or k != 3:
        pytest.skip("one kernel only for this problem")
    if kernel == "general":
        return _lib.CONV_KERNEL_GENERAL
    return _lib.CONV_KERNEL_WIDE | {"wide": 0, "wide_alt": _lib.CONV_KERNEL_WIDE_ALT, "wide128": _lib.CONV_KERNEL_WIDE_128,
                                    "wide256": _lib.CONV_KERNEL_WIDE_256}[kernel]


def _check_conv(rc, kernel, what):
    """_lib.check, except that the 128-row tiling of conv_wide16_kernel may refuse a problem whose patch does not fit its LDS"""
    if kernel == "wide128" and rc == -2:
        pytest.skip("the 128-row tiling does not take this problem (patch rows)")
    _lib.check(rc, what)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, pool, residual      (shapes of the 14 unique convs, SURVEY.md 8a-3, + ragged/edge ones)
    (2, 84, 84, 64, 64, 3, False, False),
    (1, 84, 84, 64, 64, 3, True, True),
    (2, 42, 42, 64, 160, 3, False, False),
    (2, 42, 42, 160, 160, 3, True, True),
    (2, 42, 42, 64, 160, 1, False, False),
    (3, 21, 21, 160, 320, 3, False, False),
    (3, 21, 21, 320, 320, 3, True, True),          # 21 -> 10 floor pooling drops the last row/col
    (2, 21, 21, 160, 320, 1, False, False),
    (5, 10, 10, 320, 320, 3, False, True),         # identity shortcut, no pool (layer3.1)
    (5, 10, 10, 320, 640, 3, False, False),
    (5, 10, 10, 640, 640, 3, True, True),
    (4, 10, 10, 320, 640, 1, False, False),
    (7, 5, 5, 640, 640, 3, False, True),
    (1, 5, 7, 32, 96, 3, False, False),            # non-square, Cout not a multiple of 64/160 (N tail), tiny M
    (1, 6, 6, 32, 32, 3, True, False),
    (3, 9, 4, 64, 64, 1, True, True),
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "B%d_%dx%d_%d-%d_k%d_p%d_r%d" % tuple(int(v) for v in c))
def test_conv_fwd_epilogue(case, dtype):
    B, H, W, Cin, Cout, k, pool, use_res = case
    lib = _lib.load()
    rs = np.random.RandomState(hash(case) % (2 ** 31))
    x = rs.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rs.standard_normal((Cout, Cin, k, k)) * (1.4 / np.sqrt(Cin * k * k))).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, Cout).astype(np.float32) * rs.choice([-1, 1], Cout).astype(np.float32)
    shift = (rs.standard_normal(Cout) * 0.3).astype(np.float32)
    res = rs.standard_normal((B, Cout, H, W)).astype(np.float32) if use_res else None
    if dtype == "bf16":            # the oracle sees the same rounded operands; accumulation stays wide on both sides
        x, w = _round_bf16(x), _round_bf16(w)
        res = _round_bf16(res) if use_res else None
    want = rr.conv_nhwc(rr._nhwc(x).astype(np.float64), w.astype(np.float64)) * scale + shift
    if use_res:
        want = want + rr._nhwc(res)
    want = rr.maxpool_nhwc(rr.leaky_relu(want), 2 if pool else 1)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.full((B * Ho * Wo * Cout,), float("nan"), dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, device=_dev())
    flags = _lib.CONV_LRELU | (_lib.CONV_POOL2 if pool else 0)
    rd = _nhwc_dev(res, dtype) if use_res else None
    xd, wd, scd, shd = _nhwc_dev(x, dtype), _pack_w(w, dtype), _t(scale), _t(shift)   # keep the device buffers alive
    _lib.check(lib.subreg_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), _lib.ptr(scd), _lib.ptr(shd), _lib.ptr(rd), None,
                                   None, None, 0, B, H, W, Cin, Cout, k, flags, _lib.dtype_code(dtype), _lib.stream_ptr()),
               "conv_fwd")
    got = _nchw_host(y, B, Cout, Ho, Wo, dtype)
    a, r = _tol(dtype, np.abs(want).max())
    _cmp("conv", got, rr._nchw(want), a, r)


def _pack_w_first(w_oihw, fold=None):
    """mode-1 layout of subreg_pack_conv_weight: [Cout][32] bf16, k = 3 tap + c (3x3) or k = 12 + c (the 1x1 shortcut)."""
    lib = _lib.load()
    O, I, k, _ = w_oihw.shape
    out = torch.empty(O * 32, dtype=torch.bfloat16, device=_dev())
    fd = _t(fold) if fold is not None else None
    _lib.check(lib.subreg_pack_conv_weight(_lib.ptr(_t(w_oihw)), _lib.ptr(fd), _lib.ptr(out), O, I, k, 1, _lib.BF16, _lib.stream_ptr()))
    return out


@pytest.mark.parametrize("shape", [(3, 84, 84), (70, 84, 84), (2, 32, 32), (5, 21, 10), (1, 9, 77), (2, 13, 200), (1, 1, 8)])
def test_conv_first_direct_from_the_fp32_image(shape):
    """conv1 of layer1.0 (models/resnet_language.py:249-251: conv3x3(3 -> 64) + eval BN + LeakyReLU) straight from the NCHW fp32
    image, no im2col buffer (csrc/conv_first.hip), against the oracle on the same bf16-rounded operands: whole images rows per
    tile with a short last tile (13 rows), one-row images, W below and above one 128-pixel wave stride, grid-stride tiles (70)."""
    B, H, W = shape
    lib = _lib.load()
    rs = np.random.RandomState(B * 1000 + H + W)
    x = rs.standard_normal((B, 3, H, W)).astype(np.float32)
    w = (rs.standard_normal((64, 3, 3, 3)) * (1.4 / np.sqrt(27))).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, 64).astype(np.float32) * rs.choice([-1, 1], 64).astype(np.float32)
    shift = (rs.standard_normal(64) * 0.3).astype(np.float32)
    wd = _pack_w_first(w, scale)
    wq = _round_bf16(w * scale[:, None, None, None])                    # what the kernel multiplies with
    want = rr.leaky_relu(rr.conv_nhwc(rr._nhwc(_round_bf16(x)).astype(np.float64), wq.astype(np.float64)) + shift)
    y = torch.full((B * H * W * 64,), float("nan"), dtype=torch.bfloat16, device=_dev())
    xd, shd = _t(x), _t(shift)
    _lib.check(lib.subreg_conv_first_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), _lib.ptr(shd), B, H, W, 64, _lib.CONV_LRELU,
                                         _lib.BF16, _lib.stream_ptr()), "conv_first_fwd")
    got = _nchw_host(y, B, 64, H, W, "bf16")
    a, r = _tol("bf16", np.abs(want).max())
    _cmp("conv_first", got, rr._nchw(want), a, r)
    # without the activation flag
    _lib.check(lib.subreg_conv_first_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), _lib.ptr(shd), B, H, W, 64, 0, _lib.BF16,
                                         _lib.stream_ptr()), "conv_first_fwd")
    want0 = rr.conv_nhwc(rr._nhwc(_round_bf16(x)).astype(np.float64), wq.astype(np.float64)) + shift
    _cmp("conv_first (no act)", _nchw_host(y, B, 64, H, W, "bf16"), rr._nchw(want0), a, r)
    # the entry refuses what it does not implement instead of computing something else
    assert lib.subreg_conv_first_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), _lib.ptr(shd), B, H, W, 64, 0, _lib.F32, _lib.stream_ptr()) != 0
    assert lib.subreg_conv_first_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), _lib.ptr(shd), B, H, W, 160, 0, _lib.BF16, _lib.stream_ptr()) != 0


@pytest.mark.parametrize("shape", [(2, 84, 84), (37, 84, 84), (3, 85, 84), (2, 20, 94)])
def test_layer1_conv3_with_the_shortcut_fed_from_the_image(shape):
    """conv3 of layer1.0 with its 1x1 shortcut conv (models/resnet_language.py:146-147,254-256,286-290) reading the fp32 image
    directly (conv64_resident.hip IMG kernels) against the oracle, and against the im2col route of subreg_conv_fwd it replaces:
    odd sizes (floor pooling drops the last row / column), several tiles per workgroup (37 images)."""
    B, H, W = shape
    lib = _lib.load()
    assert lib.subreg_layer1_direct_supported(B, H, W, _lib.BF16) == 1 and lib.subreg_layer1_direct_supported(B, H, W, _lib.F32) == 0
    assert lib.subreg_layer1_direct_supported(B, 32, 32, _lib.BF16) == 0          # 32x32 goldens keep the general kernels
    assert lib.subreg_layer1_direct_supported(B, 64, 70, _lib.BF16) == 0          # no 48..64-window tiling of 35-window rows
    rs = np.random.RandomState(B + H + W)
    img = rs.standard_normal((B, 3, H, W)).astype(np.float32)
    x = _round_bf16(rs.standard_normal((B, 64, H, W)).astype(np.float32))
    w = _round_bf16((rs.standard_normal((64, 64, 3, 3)) * (1.4 / np.sqrt(576))).astype(np.float32))
    w2 = (rs.standard_normal((64, 3, 1, 1)) * 0.6).astype(np.float32)
    shift = (rs.standard_normal(64) * 0.3).astype(np.float32)
    want = rr.conv_nhwc(rr._nhwc(x).astype(np.float64), w.astype(np.float64)) + \
        rr.conv_nhwc(rr._nhwc(_round_bf16(img)).astype(np.float64), _round_bf16(w2).astype(np.float64)) + shift
    want = rr.maxpool_nhwc(rr.leaky_relu(want), 2)
    Ho, Wo = H // 2, W // 2
    xd, wd, w2d, shd, imgd = _nhwc_dev(x, "bf16"), _pack_w(w, "bf16"), _pack_w_first(w2), _t(shift), _t(img)
    y = torch.full((B * Ho * Wo * 64,), float("nan"), dtype=torch.bfloat16, device=_dev())
    flags = _lib.CONV_LRELU | _lib.CONV_POOL2
    _lib.check(lib.subreg_conv_fwd_image_shortcut(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), _lib.ptr(shd), _lib.ptr(imgd), _lib.ptr(w2d),
                                                  B, H, W, 64, 64, flags, _lib.BF16, _lib.stream_ptr()), "conv_fwd_image_shortcut")
    got = _nchw_host(y, B, 64, Ho, Wo, "bf16")
    a, r = _tol("bf16", np.abs(want).max())
    _cmp("conv3 + image shortcut", got, rr._nchw(want), a, r)
    # the route it replaces: im2col rows + the K = 32 shortcut GEMM
    col = torch.empty(B * H * W * 32, dtype=torch.bfloat16, device=_dev())
    _lib.check(lib.subreg_pack_input(_lib.ptr(imgd), _lib.ptr(col), B, H, W, _lib.BF16, _lib.stream_ptr()))
    y2 = torch.full_like(y, float("nan"))
    _lib.check(lib.subreg_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y2), None, _lib.ptr(shd), None, None, _lib.ptr(col), _lib.ptr(w2d),
                                   32, B, H, W, 64, 64, 3, flags, _lib.BF16, _lib.stream_ptr()), "conv_fwd")
    torch.cuda.synchronize()
    d = (y.float() - y2.float()).abs().max().item()
    assert d <= 2.0 ** -7 * float(np.abs(want).max()), d            # same products, another accumulation order: <= 1 bf16 ulp


@pytest.mark.parametrize("kernel", ["8wave", "wide"])
@pytest.mark.parametrize("shape", [(2, 84, 84), (41, 84, 84), (3, 83, 84), (2, 10, 80), (1, 85, 84), (563, 84, 84)])
def test_layer1_conv1_conv2_fused_from_the_image(shape, kernel):
    """conv1 + BN + LeakyReLU + conv2 + BN + LeakyReLU of layer1.0 (models/resnet_language.py:249-253) in ONE launch, the
    64-channel intermediate kept in LDS (conv64_resident.hip::conv64_fused_first_kernel, and its one-wave-per-SIMD form
    conv64_wide_fused_kernel behind SUBREG_CONV_KERNEL_WIDE), against the oracle (which rounds the intermediate to bf16 where the
    kernels do) and against the two-launch route: last tiles of 1 and 2 rows (83 / 85 rows), several tiles per workgroup (41
    images: every workgroup crosses image boundaries, i.e. both the rolling and the from-scratch conv1 of the wide kernel), a map
    narrower than 84."""
    B, H, W = shape
    lib = _lib.load()
    kflag = _lib.CONV_KERNEL_WIDE if kernel == "wide" else _lib.CONV_KERNEL_GENERAL
    rs = np.random.RandomState(B * 7 + H + W)
    x = rs.standard_normal((B, 3, H, W)).astype(np.float32)
    w1 = (rs.standard_normal((64, 3, 3, 3)) * (1.4 / np.sqrt(27))).astype(np.float32)
    w2 = _round_bf16((rs.standard_normal((64, 64, 3, 3)) * (1.4 / np.sqrt(576))).astype(np.float32))
    sc1 = rs.uniform(0.5, 1.5, 64).astype(np.float32)
    sh1, sh2 = (rs.standard_normal(64) * 0.3).astype(np.float32), (rs.standard_normal(64) * 0.3).astype(np.float32)
    xd, w1d, w2d, sh1d, sh2d = _t(x), _pack_w_first(w1, sc1), _pack_w(w2, "bf16"), _t(sh1), _t(sh2)
    y = torch.full((B * H * W * 64,), float("nan"), dtype=torch.bfloat16, device=_dev())
    _lib.check(lib.subreg_conv12_first_fused(_lib.ptr(xd), _lib.ptr(w1d), _lib.ptr(sh1d), _lib.ptr(w2d), _lib.ptr(sh2d), _lib.ptr(y), B, H, W,
                                             _lib.CONV_LRELU | kflag, _lib.BF16, _lib.stream_ptr()), "conv12_first_fused")
    wmax = 4.0
    if B <= 64:                    # (the 563-image case - every workgroup walks ~15 tiles, as in the benchmark - is checked against
        w1q = _round_bf16(w1 * sc1[:, None, None, None])         # the two-launch route only: the NumPy oracle would need minutes)
        mid = rr.leaky_relu(rr.conv_nhwc(rr._nhwc(_round_bf16(x)).astype(np.float64), w1q.astype(np.float64)) + sh1)
        mid = _round_bf16(mid.astype(np.float32)).astype(np.float64)               # the intermediate is bf16 in LDS
        want = rr.leaky_relu(rr.conv_nhwc(mid, w2.astype(np.float64)) + sh2)
        got = _nchw_host(y, B, 64, H, W, "bf16")
        a, r = _tol("bf16", np.abs(want).max())
        _cmp("conv1+conv2 fused", got, rr._nchw(want), a, r)
        wmax = float(np.abs(want).max())
    # two launches: conv1 from the image, then the 64 -> 64 kernel
    y1 = torch.empty(B * H * W * 64, dtype=torch.bfloat16, device=_dev())
    y2 = torch.full_like(y, float("nan"))
    _lib.check(lib.subreg_conv_first_fwd(_lib.ptr(xd), _lib.ptr(w1d), _lib.ptr(y1), _lib.ptr(sh1d), B, H, W, 64, _lib.CONV_LRELU, _lib.BF16,
                                         _lib.stream_ptr()))
    _lib.check(lib.subreg_conv_fwd(_lib.ptr(y1), _lib.ptr(w2d), _lib.ptr(y2), None, _lib.ptr(sh2d), None, None, None, None, 0, B, H, W, 64, 64, 3,
                                   _lib.CONV_LRELU, _lib.BF16, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y.float()).all())
    d = (y.float() - y2.float()).abs().max().item()
    assert d <= 2.0 ** -7 * max(wmax, float(y2.float().abs().max())), d            # same products, another accumulation order: <= 1 bf16 ulp
    assert lib.subreg_conv12_first_fused(_lib.ptr(xd), _lib.ptr(w1d), _lib.ptr(sh1d), _lib.ptr(w2d), _lib.ptr(sh2d), _lib.ptr(y), B, 32, 32,
                                         _lib.CONV_LRELU, _lib.BF16, _lib.stream_ptr()) != 0     # 32x32: refused, not mis-computed


FUSED_CASES = [
    # B, H, W, Cin, Cout, Cin2 (0 = identity shortcut), pool      conv3 of every block type with its fused shortcut GEMM
    (2, 84, 84, 64, 64, 32, True),       # layer1.0: shortcut = 1x1 conv over the K=32 im2col rows
    (2, 42, 42, 160, 160, 64, True),     # layer2.0
    (3, 21, 21, 320, 320, 160, True),    # layer3.0 (floor pooling)
    (5, 10, 10, 320, 320, 0, False),     # layer3.1: identity shortcut as a GEMM with I
    (5, 10, 10, 640, 640, 320, True),    # layer4.0
    (7, 5, 5, 640, 640, 0, False),       # layer4.1
    (2, 7, 9, 64, 96, 32, False),        # ragged N tail + ragged M
    (6, 10, 10, 640, 320, 640, False),   # the fused input-gradient call of layer4.0 (dX(conv1) + dX(shortcut)), 64-row tiles
    (6, 21, 21, 320, 160, 320, False),   # ... of layer3.0
    (6, 42, 42, 160, 64, 160, False),    # ... of layer2.0 (Cout = 64 tile shape)
]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", FUSED_CASES, ids=lambda c: "B%d_%dx%d_%d-%d_sc%d_p%d" % tuple(int(v) for v in c))
def test_conv_folded_scale_and_fused_shortcut(case, dtype, kernel):
    """Eval-mode conv3 as the backbone runs it: BN scale folded into the packed weights, the shortcut branch
    accumulated as a second GEMM (x2 * w2^T), one shift, LeakyReLU, optional 2x2 max-pool."""
    B, H, W, Cin, Cout, Cin2, pool = case
    lib = _lib.load()
    rs = np.random.RandomState(hash(case) % (2 ** 31))
    x = rs.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rs.standard_normal((Cout, Cin, 3, 3)) * (1.4 / np.sqrt(Cin * 9))).astype(np.float32)
    sc3 = rs.uniform(0.5, 1.5, Cout).astype(np.float32) * rs.choice([-1, 1], Cout).astype(np.float32)
    shift = (rs.standard_normal(Cout) * 0.3).astype(np.float32)
    ident = Cin2 == 0
    c2 = Cout if ident else Cin2
    x2 = rs.standard_normal((B, c2, H, W)).astype(np.float32)
    w2 = np.eye(Cout, dtype=np.float32)[:, :, None, None] if ident else \
        (rs.standard_normal((Cout, c2, 1, 1)) / np.sqrt(c2)).astype(np.float32)
    sc2 = np.ones(Cout, np.float32) if ident else rs.uniform(0.5, 1.5, Cout).astype(np.float32)
    wf, w2f = w * sc3[:, None, None, None], w2 * sc2[:, None, None, None]
    if dtype == "bf16":            # compare against the operands the kernel really sees (fold, then round)
        x, x2, wf, w2f = _round_bf16(x), _round_bf16(x2), _round_bf16(wf), _round_bf16(w2f)
    want = rr.conv_nhwc(rr._nhwc(x).astype(np.float64), wf.astype(np.float64)) + \
        rr.conv_nhwc(rr._nhwc(x2).astype(np.float64), w2f.astype(np.float64)) + shift
    want = rr.maxpool_nhwc(rr.leaky_relu(want), 2 if pool else 1)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.full((B * Ho * Wo * Cout,), float("nan"), dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, device=_dev())
    xd, x2d, shd = _nhwc_dev(x, dtype), _nhwc_dev(x2, dtype), _t(shift)
    wd = _pack_w(w, dtype, fold=sc3)
    if ident:
        w2d = torch.empty(Cout * Cout, dtype=y.dtype, device=_dev())
        _lib.check(lib.subreg_pack_identity(_lib.ptr(w2d), Cout, _lib.dtype_code(dtype), _lib.stream_ptr()))
    else:
        w2d = _pack_w(w2, dtype, fold=sc2)
    flags = _lib.CONV_LRELU | (_lib.CONV_POOL2 if pool else 0) | _kernel_flag(kernel, dtype, Cout)
    _check_conv(lib.subreg_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), None, _lib.ptr(shd), None, None, _lib.ptr(x2d),
                                    _lib.ptr(w2d), c2, B, H, W, Cin, Cout, 3, flags, _lib.dtype_code(dtype), _lib.stream_ptr()),
                kernel, "conv_fwd(fused)")
    got = _nchw_host(y, B, Cout, Ho, Wo, dtype)
    a, r = _tol(dtype, np.abs(want).max())
    _cmp("fused conv3", got, rr._nchw(want), a, r)


# ---------------------------------------------------------------- layer-1 kernel (conv64_resident.hip) off its home shape
# subreg_conv_fwd sends Cin = Cout = 64, 3x3, folded-scale, bf16 calls with 64 <= W <= 95 to the persistent kernel; the
# backbone only ever calls it at 84x84.  Other widths / heights exercise what 84 does not: tiles of 2 or 3 image rows with a
# ragged last tile (H % R != 0), DMA pieces without pixels (W < 80), the last pad column shared with the next block
# (W = 95), odd H / W under the pooling floor, a single image, the un-activated epilogue.
R64_CASES = [
    # B, H, W, pool, shortcut (Cin2 = 32), act
    (3, 66, 80, False, False, True),
    (2, 85, 94, False, False, True),       # R = 2 rows per tile, ragged last tile
    (1, 64, 95, False, False, False),      # widest admitted; no activation
    (5, 7, 65, False, False, True),        # 3 rows per tile, DMA pieces without pixels, tiny image (W = 64 would need 4 rows: general kernel)
    (2, 85, 94, True, True, True),         # pooled: odd height (floor), windows per tile searched by the host
    (3, 66, 70, True, False, True),
    (1, 65, 95, True, True, True),
    (4, 84, 84, True, False, False),
]


@pytest.mark.parametrize("case", R64_CASES, ids=lambda c: "B%d_%dx%d_p%d_sc%d_a%d" % tuple(int(v) for v in c))
def test_conv64_resident_other_shapes(case):
    B, H, W, pool, sc, act = case
    lib = _lib.load()
    rs = np.random.RandomState(hash(case) % (2 ** 31))
    x = _round_bf16(rs.standard_normal((B, 64, H, W)).astype(np.float32))
    wf = _round_bf16((rs.standard_normal((64, 64, 3, 3)) * (1.4 / 24)).astype(np.float32))
    shift = (rs.standard_normal(64) * 0.3).astype(np.float32)
    want = rr.conv_nhwc(rr._nhwc(x).astype(np.float64), wf.astype(np.float64)) + shift
    x2d = w2d = None
    if sc:
        x2 = _round_bf16(rs.standard_normal((B, 32, H, W)).astype(np.float32))
        w2 = _round_bf16((rs.standard_normal((64, 32, 1, 1)) / np.sqrt(32)).astype(np.float32))
        want = want + rr.conv_nhwc(rr._nhwc(x2).astype(np.float64), w2.astype(np.float64))
        x2d, w2d = _nhwc_dev(x2, "bf16"), _pack_w(w2, "bf16")
    if act:
        want = rr.leaky_relu(want)
    want = rr.maxpool_nhwc(want, 2 if pool else 1)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    y = torch.full((B * Ho * Wo * 64,), float("nan"), dtype=torch.bfloat16, device=_dev())
    xd, wd, shd = _nhwc_dev(x, "bf16"), _pack_w(wf, "bf16"), _t(shift)
    flags = (_lib.CONV_LRELU if act else 0) | (_lib.CONV_POOL2 if pool else 0)
    _lib.check(lib.subreg_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), None, _lib.ptr(shd), None, None, _lib.ptr(x2d),
                                   _lib.ptr(w2d), 32 if sc else 0, B, H, W, 64, 64, 3, flags, _lib.BF16, _lib.stream_ptr()), "conv_fwd")
    got = _nchw_host(y, B, 64, Ho, Wo, "bf16")
    a, r = _tol("bf16", np.abs(want).max())
    _cmp("conv64", got, rr._nchw(want), a, r)


# ---------------------------------------------------------------- production tile shapes (bench-scale batches)
# Every tiling subreg_conv_fwd selects at the batch sizes the benchmark / the pretraining step run (conv_fwd.hip:
# 256-row / 128-row one-tap / 128-row three-tap two-wave tiles for Cout % 160 == 0, the three Cout = 64 tilings) needs
# grids far beyond what a full-tensor NumPy oracle finishes in seconds, so the oracle is evaluated on a SAMPLE of output
# positions: all image corners / borders of the first and last image, rows around the 128/256-row tile boundaries, and
# 384 random positions.
BIG_CONV_CASES = [
    # B, H, W, Cin, Cout, k, pool, Cin2 (fused shortcut GEMM: -1 none, 0 identity)     -> tiling reached (bf16)
    (64, 84, 84, 64, 64, 3, False, -1),       # Cout=64, 256 rows, 3 taps / step
    (64, 84, 84, 64, 64, 3, True, 32),        # Cout=64 pooled, 1 tap / step + K=32 shortcut (layer1.0 conv3)
    (64, 84, 84, 32, 64, 1, False, -1),       # K=32 first conv, 128-row streaming tiles
    (64, 42, 42, 160, 160, 3, False, -1),     # 441 tiles of 256 rows (>= 384): 256x160
    (64, 42, 42, 160, 160, 3, True, 64),      # ... pooled + shortcut (layer2.0 conv3)
    (40, 42, 42, 64, 160, 3, False, -1),      # 276 tiles of 256 rows (< 384): 128-row / 1 tap / 3 per CU at 42x42 (AROWS 224)
    (160, 21, 21, 320, 320, 3, False, -1),    # 552 x 256-row tiles = 2 rounds -> 1104 x 128-row tiles (1 tap, 3 per CU)
    (160, 21, 21, 320, 320, 3, True, 160),    # layer3.0 conv3 at that batch (floor pooling 21 -> 10)
    (300, 10, 10, 320, 640, 3, False, -1),    # 472 x 256-row tiles: one round of 256 rows
    (700, 5, 5, 640, 640, 3, False, 0),       # 548 x 128-row tiles, identity shortcut (layer4.1 conv3)
    (120, 10, 10, 320, 320, 3, False, 0),     # 94 m-tiles x 2: <= 256 workgroups -> 3 taps / step, two waves per tile
]


def _conv_at(x_nhwc, w_oihw, b, h, wq):
    """Oracle conv (cross-correlation, zero padding; resnet_language.py:402-405) at the sampled pixels only -> [S, O] f64."""
    _B, H, W, _C = x_nhwc.shape
    O, _, k, _ = w_oihw.shape
    r = k // 2
    out = np.zeros((len(b), O))
    for dy in range(k):
        for dx in range(k):
            hh, ww = h + dy - r, wq + dx - r
            ok = (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)
            a = x_nhwc[b, np.clip(hh, 0, H - 1), np.clip(ww, 0, W - 1), :].astype(np.float64) * ok[:, None]
            out += a @ w_oihw[:, :, dy, dx].T.astype(np.float64)
    return out


def _sample_positions(rs, B, Ho, Wo, n_random=384):
    pts = set()
    for b in (0, B - 1):
        for h in (0, 1, Ho // 2, Ho - 2, Ho - 1):
            for w in (0, 1, Wo // 2, Wo - 2, Wo - 1):
                pts.add((b, max(h, 0), max(w, 0)))
    n = B * Ho * Wo
    for t in (128, 256, 384, 512, 768, 1152, n - 384, n - 256, n - 128):      # output rows next to tile boundaries (LINEAR order)
        for d in (-1, 0, 1):
            m = min(max(t + d, 0), n - 1)
            pts.add((m // (Ho * Wo), (m // Wo) % Ho, m % Wo))
    for m in rs.randint(0, n, n_random):
        pts.add((int(m) // (Ho * Wo), (int(m) // Wo) % Ho, int(m) % Wo))
    p = np.array(sorted(pts), dtype=np.int64)
    return p[:, 0], p[:, 1], p[:, 2]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("case", BIG_CONV_CASES, ids=lambda c: "B%d_%dx%d_%d-%d_k%d_p%d_sc%d" % tuple(int(v) for v in c))
def test_conv_fwd_production_tiles(case, dtype, kernel):
    B, H, W, Cin, Cout, k, pool, Cin2 = case
    lib = _lib.load()
    rs = np.random.RandomState(hash(case) % (2 ** 31))
    td = torch.bfloat16 if dtype == "bf16" else torch.float32

    def rnd(a):
        return _round_bf16(a) if dtype == "bf16" else a
    x = rnd(rs.standard_normal((B, H, W, Cin)).astype(np.float32))                 # NHWC, as the kernel reads it
    w = (rs.standard_normal((Cout, Cin, k, k)) * (1.4 / np.sqrt(Cin * k * k))).astype(np.float32)
    sc3 = rs.uniform(0.5, 1.5, Cout).astype(np.float32) * rs.choice([-1, 1], Cout).astype(np.float32)
    shift = (rs.standard_normal(Cout) * 0.3).astype(np.float32)
    wf = rnd(w * sc3[:, None, None, None])                                         # BN scale folded, then rounded
    x2 = w2f = None
    if Cin2 >= 0:
        c2 = Cout if Cin2 == 0 else Cin2
        x2 = rnd(rs.standard_normal((B, H, W, c2)).astype(np.float32))
        w2 = np.eye(Cout, dtype=np.float32)[:, :, None, None] if Cin2 == 0 else \
            (rs.standard_normal((Cout, c2, 1, 1)) / np.sqrt(c2)).astype(np.float32)
        sc2 = np.ones(Cout, np.float32) if Cin2 == 0 else rs.uniform(0.5, 1.5, Cout).astype(np.float32)
        w2f = rnd(w2 * sc2[:, None, None, None])
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    pb, ph, pw = _sample_positions(rs, B, Ho, Wo)

    def pre_act(h, wq):
        v = _conv_at(x, wf, pb, h, wq) + shift
        if x2 is not None:
            v = v + _conv_at(x2, w2f, pb, h, wq)
        return v
    if pool:
        want = np.maximum.reduce([pre_act(2 * ph + dy, 2 * pw + dx) for dy in (0, 1) for dx in (0, 1)])
    else:
        want = pre_act(ph, pw)
    want = rr.leaky_relu(want)
    xd = torch.from_numpy(x).to(_dev(), td)
    x2d = torch.from_numpy(x2).to(_dev(), td) if x2 is not None else None
    wd = _pack_w(w, dtype, fold=sc3)
    w2d = None
    if Cin2 == 0:
        w2d = torch.empty(Cout * Cout, dtype=td, device=_dev())
        _lib.check(lib.subreg_pack_identity(_lib.ptr(w2d), Cout, _lib.dtype_code(dtype), _lib.stream_ptr()))
    elif Cin2 > 0:
        w2d = _pack_w(w2, dtype, fold=sc2)
    y = torch.full((B * Ho * Wo, Cout), float("nan"), dtype=td, device=_dev())
    shd = _t(shift)
    flags = _lib.CONV_LRELU | (_lib.CONV_POOL2 if pool else 0) | _kernel_flag(kernel, dtype, Cout, k)
    _check_conv(lib.subreg_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), None, _lib.ptr(shd), None, None, _lib.ptr(x2d),
                                    _lib.ptr(w2d), (Cout if Cin2 == 0 else max(Cin2, 0)), B, H, W, Cin, Cout, k, flags,
                                    _lib.dtype_code(dtype), _lib.stream_ptr()), kernel, "conv_fwd(big)")
    torch.cuda.synchronize()
    assert not torch.isnan(y.float()).any().item(), "an output row was never written"
    rows = torch.from_numpy((pb * Ho + ph) * Wo + pw).to(_dev())
    got = y[rows].float().cpu().numpy()
    a, r = _tol(dtype, np.abs(want).max())
    _cmp("conv (sampled)", got, want, a, r)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 21, 21, 160, 320, 3), (2, 42, 42, 64, 160, 1), (5, 5, 5, 640, 640, 3)])
def test_conv_raw_stats_and_bn_train(shape, dtype):
    """Train-mode path: raw conv + partial sums -> finalize -> batch scale/shift and running-stat update."""
    B, H, W, Cin, Cout, k = shape
    lib = _lib.load()
    rs = np.random.RandomState(7)
    x = rs.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rs.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    if dtype == "bf16":
        x, w = _round_bf16(x), _round_bf16(w)
    gw, gb = rs.uniform(0.5, 1.5, Cout).astype(np.float32), (rs.standard_normal(Cout) * 0.1).astype(np.float32)
    rm, rv = (rs.standard_normal(Cout) * 0.1).astype(np.float32), rs.uniform(0.5, 1.5, Cout).astype(np.float32)
    raw = rr.conv_nhwc(rr._nhwc(x).astype(np.float64), w.astype(np.float64))
    want, wrm, wrv = rr.bn_train_nhwc(raw, gw, gb, rm, rv)
    want = rr.leaky_relu(want)
    dt = _lib.dtype_code(dtype)
    rows = lib.subreg_conv_stats_rows(dt, B, H, W, Cout)
    stats = torch.zeros(rows * Cout * 2, dtype=torch.float32, device=_dev())
    y = torch.empty(B * H * W * Cout, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, device=_dev())
    xd, wd = _nhwc_dev(x, dtype), _pack_w(w, dtype)                                   # keep the device buffers alive
    _lib.check(lib.subreg_conv_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), None, None, None, _lib.ptr(stats), None, None, 0,
                                   B, H, W, Cin, Cout, k, _lib.CONV_RAW_STATS, dt, _lib.stream_ptr()))
    drm, drv, gwd, gbd = _t(rm), _t(rv), _t(gw), _t(gb)
    sc, sh = torch.empty(Cout, device=_dev()), torch.empty(Cout, device=_dev())
    _lib.check(lib.subreg_bn_train_finalize(_lib.ptr(stats), rows, Cout, B * H * W, _lib.ptr(gwd), _lib.ptr(gbd),
                                            _lib.ptr(drm), _lib.ptr(drv), 0.1, 1e-5, _lib.ptr(sc), _lib.ptr(sh), None, None,
                                            _lib.stream_ptr()))
    _lib.check(lib.subreg_bn_apply(_lib.ptr(y), _lib.ptr(sc), _lib.ptr(sh), None, None, None, None, 1.0, None, _lib.ptr(y), B, H, W,
                                   Cout, _lib.CONV_LRELU, dt, _lib.stream_ptr()))
    got = _nchw_host(y, B, Cout, H, W, dtype)
    a, r = _tol(dtype, np.abs(want).max())
    _cmp("bn-train out", got, rr._nchw(want), a, r)
    _cmp("running_mean", drm.cpu().numpy(), wrm, 1e-5 if dtype == "f32" else 2e-3, 1e-4)
    _cmp("running_var", drv.cpu().numpy(), wrv, 1e-5 if dtype == "f32" else 2e-3, 1e-4 if dtype == "f32" else 1e-2)


@pytest.mark.parametrize("shape", [(64, 5, 5, 640, 640), (16, 10, 10, 320, 320), (64, 5, 5, 320, 640), (5, 5, 5, 640, 320), (3, 9, 7, 160, 160)])
def test_conv_split_k_with_workspace(shape):
    """subreg_conv_fwd_ws on the small-M 3x3 layers of the pretraining step (layer 3.1 / 4.x at B = 64: 52-100 tiles): K split
    over several workgroups per tile + reduce pass, in both of its modes - raw conv + batch-statistics partials (train-mode
    forward) and plain scale / shift / LeakyReLU (the dX convolutions) - against the oracle, and against the one-launch path
    (same bf16 results up to the summation order of the fp32 partial sums)."""
    B, H, W, Cin, Cout = shape
    lib = _lib.load()
    dt = _lib.BF16
    need = lib.subreg_conv_splitk_floats(B, H, W, Cin, Cout, 3, dt)
    assert need >= 2 * B * H * W * Cout, "the K split was not planned for this shape"
    assert lib.subreg_conv_splitk_floats(700, 42, 42, 160, 160, 3, dt) == 0          # big layers: never
    assert lib.subreg_conv_splitk_floats(B, H, W, Cin, Cout, 1, dt) == 0 and lib.subreg_conv_splitk_floats(B, H, W, Cin, Cout, 3, _lib.F32) == 0
    rs = np.random.RandomState(17)
    x = _round_bf16(rs.standard_normal((B, Cin, H, W)).astype(np.float32))
    w = _round_bf16((rs.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32))
    raw = rr.conv_nhwc(rr._nhwc(x).astype(np.float64), w.astype(np.float64))
    xd, wd = _nhwc_dev(x, "bf16"), _pack_w(w, "bf16")
    ws = torch.full((need,), float("nan"), dtype=torch.float32, device=_dev())
    M = B * H * W
    rows = lib.subreg_conv_stats_rows(dt, B, H, W, Cout)
    outs = []
    for use_ws in (True, False):
        stats = torch.full((rows * Cout * 2,), float("nan"), dtype=torch.float32, device=_dev())
        y = torch.empty(M * Cout, dtype=torch.bfloat16, device=_dev())
        _lib.check(lib.subreg_conv_fwd_ws(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), None, None, None, _lib.ptr(stats), None, None, 0,
                                          B, H, W, Cin, Cout, 3, _lib.CONV_RAW_STATS, dt, _lib.ptr(ws) if use_ws else None,
                                          need if use_ws else 0, _lib.stream_ptr()))
        st = stats.cpu().numpy().reshape(rows, Cout, 2).astype(np.float64)
        assert np.isfinite(st).all()
        outs.append((y.float().cpu().numpy().reshape(M, Cout), st.sum(axis=0)))
    a, r = _tol("bf16", np.abs(raw).max())
    flat = raw.reshape(M, Cout)
    _cmp("raw (split)", outs[0][0], flat, a, r)
    _cmp("sum", outs[0][1][:, 0], flat.sum(0), 1e-3 * np.abs(flat).sum(0).max(), 1e-4)
    _cmp("sumsq", outs[0][1][:, 1], (flat * flat).sum(0), 1e-3 * (flat * flat).sum(0).max(), 1e-4)
    _cmp("raw: split vs one launch", outs[0][0], outs[1][0], 2.0 ** -7 * np.abs(flat).max(), 0)   # one bf16 ulp at the largest value
    _cmp("stats: split vs one launch", outs[0][1], outs[1][1], 1e-4 * np.abs(outs[1][1]).max(), 1e-5)
    # plain mode with scale, shift and LeakyReLU (the dX convolutions pass shift = 0, no activation)
    sc, sh = rs.uniform(0.5, 1.5, Cout).astype(np.float32), (rs.standard_normal(Cout) * 0.2).astype(np.float32)
    want = rr.leaky_relu(flat * sc + sh)
    scd, shd = _t(sc), _t(sh)
    y = torch.empty(M * Cout, dtype=torch.bfloat16, device=_dev())
    _lib.check(lib.subreg_conv_fwd_ws(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), _lib.ptr(scd), _lib.ptr(shd), None, None, None, None, 0,
                                      B, H, W, Cin, Cout, 3, _lib.CONV_LRELU, dt, _lib.ptr(ws), need, _lib.stream_ptr()))
    a, r = _tol("bf16", np.abs(want).max())
    _cmp("scale/shift/act (split)", y.float().cpu().numpy().reshape(M, Cout), want, a, r)
    # ... and with a residual added before the activation (eval mode, conv3 of a block with an identity shortcut, SUBREG_EVAL_SPLITK=1)
    if Cin == Cout:
        want = rr.leaky_relu(flat + sh + rr._nhwc(x).astype(np.float64).reshape(M, Cout))
        y = torch.full((M * Cout,), float("nan"), dtype=torch.bfloat16, device=_dev())
        _lib.check(lib.subreg_conv_fwd_ws(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), None, _lib.ptr(shd), _lib.ptr(xd), None, None, None, 0,
                                          B, H, W, Cin, Cout, 3, _lib.CONV_LRELU, dt, _lib.ptr(ws), need, _lib.stream_ptr()))
        a, r = _tol("bf16", np.abs(want).max())
        _cmp("shift/residual/act (split)", y.float().cpu().numpy().reshape(M, Cout), want, a, r)


def test_eval_forward_with_the_k_split_switched_on():
    """SUBREG_EVAL_SPLITK=1 (off by default: profiles/r05_eval_splitk.txt): the eval forward of the small maps through
    subreg_conv_fwd_ws + reduce pass, the identity shortcut added there - the backbone goldens of the reference in a fresh process
    (the switch is read once)."""
    import subprocess
    import sys
    env = dict(os.environ, SUBREG_EVAL_SPLITK="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "test_backbone_golden", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and "4 passed" in p.stdout, p.stdout[-1500:] + p.stderr[-500:]


# ---------------------------------------------------------------- backbone against the reference's golden vectors
def _params_from_sd(sd):
    return {k: _t(v) for k, v in sd.items() if v.dtype != np.int64}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("hw", [32, 84])
def test_backbone_golden(hw, dtype):
    from subreg_hip.backbone import HipBackbone
    g = np.load(os.path.join(GOLDEN, "backbone.npz"))
    sd = syn.make_state_dict(21)
    params = _params_from_sd(sd)
    hb = HipBackbone(params, (1, 1, 2, 2), dtype, block_size=1)
    x = _t(syn.make_images(31, 4, hw))
    feat, stages = hb.forward(x, return_stages=True)
    torch.cuda.synchronize()
    a, r = _tol(dtype, float(np.abs(g["hw%d.eval_feat" % hw]).max()), bf16=3e-2)
    _cmp("stage1 img0", stages[0][0].cpu().numpy(), g["hw%d.eval_f0_b0" % hw], a, r)
    _cmp("feat", feat.cpu().numpy(), g["hw%d.eval_feat" % hw], a, r)
    logits = feat.cpu().numpy() @ sd["classifier.weight"].T
    _cmp("logits", logits, g["hw%d.eval_logits" % hw], a, r)
    # train-mode forward with the same injected masks as the reference run
    feat_t = hb.forward(x, train=True, masks=rr.MaskSource(41))
    torch.cuda.synchronize()
    _cmp("train logits", feat_t.cpu().numpy() @ sd["classifier.weight"].T, g["hw%d.train_logits" % hw],
         5e-4 if dtype == "f32" else a, 5e-4 if dtype == "f32" else r)
    for k in ("layer1.0.bn1", "layer2.0.downsample.1", "layer3.1.bn2", "layer4.1.bn3"):
        _cmp(k + ".running_mean", params[k + ".running_mean"].cpu().numpy(), g["hw%d.%s.running_mean" % (hw, k)],
             1e-5 if dtype == "f32" else 3e-3, 1e-4 if dtype == "f32" else 1e-2)
        _cmp(k + ".running_var", params[k + ".running_var"].cpu().numpy(), g["hw%d.%s.running_var" % (hw, k)],
             1e-5 if dtype == "f32" else 3e-3, 1e-4 if dtype == "f32" else 2e-2)
    feat2 = hb.forward(x)
    torch.cuda.synchronize()
    _cmp("eval after stats moved", feat2.cpu().numpy() @ sd["classifier.weight"].T, g["hw%d.eval2_logits" % hw],
         5e-4 if dtype == "f32" else a, 5e-4 if dtype == "f32" else r)


def test_graphed_eval_forward_equals_eager_and_follows_weight_updates():
    """HipBackbone.forward_graphed (what ResNet.forward runs in eval mode: the unchanged reference loop calls it with the same shapes
    every epoch) replays a cached hipGraph from the third call of a shape on.  It must return what the eager forward returns,
    for whatever input is passed, and re-capture when a weight tensor changes (refresh() sees tensor._version) or a larger
    shape moves the workspaces."""
    from subreg_hip.backbone import HipBackbone
    sd = syn.make_state_dict(3)
    params = {k: _t(v) for k, v in sd.items() if v.dtype != np.int64}
    hb = HipBackbone(params, (1, 1, 2, 2), "bf16")
    xs = [_t(syn.make_images(5 + i, 6, 84)) for i in range(3)]
    want = [hb.forward(x).clone() for x in xs]
    for rep in range(3):                                   # eager, capture + replay, replay
        for x, w in zip(xs, want):
            got = hb.forward_graphed(x)
            assert torch.equal(got, w), (rep, (got - w).abs().max().item())
    assert hb._graphs[(tuple(xs[0].shape), 0)]["graph"] is not None
    # input-sequence prefetch: the same three tensors in the same order - the later rounds are served by forwards started ahead of their
    # call (same results, asserted above); an in-place edit of a predicted tensor (its _version moves) drops the prediction
    if hb.EVAL_PREFETCH > 0:
        assert hb.prefetch_hits >= 2, hb.prefetch_hits
        hb.forward_graphed(xs[0])                          # (starts the forwards of xs[1], xs[2] as they were)
        hits = hb.prefetch_hits
        with torch.no_grad():
            xs[1].mul_(0.5)
        want_half = hb.forward(xs[1]).clone()
        assert not torch.equal(want_half, want[1])
        assert torch.equal(hb.forward_graphed(xs[1]), want_half) and hb.prefetch_hits == hits
        assert torch.equal(hb.forward_graphed(xs[2]), want[2])
        want[1] = want_half
        for x, w in zip(xs, want):                         # and the chain is learnt again
            assert torch.equal(hb.forward_graphed(x), w)
        torch.cuda.synchronize()
    with torch.no_grad():
        params["layer2.0.conv2.weight"].mul_(1.25)         # in-place update: version bump -> re-pack -> cached graph dropped
    want2 = hb.forward(xs[0]).clone()
    assert not torch.equal(want2, want[0])
    for rep in range(3):
        assert torch.equal(hb.forward_graphed(xs[0]), want2), rep
    big = _t(syn.make_images(9, 40, 84))                   # a larger shape re-allocates the workspaces under the cached graph
    want_big = hb.forward(big).clone()
    for rep in range(3):
        assert torch.equal(hb.forward_graphed(big), want_big)
        assert torch.equal(hb.forward_graphed(xs[0]), want2)
    # the cache is bounded (least recently used shape goes) and holds no entry captured under older weights / workspaces
    for b in range(1, hb.GRAPH_CACHE + 4):
        xb = _t(syn.make_images(20 + b, b, 84))
        for rep in range(2):
            hb.forward_graphed(xb)
    assert len(hb._graphs) <= hb.GRAPH_CACHE
    tokens = (hb._fold_token, hb._ws_token)
    assert all(e["graph"] is None or e["tokens"] == tokens for e in hb._graphs.values())
    # a capture that fails leaves the shape on the eager path (no retry every call), with a warning, and the right result
    import warnings
    x7 = _t(syn.make_images(77, hb.GRAPH_CACHE + 9, 84))       # (a batch size the loop above has not captured)
    want7 = hb.forward(x7).clone()
    hb.forward_graphed(x7)
    real_forward, hit = hb.forward, []

    def failing_forward(x, *a, out=None, **k):
        if out is not None:                                # (the call inside the capture)
            hit.append(1)
            raise RuntimeError("injected capture failure")
        return real_forward(x, *a, out=out, **k)
    hb.forward = failing_forward
    try:
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            assert torch.equal(hb.forward_graphed(x7), want7)
            assert torch.equal(hb.forward_graphed(x7), want7)
        assert len(hit) == 1 and any("could not be captured" in str(w.message) for w in wl)
        assert hb._graphs[(tuple(x7.shape), 0)]["eager_only"]
    finally:
        hb.forward = real_forward


def test_prefetched_two_lane_forwards_equal_the_eager_ones():
    """forward_graphed's input-sequence prefetch with forwards that are themselves split over two lanes (>= 128 images): a prefetched
    forward runs on workspace sets of its own (lane_base 2, 3 / 4, 5) while the caller's stream may run another forward of the same
    backbone on sets 0, 1 - results must be those of the eager forward, whichever path served the call."""
    from subreg_hip.backbone import HipBackbone
    sd = syn.make_state_dict(4)
    params = {k: _t(v) for k, v in sd.items() if v.dtype != np.int64}
    hb = HipBackbone(params, (1, 1, 2, 2), "bf16")
    if hb.EVAL_PREFETCH <= 0:
        pytest.skip("prefetch switched off in this environment")
    xs = [_t(syn.make_images(40 + i, 130, 84)) for i in range(4)]
    want = [hb.forward(x).clone() for x in xs]
    x70 = xs[3][:70].contiguous()
    want70 = hb.forward(x70).clone()
    for rep in range(4):
        for i, x in enumerate(xs):
            got = hb.forward_graphed(x)
            assert torch.equal(got, want[i]), (rep, i, (got - want[i]).abs().max().item())
            if rep == 3 and i == 1:                        # an eager forward of another batch on sets 0, 1 while two prefetches are in flight
                assert torch.equal(hb.forward(x70), want70)
    torch.cuda.synchronize()
    assert hb.prefetch_hits >= 6, hb.prefetch_hits


def test_workspace_capacity_is_not_monotone_in_batch():
    """A train-mode forward at B=64 followed by B=55 (84x84, bf16): the smaller batch needs MORE BN-partial floats (layer 2
    switches to 128-row tiles below 56 images).  The workspace must grow, and the result must equal a fresh backbone's."""
    from subreg_hip.backbone import HipBackbone
    sd = syn.make_state_dict(21)
    lib = _lib.load()
    outs = []
    for warm in (True, False):
        hb = HipBackbone(_params_from_sd(sd), (1, 1, 2, 2), "bf16", block_size=1)
        if warm:
            hb.forward(_t(syn.make_images(5, 64, 84)), train=True, masks=rr.OnesMaskSource())
            for k, v in sd.items():                                   # undo the running-stat update of the warm-up pass
                if "running_" in k:
                    hb.params[k].copy_(_t(v))
            hb.nbt = [0] * 6
        feat = hb.forward(_t(syn.make_images(6, 55, 84)), train=True, masks=rr.OnesMaskSource())
        torch.cuda.synchronize()
        need = lib.subreg_backbone_stats_floats(C.byref(hb._desc), 55, 84, 84)
        assert hb._stats.numel() >= need, (hb._stats.numel(), need)
        outs.append(feat.cpu().numpy())
    assert lib.subreg_backbone_stats_floats(C.byref(hb._desc), 55, 84, 84) > lib.subreg_backbone_stats_floats(C.byref(hb._desc), 64, 84, 84)
    assert np.array_equal(outs[0], outs[1])


def test_backbone_matches_oracle_dropblock5():
    """block_size 5 (no --no_dropblock): host-prepared block masks with the reference's pairing quirk."""
    from subreg_hip.backbone import HipBackbone
    sd = syn.make_state_dict(5)
    ref_sd = rr.copy_state_dict(sd)
    hb = HipBackbone(_params_from_sd(sd), (1, 1, 2, 2), "f32", block_size=5)
    onet = rr.ResNetRef(ref_sd, block_size=5)
    hb.nbt = [39999] * 6
    for k in onet.nbt:
        onet.nbt[k] = 39999
    x = syn.make_images(3, 6, 84)
    onet.train()
    want = onet.features(x, rr.MaskSource(9))
    got = hb.forward(_t(x), train=True, masks=rr.MaskSource(9))
    torch.cuda.synchronize()
    _cmp("dropblock5 train feat", got.cpu().numpy(), want, 5e-4, 5e-4)


# ---------------------------------------------------------------- classifier / regularizers against the goldens
@pytest.mark.parametrize("case", ["rand.k5", "rand.k40", "trained.k5", "trained.k40"])
def test_subspace_kernels_golden(case):
    from subreg_hip import functional as HF
    g = np.load(os.path.join(GOLDEN, "reg.npz"))
    wb, w = _t(g[case + ".w_base"]), _t(g[case + ".w"]).requires_grad_(True)
    q, info = HF.subspace_basis(wb)
    assert int(info.item()) == 0
    qq = q.cpu().numpy().astype(np.float64)
    _cmp("orthonormal", qq @ qq.T, np.eye(qq.shape[0]), 1e-6, 0)
    P = HF.SubspaceProjectFn.apply(w, q)
    _cmp("P", P.detach().cpu().numpy(), g[case + ".P"], 1e-6, 1e-5)
    loss = HF.SqDiffFn.apply(P, w, 0.7)
    loss.backward()
    _cmp("loss1", loss.item(), g[case + ".loss1"], 1e-6, 1e-5)
    _cmp("grad", w.grad.cpu().numpy(), g[case + ".grad"], 1e-6, 1e-5)


def test_subspace_kernels_351_base_classes():
    """tieredImageNet-sized base set (351 classes, BASELINE.json configs[4]): basis + projection + loss1 gradient against the
    oracle's QR restatement (the same code the 60-class goldens pin)."""
    from oracle import subspace_ref as sr
    from subreg_hip import functional as HF
    rs = np.random.RandomState(17)
    wb = (rs.standard_normal((351, 640)) * 0.05 + rs.standard_normal((351, 1)) * rs.standard_normal((1, 640)) * 0.05).astype(np.float32)
    w = (rs.standard_normal((5, 640)) * 0.04).astype(np.float32)
    q, info = HF.subspace_basis(_t(wb))
    assert int(info.item()) == 0
    qq = q.cpu().numpy().astype(np.float64)
    _cmp("orthonormal", qq @ qq.T, np.eye(351), 1e-5, 0)
    wt = _t(w).requires_grad_(True)
    P = HF.SubspaceProjectFn.apply(wt, q)
    _cmp("P", P.detach().cpu().numpy(), sr.get_projected_weight(wb, w, np.float64), 1e-5, 1e-4)
    loss = HF.SqDiffFn.apply(P, wt, 0.7)
    loss.backward()
    l_ref, g_ref = sr.loss1_and_grad(0.7, wb, w)
    _cmp("loss1", loss.item(), l_ref, 1e-6, 1e-4)
    _cmp("grad", wt.grad.cpu().numpy(), g_ref, 1e-5, 1e-4)


def test_frob_kernels_golden():
    from subreg_hip import functional as HF
    g = np.load(os.path.join(GOLDEN, "reg.npz"))
    W = _t(g["frob.W"]).requires_grad_(True)
    l0 = HF.FrobFn.apply(W[:60], _t(g["frob.base"]), 0.2)
    l0.backward()
    assert l0.item() == 0.0 and not W.grad.cpu().numpy().any()          # zero sub-gradient at 0
    W2 = _t(g["frob.W2"]).requires_grad_(True)
    l1 = HF.FrobFn.apply(W2[:60], _t(g["frob.base"]), 0.2) + HF.FrobFn.apply(W2[60:70], _t(g["frob.prev"]), 0.1)
    l1.backward()
    _cmp("loss", l1.item(), g["frob.loss"], 1e-6, 1e-5)
    _cmp("grad", W2.grad.cpu().numpy(), g["frob.grad"], 1e-7, 1e-5)


def test_linear_fwd_bwd():
    from subreg_hip import functional as HF
    rs = np.random.RandomState(3)
    f, w, b = rs.standard_normal((37, 640)).astype(np.float32), rs.standard_normal((65, 640)).astype(np.float32) * 0.05, \
        rs.standard_normal(65).astype(np.float32)
    ft, wt, bt = _t(f).requires_grad_(True), _t(w).requires_grad_(True), _t(b).requires_grad_(True)
    out = HF.LinearFn.apply(ft, wt, bt)
    go = rs.standard_normal((37, 65)).astype(np.float32)
    out.backward(_t(go))
    _cmp("logits", out.detach().cpu().numpy(), f.astype(np.float64) @ w.T + b, 1e-5, 1e-5)
    _cmp("dW", wt.grad.cpu().numpy(), go.T.astype(np.float64) @ f, 1e-4, 1e-5)
    _cmp("db", bt.grad.cpu().numpy(), go.sum(0), 1e-5, 1e-5)
    _cmp("dfeat", ft.grad.cpu().numpy(), go.astype(np.float64) @ w, 1e-5, 1e-5)


def test_validate_sets_top1_and_top5_with_ties():
    """validate over several query sets in one launch (language_eval.py:321-326 -> validate :18-43): top-1 and top-5 hit counts
    per set against the oracle's stable-sort rule; features are small integers so logits are exact and ties are real.  A
    classifier with fewer than five rows makes every row a top-5 hit."""
    from oracle import loop_ref
    lib = _lib.load()
    rs = np.random.RandomState(5)
    for N, D in ((65, 640), (351, 640), (4, 64)):
        rows = [125, 125, 37]
        n = sum(rows)
        f = rs.randint(-1, 2, (n, D)).astype(np.float32)
        w = rs.randint(-1, 2, (N, D)).astype(np.float32)
        w[1] = w[0]                                               # duplicate rows: exact ties between classes 0 and 1
        if N > 8:
            w[7] = w[3]
        y = rs.randint(0, N, n).astype(np.int64)
        ft, wt, yt = _t(f), _t(w), torch.from_numpy(y).cuda()
        c1 = torch.zeros(2 * 3, dtype=torch.int32, device="cuda")
        c5 = torch.zeros(2 * 3, dtype=torch.int32, device="cuda")
        set_rows = (C.c_int * 3)(*rows)
        _lib.check(lib.subreg_validate_sets(_lib.ptr(ft), _lib.ptr(yt), _lib.ptr(wt), None, set_rows, 3, N, D, None, _lib.ptr(c1),
                                            _lib.ptr(c5), 3, 0, None), "validate_sets")
        _lib.check(lib.subreg_validate_sets(_lib.ptr(ft), _lib.ptr(yt), _lib.ptr(wt), None, set_rows, 3, N, D, None, _lib.ptr(c1),
                                            None, 3, 0, None), "validate_sets without top-5")
        torch.cuda.synchronize()
        logits = f.astype(np.float64) @ w.T.astype(np.float64)
        o = 0
        for j, r in enumerate(rows):
            lg, yy = logits[o:o + r], y[o:o + r]
            want1 = int((np.argmax(lg, 1) == yy).sum())
            want5 = int(round(loop_ref.accuracy_topk(lg, yy, 5) * r / 100.0))
            assert int(c1[j]) == 2 * want1, (N, j, int(c1[j]), want1)           # two launches accumulated
            assert int(c5[j]) == want5, (N, j, int(c5[j]), want5)
            if N < 5:
                assert want5 == r
            o += r
        assert int(c1[3:].sum()) == 0 and int(c5[3:].sum()) == 0


def test_dropblock_rescale_factor_stays_on_the_device():
    """DropBlock's `countM / count_ones` (models/resnet_language.py:318-323) without a host read: subreg_random_keep_mask counts
    the kept elements, subreg_mask_scale turns the counter into the factor, and subreg_bn_apply / subreg_block_tail_bwd read it from
    the device (mask_scale_dev) - same results as with the host float."""
    lib = _lib.load()
    B, H, W, Cc = 3, 6, 5, 64
    n = B * H * W * Cc
    keep = torch.empty(n, dtype=torch.uint8, device=_dev())
    cnt = torch.zeros(1, dtype=torch.int32, device=_dev())
    _lib.check(lib.subreg_random_keep_mask(_lib.ptr(keep), n, 12345, 0.3, _lib.ptr(cnt), _lib.stream_ptr()))
    scale_dev = torch.zeros(1, dtype=torch.float32, device=_dev())
    _lib.check(lib.subreg_mask_scale(_lib.ptr(cnt), n, _lib.ptr(scale_dev), _lib.stream_ptr()))
    kept = int(keep.sum().item())
    assert int(cnt.item()) == kept and 0.6 * n < kept < 0.8 * n
    want = np.float32(n / kept)
    assert scale_dev.item() == want
    rs = np.random.RandomState(4)
    x = torch.from_numpy(rs.standard_normal(n).astype(np.float32)).to(_dev()).to(torch.bfloat16)
    sc, sh = _t(rs.uniform(0.5, 1.5, Cc).astype(np.float32)), _t(rs.standard_normal(Cc).astype(np.float32))
    outs = []
    for dev_scale in (False, True):
        y = torch.empty(n, dtype=torch.bfloat16, device=_dev())
        _lib.check(lib.subreg_bn_apply(_lib.ptr(x), _lib.ptr(sc), _lib.ptr(sh), None, None, None, _lib.ptr(keep),
                                       0.0 if dev_scale else float(want), _lib.ptr(scale_dev) if dev_scale else None, _lib.ptr(y),
                                       B, H, W, Cc, _lib.CONV_LRELU, _lib.BF16, _lib.stream_ptr()))
        outs.append(y.float().cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.abs(outs[0]).max() > 0


@pytest.mark.parametrize("case", [(2, 8, 10, 10, 5, 7), (3, 16, 10, 10, 5, 25), (1, 8, 5, 5, 5, 5), (2, 8, 9, 7, 3, 6), (2, 8, 6, 6, 2, 9),
                                  (1, 8, 10, 10, 5, 0), (4, 64, 10, 10, 5, 400)])
def test_dropblock_block_mask_on_device(case):
    """DropBlock._compute_block_mask (models/resnet_language.py:327-357) for block_size > 1 as a device kernel against the
    oracle's restatement (pinned by blocks.npz against the reference itself), including the seed / offset PAIRING quirk: seed
    counts n with gcd(n, bs^2) = 1 (every offset), = bs (n = 5), = bs^2 (n = 25: one offset per seed), even block sizes,
    no seed at all, and many seeds spread over all 1024 scan chunks."""
    B, Cc, H, W, bs, n_seeds = case
    lib = _lib.load()
    rs = np.random.RandomState(B * 100 + n_seeds)
    shape = (B, Cc, H - bs + 1, W - bs + 1)
    sample = np.zeros(int(np.prod(shape)), np.float32)
    sample[rs.choice(sample.size, n_seeds, replace=False)] = 1.0
    sample = sample.reshape(shape)
    want = rr.dropblock_block_mask(sample, bs)                      # NCHW {0,1}
    sd, keep = torch.from_numpy(sample.astype(np.uint8)).cuda(), torch.full((B * H * W * Cc,), 7, dtype=torch.uint8, device="cuda")
    cnt = torch.full((1,), 123, dtype=torch.int32, device="cuda")
    _lib.check(lib.subreg_dropblock_mask(_lib.ptr(sd), _lib.ptr(keep), B, Cc, H, W, bs, _lib.ptr(cnt), None), "dropblock_mask")
    torch.cuda.synchronize()
    got = keep.cpu().numpy().reshape(B, H, W, Cc).transpose(0, 3, 1, 2)
    assert np.array_equal(got, want.astype(np.uint8)), (case, int((got != want).sum()))
    assert int(cnt[0]) == int(want.sum())


def test_validate_sets_out_of_range_label_is_a_miss():
    """A query label that is not (yet) a row of the classifier counts as wrong for top-1 AND top-5 (it used to index the logits
    out of range in the top-5 path)."""
    lib = _lib.load()
    rs = np.random.RandomState(8)
    N, D, n = 65, 640, 40
    f, w = rs.standard_normal((n, D)).astype(np.float32), rs.standard_normal((N, D)).astype(np.float32)
    y = np.argmax(f @ w.T, 1).astype(np.int64)                       # every row a top-1 (and top-5) hit ...
    y[::4] = N + 3                                                   # ... except these: labels beyond the classifier
    y[1] = -1
    c1 = torch.zeros(1, dtype=torch.int32, device="cuda")
    c5 = torch.zeros(1, dtype=torch.int32, device="cuda")
    ft, yt, wt = _t(f), torch.from_numpy(y).cuda(), _t(w)            # (keep the device buffers alive across the launch)
    _lib.check(lib.subreg_validate_sets(_lib.ptr(ft), _lib.ptr(yt), _lib.ptr(wt), None, (C.c_int * 1)(n), 1, N, D,
                                        None, _lib.ptr(c1), _lib.ptr(c5), 1, 0, None), "validate_sets")
    torch.cuda.synchronize()
    ok = int(((y >= 0) & (y < N)).sum())
    assert int(c1[0]) == ok and int(c5[0]) == ok, (int(c1[0]), int(c5[0]), ok)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsubreg_hip.so")
    with pytest.raises(RuntimeError):
        _lib.load()
