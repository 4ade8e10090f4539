"""GPU parity of the pretraining step (train_supervised.py:205-268: forward in train mode, loss.backward(), SGD step):
backward kernels through the C ABI against the oracle, and the whole step against the golden produced by the
reference's own autograd (tests/golden/train_step.npz).

Tolerances: f32 mode - gradients 1e-3 of the tensor's magnitude (the reference itself accumulates weight gradients over
B*H*W pixels in fp32), BN/activation gradients 2e-4; bf16 mode - 5e-2 of the tensor's magnitude (activations AND
gradient tensors are rounded to bf16 between layers) on the gradient norms and the sampled tensors.
"""
import ctypes as C
import os
from types import SimpleNamespace

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import backward_ref as br, resnet_ref as rr             # noqa: E402
from oracle.resnet_ref import MaskSource                           # noqa: E402
from subreg_hip import _lib, synthetic as syn                       # noqa: E402

from conftest import GOLDEN                                         # noqa: E402
from test_hip_kernels import _cmp, _dev, _nchw_host, _nhwc_dev, _round_bf16, _t   # noqa: E402


def _td(dtype):
    return torch.bfloat16 if dtype == "bf16" else torch.float32


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 21, 21, 64, 96, 3), (3, 10, 10, 160, 64, 1), (2, 9, 7, 32, 32, 3), (4, 5, 5, 320, 320, 3),
                                   (2, 42, 42, 160, 160, 3), (3, 84, 84, 32, 64, 1), (2, 84, 84, 64, 64, 3), (3, 10, 10, 640, 320, 1),
                                   (1, 3, 3, 32, 32, 3), (5, 21, 10, 96, 32, 3), (64, 5, 5, 64, 64, 3), (12, 84, 84, 64, 64, 3)])
def test_conv_wgrad_and_dgrad(shape, dtype):
    # (the two 84x84 64 -> 64 cases send the bf16 dX convolution - dgrad-packed weights, no activation - through the persistent
    # conv64_resident kernels: 2 images = a few tiles, 12 images = every workgroup walks several tiles)
    B, H, W, Cin, Cout, k = shape
    lib = _lib.load()
    rs = np.random.RandomState(11)
    x = rs.standard_normal((B, Cin, H, W)).astype(np.float32)
    dy = rs.standard_normal((B, Cout, H, W)).astype(np.float32)
    w = (rs.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    if dtype == "bf16":
        x, dy, w = _round_bf16(x), _round_bf16(dy), _round_bf16(w)
    dx_ref, dw_ref = br.conv_backward(rr._nhwc(x), w, rr._nhwc(dy))
    dt = _lib.dtype_code(dtype)
    xd, dyd = _nhwc_dev(x, dtype), _nhwc_dev(dy, dtype)
    nsplit = lib.subreg_conv_wgrad_splits(B, H, W, Cin, Cout, k, dt)
    assert nsplit >= 1 and (nsplit == 1 or (dtype == "bf16" and k == 3))
    gw = torch.full((nsplit * Cout * k * k * Cin,), float("nan"), dtype=torch.float32, device=_dev())   # every copy must be written
    grad = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device=_dev())
    pads = [torch.empty(B * (H + 2) * (W + 2) * c, dtype=_td(dtype), device=_dev()) for c in (Cin, Cout)] if dtype == "bf16" else [None, None]
    _lib.check(lib.subreg_conv_wgrad(_lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(gw), _lib.ptr(pads[0]), _lib.ptr(pads[1]), B, H, W, Cin, Cout,
                                     k, dt, _lib.stream_ptr()))
    _lib.check(lib.subreg_unpack_wgrad(_lib.ptr(gw), _lib.ptr(grad), Cout, Cin, k, 0, nsplit, _lib.stream_ptr()))
    torch.cuda.synchronize()
    _cmp("dW", grad.cpu().numpy(), dw_ref, 1e-3 * np.abs(dw_ref).max(), 1e-4)
    # dX = forward kernel on dY with the flipped / transposed weights
    wt, wd = _t(w), torch.empty(Cin * k * k * Cout, dtype=_td(dtype), device=_dev())
    _lib.check(lib.subreg_pack_conv_weight_dgrad(_lib.ptr(wt), _lib.ptr(wd), Cout, Cin, k, dt, _lib.stream_ptr()))
    dx = torch.empty(B * H * W * Cin, dtype=_td(dtype), device=_dev())
    zero = torch.zeros(Cin, device=_dev())
    _lib.check(lib.subreg_conv_fwd(_lib.ptr(dyd), _lib.ptr(wd), _lib.ptr(dx), None, _lib.ptr(zero), None, None, None, None, 0, B, H, W,
                                   Cout, Cin, k, 0, dt, _lib.stream_ptr()))
    got = _nchw_host(dx, B, Cin, H, W, dtype)
    tol = 2e-4 if dtype == "f32" else 1e-2
    _cmp("dX", got, rr._nchw(dx_ref), tol * np.abs(dx_ref).max(), tol)


@pytest.mark.parametrize("shape", [(64, 84, 84, 64, 64, 3), (64, 42, 42, 160, 160, 3), (64, 10, 10, 640, 640, 3), (64, 10, 10, 320, 640, 3)])
def test_conv_wgrad_pretrain_batch(shape):
    """The streaming bf16 dW kernel at the pretraining batch (B = 64, train_supervised.py:205-268): its K = pixels split runs
    over ~2048 workgroups there, which the small cases above never reach.  Oracle: dW only (nine [O, pixels] x [pixels, C]
    GEMMs in float64 on the same bf16-rounded operands)."""
    B, H, W, Cin, Cout, k = shape
    lib = _lib.load()
    rs = np.random.RandomState(13)
    x = _round_bf16(rs.standard_normal((B, H, W, Cin)).astype(np.float32))
    dy = _round_bf16(rs.standard_normal((B, H, W, Cout)).astype(np.float32))
    xp = np.zeros((B, H + 2, W + 2, Cin), np.float32)
    xp[:, 1:-1, 1:-1] = x
    d2 = dy.reshape(-1, Cout).astype(np.float64)
    dw_ref = np.zeros((Cout, Cin, 3, 3))
    for ky in range(3):
        for kx in range(3):
            dw_ref[:, :, ky, kx] = d2.T @ np.ascontiguousarray(xp[:, ky:ky + H, kx:kx + W]).reshape(-1, Cin).astype(np.float64)
    dt = _lib.BF16
    xd, dyd = torch.from_numpy(x).to(_dev(), torch.bfloat16), torch.from_numpy(dy).to(_dev(), torch.bfloat16)
    nsplit = lib.subreg_conv_wgrad_splits(B, H, W, Cin, Cout, k, dt)
    assert nsplit > 1, "the split-K streaming kernel was not selected"
    gw = torch.full((nsplit * Cout * k * k * Cin,), float("nan"), dtype=torch.float32, device=_dev())
    grad = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device=_dev())
    pads = [torch.empty(B * (H + 2) * (W + 2) * c, dtype=torch.bfloat16, device=_dev()) for c in (Cin, Cout)]
    _lib.check(lib.subreg_conv_wgrad(_lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(gw), _lib.ptr(pads[0]), _lib.ptr(pads[1]), B, H, W, Cin,
                                     Cout, k, dt, _lib.stream_ptr()))
    _lib.check(lib.subreg_unpack_wgrad(_lib.ptr(gw), _lib.ptr(grad), Cout, Cin, k, 0, nsplit, _lib.stream_ptr()))
    torch.cuda.synchronize()
    _cmp("dW (B=64)", grad.cpu().numpy(), dw_ref, 1e-3 * np.abs(dw_ref).max(), 1e-4)


@pytest.mark.parametrize("switch", ["SUBREG_WGRAD_PADDED", "SUBREG_WGRAD_1WAVE"])
def test_conv_wgrad_fallback_kernels(switch):
    """The dW routes behind the A/B switches - the four-wave tile kernel on zero-bordered copies (SUBREG_WGRAD_PADDED=1) and the
    one-wave streaming kernel (SUBREG_WGRAD_1WAVE=1) - stay correct: the library reads the switches once per process, so the dW
    tests above run again in a child process with the switch set."""
    import subprocess
    import sys
    env = dict(os.environ)
    env[switch] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k",
                        "test_conv_wgrad_pretrain_batch or (test_conv_wgrad_and_dgrad and bf16)"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("with_act", [False, True])
def test_bn_backward(dtype, with_act):
    B, H, W, Cc = 3, 10, 9, 96
    lib = _lib.load()
    rs = np.random.RandomState(5)
    raw = rs.standard_normal((B, H, W, Cc)) * 1.5 + 0.3
    dy = rs.standard_normal((B, H, W, Cc))
    gamma, beta = rs.uniform(0.5, 1.5, Cc).astype(np.float32), (rs.standard_normal(Cc) * 0.1).astype(np.float32)
    if dtype == "bf16":
        raw, dy = _round_bf16(raw.astype(np.float32)).astype(np.float64), _round_bf16(dy.astype(np.float32)).astype(np.float64)
    y, cache = br.bn_train_forward(raw, gamma, beta)
    act = rr.leaky_relu(y)
    if dtype == "bf16":
        act = _round_bf16(act.astype(np.float32)).astype(np.float64)
    g = br.lrelu_backward(dy, act) if with_act else dy
    dx_ref, dg_ref, db_ref = br.bn_train_backward(g, cache, gamma)
    dt, td = _lib.dtype_code(dtype), _td(dtype)
    dev = _dev()
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev, td)
    rawd, dyd, actd = f(raw), f(dy), f(act)
    mean, invstd = _t(cache[2].astype(np.float32)), _t(cache[1].astype(np.float32))
    npix = B * H * W
    part = torch.empty(lib.subreg_bn_bwd_slices(npix) * Cc * 2, dtype=torch.float64, device=dev)
    dgam, dbet, dx = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(npix * Cc, dtype=td, device=dev)
    gd = _t(gamma)
    _lib.check(lib.subreg_bn_bwd(_lib.ptr(dyd), _lib.ptr(actd) if with_act else None, _lib.ptr(rawd), _lib.ptr(mean), _lib.ptr(invstd),
                                 _lib.ptr(gd), _lib.ptr(part), _lib.ptr(dgam), _lib.ptr(dbet), _lib.ptr(dx), npix, Cc, dt, _lib.stream_ptr()))
    torch.cuda.synchronize()
    tol = 2e-4 if dtype == "f32" else 1e-2
    _cmp("dgamma", dgam.cpu().numpy(), dg_ref, tol * np.abs(dg_ref).max(), tol)
    _cmp("dbeta", dbet.cpu().numpy(), db_ref, tol * np.abs(db_ref).max(), tol)
    _cmp("dx", dx.float().cpu().numpy().reshape(B, H, W, Cc), dx_ref, tol * np.abs(dx_ref).max(), tol)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("pool", [0, 1])
def test_block_tail_backward(pool, dtype):
    """keep mask * scale, MaxPool2d(2) routing to the first maximum (floor mode: 9x7 -> 4x3), LeakyReLU'.  dV starts as NaN:
    the kernel itself must zero the non-maximum pixels AND the row / column floor pooling drops."""
    B, H, W, Cc = 2, 9, 7, 64
    lib = _lib.load()
    rs = np.random.RandomState(8)
    raw3, res = rs.standard_normal((B, H, W, Cc)), rs.standard_normal((B, H, W, Cc))
    raw3[0, 0:2, 0:2, :8] = 0.25                      # ties inside a window: gradient must go to the first element
    res[0, 0:2, 0:2, :8] = 0.0
    sc, sh = rs.uniform(0.5, 1.5, Cc).astype(np.float32), (rs.standard_normal(Cc) * 0.1).astype(np.float32)
    sc[:8], sh[:8] = 1.0, 0.0
    rsc, rsh = rs.uniform(0.5, 1.5, Cc).astype(np.float32), (rs.standard_normal(Cc) * 0.1).astype(np.float32)
    v = raw3 * sc + sh + res * rsc + rsh
    z = rr.leaky_relu(v)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    gout = rs.standard_normal((B, Ho, Wo, Cc))
    if dtype == "bf16":                               # (rounded BEFORE the oracle sees them; v is recomputed from the rounded operands)
        raw3, res, gout = (_round_bf16(a.astype(np.float32)).astype(np.float64) for a in (raw3, res, gout))
        v = raw3 * sc + sh + res * rsc + rsh
        z = rr.leaky_relu(v)
    keep = (rs.random_sample((B, Ho, Wo, Cc)) > 0.2)
    want = br.lrelu_backward(br.maxpool_backward(gout * keep * 1.25, z, 2 if pool else 1), v)
    dev = _dev()
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev, _td(dtype))
    r3, rd, gd, kd = f(raw3), f(res), f(gout), torch.from_numpy(keep.astype(np.uint8)).to(dev)
    scd, shd, rscd, rshd = _t(sc), _t(sh), _t(rsc), _t(rsh)
    dv = torch.full((B * H * W * Cc,), float("nan"), device=dev, dtype=_td(dtype))
    _lib.check(lib.subreg_block_tail_bwd(_lib.ptr(gd), _lib.ptr(kd), 1.25, None, _lib.ptr(r3), _lib.ptr(scd), _lib.ptr(shd), _lib.ptr(rd),
                                         _lib.ptr(rscd), _lib.ptr(rshd), _lib.ptr(dv), B, H, W, Cc, pool, _lib.dtype_code(dtype), _lib.stream_ptr()))
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == "f32" else 1e-2
    _cmp("dv", dv.float().cpu().numpy().reshape(B, H, W, Cc), want, tol, tol)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("pool,shortcut_bn", [(1, True), (0, False), (1, False), (0, True)])
def test_block_tail_with_fused_bn_statistics(pool, shortcut_bn, dtype):
    """subreg_block_tail_bwd_stats + subreg_bn_bwd_partials (the block tail emits the reduce pass of bn3 and of the shortcut's BatchNorm)
    against the three separate passes subreg_block_tail_bwd + 2 x subreg_bn_bwd on the same inputs: dV identical, d gamma / d beta / d x of
    both BatchNorms equal up to the summation order (BasicBlock.forward :288-299 backwards)."""
    B, H, W, Cc = 3, 10, 12, 160
    lib = _lib.load()
    rs = np.random.RandomState(31 + pool + 2 * shortcut_bn)
    dev = _dev()
    td = _td(dtype)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev, td)
    raw3, res = f(rs.standard_normal((B, H, W, Cc))), f(rs.standard_normal((B, H, W, Cc)))
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    gout = f(rs.standard_normal((B, Ho, Wo, Cc)))
    keep = torch.from_numpy((rs.random_sample((B, Ho, Wo, Cc)) > 0.2).astype(np.uint8)).to(dev)
    sc, sh, rsc, rsh = (_t(rs.uniform(0.5, 1.5, Cc).astype(np.float32)), _t((rs.standard_normal(Cc) * 0.1).astype(np.float32)),
                        _t(rs.uniform(0.5, 1.5, Cc).astype(np.float32)), _t((rs.standard_normal(Cc) * 0.1).astype(np.float32)))
    mean3, inv3, g3 = _t(rs.standard_normal(Cc) * 0.1), _t(rs.uniform(0.5, 2.0, Cc)), _t(rs.uniform(0.5, 1.5, Cc))
    meand, invd, gd_ = _t(rs.standard_normal(Cc) * 0.1), _t(rs.uniform(0.5, 2.0, Cc)), _t(rs.uniform(0.5, 1.5, Cc))
    npix, dt = B * H * W, _lib.dtype_code(dtype)
    nsl = lib.subreg_bn_bwd_slices(npix)
    part = lambda: torch.zeros(nsl * Cc * 2, dtype=torch.float64, device=dev)
    new = lambda: torch.full((B * H * W * Cc,), float("nan"), device=dev, dtype=td)
    vec = lambda: torch.empty(Cc, device=dev)
    rsc_p, rsh_p = (_lib.ptr(rsc), _lib.ptr(rsh)) if shortcut_bn else (None, None)
    # separate passes
    dv_a, dx3_a, dxd_a, pa = new(), new(), new(), part()
    dg3_a, db3_a, dgd_a, dbd_a = vec(), vec(), vec(), vec()
    _lib.check(lib.subreg_block_tail_bwd(_lib.ptr(gout), _lib.ptr(keep), 1.25, None, _lib.ptr(raw3), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res),
                                         rsc_p, rsh_p, _lib.ptr(dv_a), B, H, W, Cc, pool, dt, _lib.stream_ptr()))
    _lib.check(lib.subreg_bn_bwd(_lib.ptr(dv_a), None, _lib.ptr(raw3), _lib.ptr(mean3), _lib.ptr(inv3), _lib.ptr(g3), _lib.ptr(pa), _lib.ptr(dg3_a),
                                 _lib.ptr(db3_a), _lib.ptr(dx3_a), npix, Cc, dt, _lib.stream_ptr()))
    if shortcut_bn:
        _lib.check(lib.subreg_bn_bwd(_lib.ptr(dv_a), None, _lib.ptr(res), _lib.ptr(meand), _lib.ptr(invd), _lib.ptr(gd_), _lib.ptr(pa), _lib.ptr(dgd_a),
                                     _lib.ptr(dbd_a), _lib.ptr(dxd_a), npix, Cc, dt, _lib.stream_ptr()))
    # fused statistics
    dv_b, dx3_b, dxd_b, p3, pd = new(), new(), new(), part(), part()
    dg3_b, db3_b, dgd_b, dbd_b = vec(), vec(), vec(), vec()
    slices = C.c_int(0)
    _lib.check(lib.subreg_block_tail_bwd_stats(_lib.ptr(gout), _lib.ptr(keep), 1.25, None, _lib.ptr(raw3), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(res),
                                               rsc_p, rsh_p, _lib.ptr(dv_b), B, H, W, Cc, pool, dt, _lib.ptr(mean3), _lib.ptr(inv3), _lib.ptr(p3),
                                               _lib.ptr(meand) if shortcut_bn else None, _lib.ptr(invd) if shortcut_bn else None,
                                               _lib.ptr(pd) if shortcut_bn else None, C.byref(slices), _lib.stream_ptr()))
    assert 0 < slices.value < nsl
    _lib.check(lib.subreg_bn_bwd_partials(_lib.ptr(dv_b), None, _lib.ptr(raw3), _lib.ptr(mean3), _lib.ptr(inv3), _lib.ptr(g3), _lib.ptr(p3), slices.value,
                                          _lib.ptr(dg3_b), _lib.ptr(db3_b), _lib.ptr(dx3_b), npix, Cc, dt, 0, _lib.stream_ptr()))
    if shortcut_bn:
        _lib.check(lib.subreg_bn_bwd_partials(_lib.ptr(dv_b), None, _lib.ptr(res), _lib.ptr(meand), _lib.ptr(invd), _lib.ptr(gd_), _lib.ptr(pd),
                                              slices.value, _lib.ptr(dgd_b), _lib.ptr(dbd_b), _lib.ptr(dxd_b), npix, Cc, dt, 0, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dv_a.float(), dv_b.float())
    tol = 2e-5 if dtype == "f32" else 1e-2
    pairs = [("dgamma3", dg3_a, dg3_b), ("dbeta3", db3_a, db3_b), ("dx3", dx3_a, dx3_b)]
    if shortcut_bn:
        pairs += [("dgamma_d", dgd_a, dgd_b), ("dbeta_d", dbd_a, dbd_b), ("dx_d", dxd_a, dxd_b)]
    for name, a_, b_ in pairs:
        a_, b_ = a_.float().cpu().numpy(), b_.float().cpu().numpy()
        _cmp(name, b_, a_, tol * max(1.0, float(np.abs(a_).max())), tol)


def _train_net(dtype):
    from subreg_hip.resnet_language import create_model
    from test_hip_loop import make_opt
    net = create_model("resnet18", 60, make_opt(hip_dtype=dtype))
    sd = syn.make_state_dict(71)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net = net.cuda()
    net.mask_source = MaskSource(74)
    return net


# (image size, golden file): the two small cases, and the batch train_supervised.py really runs (configs.py:124: 64 images of
# 84x84 - what bench.py's pretraining leg times: there the HIP path selects its split-K workspace, the dW split targets and the
# two-stream schedule, none of which the 6- and 8-image cases reach)
TRAIN_CASES = [(32, "train_step.npz"), (84, "train_step.npz"), (84, "train_step_b64.npz")]
TRAIN_IDS = ["hw32_B8", "hw84_B6", "hw84_B64"]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", TRAIN_CASES, ids=TRAIN_IDS)
def test_train_step_against_reference_autograd(case, dtype):
    from subreg_hip.train import SGD
    hw, fname = case
    g = np.load(os.path.join(GOLDEN, fname))
    key = "hw%d" % hw
    net = _train_net(dtype)
    x = torch.from_numpy(syn.make_images(72, int(g[key + ".B"]), hw)).cuda()
    y = torch.from_numpy(g[key + ".labels"]).cuda()
    net.train()
    logits = net(x)
    loss = torch.nn.CrossEntropyLoss()(logits, y)
    loss.backward()
    f32 = dtype == "f32"
    _cmp("loss", loss.item(), g[key + ".loss"], 2e-4 if f32 else 5e-2, 2e-4 if f32 else 2e-2)
    grads = {n: p.grad.detach().cpu().numpy() for n, p in net.named_parameters()}
    # LeakyReLU has a kink: one pre-activation within rounding distance of 0 changes its slope from 1 to 0.1 between two
    # fp implementations and perturbs every gradient below it by ~1e-2 in a few channels (measured at hw=84: one element
    # of layer3.1.bn2 with |y| < 1e-6; the fp64 oracle differs from the fp32 reference in the same way at layer3.0).
    # So: element-wise 1e-3 where no such event happened (hw=32, and the layers above the event), L2-relative otherwise.
    # bf16 rounds activations and gradient tensors between all 22 layers: direction (cosine) and norm are gated.
    exact_prefixes = ("classifier", "layer4") if hw == 84 else ("classifier", "layer")
    big = int(g[key + ".B"]) > 16
    if big:
        # B = 64: ten times the pre-activations of the 6-image case, so such kink events happen in every stage (measured against the
        # reference's fp32 autograd: loss equal to 1e-7, EVERY stored tensor within 4.9e-3 relative L2, single elements up to
        # 3.6e-2 of their tensor's maximum - layer4.1.bn2.bias, 1.5e-2 on layer3.1.bn1.bias): per-tensor L2 at 1e-2 for all of them,
        # element-wise only on the classifier (above every event)
        exact_prefixes = ("classifier",)
    for k in g.files:
        if k.startswith(key + ".gnorm."):
            name = k[len(key) + 7:]
            rel = abs(np.linalg.norm(grads[name].astype(np.float64)) - float(g[k])) / max(float(g[k]), 1e-12)
            assert rel < (5e-3 if f32 else 0.25), ("gnorm", name, rel)
        elif k.startswith(key + ".grad."):
            name = k[len(key) + 6:]
            want = g[k].astype(np.float64)
            got = grads[name][:want.shape[0]].astype(np.float64)
            l2 = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-20)
            if f32:
                assert l2 < (1e-2 if big else 1.5e-2), ("l2", name, l2)
                if name.startswith(exact_prefixes):
                    _cmp("grad " + name, got, want, (5e-3 if big else 1e-3) * max(float(np.abs(want).max()), 1e-6), 2e-3)
            # (bf16: the element-wise gate is test_bf16_backward_from_its_own_forward_stash - every tensor within 5e-2 of the
            # oracle's backward over the SAME forward stash.  Against the fp32 reference a one-ulp forward difference flips a
            # MaxPool argmax / LeakyReLU side now and then and re-routes gradient discretely, so here only the loss and the
            # gradient norms above are compared.)
    if not f32 and int(g[key + ".B"]) <= 16:
        # whole step, forward flips INCLUDED: bf16 against the oracle that ROUNDS WHERE THE HIP PATH STORES (oracle/torch_ref.py::train_step_grads: packed input, raw
        # conv outputs, activations, block outputs and the gradients with respect to them in bf16; fp32 accumulation; pinned in
        # its fp32 mode by tests/test_oracle_golden.py).  What is left between the two is the accumulation order (an ulp in a
        # pre-activation flips a LeakyReLU side / MaxPool argmax now and then); measured: conv / classifier gradients cosine
        # 0.94-0.95, L2 0.30-0.34, BatchNorm affine gradients cosine >= 0.916, L2 <= 0.41 (the fp32 reference above: 0.89).
        from oracle import torch_ref
        loss_o, go = torch_ref.train_step_grads(syn.make_state_dict(71), syn.make_images(72, int(g[key + ".B"]), hw), g[key + ".labels"],
                                                MaskSource(74), bf16=True)
        _cmp("loss vs bf16 oracle", loss.item(), loss_o, 2e-2, 5e-3)
        for name, got in grads.items():
            a, b = got.astype(np.float64).ravel(), go[name].astype(np.float64).ravel()
            l2 = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
            cos = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))
            weight_like = ".conv" in name or "downsample.0" in name or name.startswith("classifier")
            assert cos > (0.91 if weight_like else 0.87) and l2 < (0.45 if weight_like else 0.55), ("bf16 oracle", name, cos, l2)
    if f32:
        sdn = net.state_dict()
        for k in ("layer1.0.bn1", "layer4.1.bn3"):
            _cmp(k, sdn[k + ".running_mean"].cpu().numpy(), g["%s.%s.running_mean" % (key, k)], 1e-5, 1e-4)
        opt = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
        opt.step()
        torch.cuda.synchronize()
        _cmp("sgd classifier", net.classifier.weight.detach().cpu().numpy(), g[key + ".after_step.classifier.weight"], 2e-5, 1e-4)
        _cmp("sgd conv1", dict(net.named_parameters())["layer1.0.conv1.weight"].detach().cpu().numpy(),
             g[key + ".after_step.layer1.0.conv1.weight"], 1e-3, 1e-3)    # lr 0.05 x the kink-event gradient perturbation


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_step_351_base_classes_against_oracle(dtype):
    """BASELINE.json configs[4], pretraining leg: the train_supervised.py step with tieredImageNet's 351 base classes
    (train_supervised.py:94) - train-mode forward, CE over 351 logits, full backward.  The reference cannot run tieredImageNet
    (train_supervised.py:74 passes unknown kwargs to its dataset), so parity is against oracle/torch_ref.py::train_step_grads
    (pinned at 60 classes by the reference's own autograd golden, tests/test_oracle_golden.py) with a 351-row classifier."""
    from oracle import torch_ref
    from subreg_hip.resnet_language import create_model
    from test_hip_loop import make_opt
    NC, B, hw = 351, 6, 32
    sd = syn.make_state_dict(81, n_cls=NC)
    net = create_model("resnet18", NC, make_opt(hip_dtype=dtype))
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net = net.cuda()
    net.mask_source = MaskSource(84)
    x = syn.make_images(82, B, hw)
    labels = np.random.RandomState(83).randint(0, NC, B).astype(np.int64)
    labels[0], labels[1] = 0, NC - 1                                 # first and last classifier row
    net.train()
    logits = net(torch.from_numpy(x).cuda())
    assert logits.shape == (B, NC)
    loss = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(labels).cuda())
    loss.backward()
    f32 = dtype == "f32"
    loss_o, go = torch_ref.train_step_grads(sd, x, labels, MaskSource(84), bf16=not f32)
    _cmp("loss", loss.item(), loss_o, 2e-4 if f32 else 2e-2, 2e-4 if f32 else 5e-3)
    grads = {n: p.grad.detach().cpu().numpy() for n, p in net.named_parameters()}
    assert grads["classifier.weight"].shape == (NC, 640)
    for name, got in grads.items():
        a, b = got.astype(np.float64).ravel(), go[name].astype(np.float64).ravel()
        l2 = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        if f32:
            assert l2 < 1.5e-2, ("l2", name, l2)                     # (the LeakyReLU-kink note of the 60-class test applies)
        else:
            cos = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))
            weight_like = ".conv" in name or "downsample.0" in name or name.startswith("classifier")
            assert cos > (0.91 if weight_like else 0.87) and l2 < (0.45 if weight_like else 0.55), ("bf16 oracle", name, cos, l2)
    if f32:
        _cmp("grad classifier", grads["classifier.weight"], go["classifier.weight"],
             1e-3 * float(np.abs(go["classifier.weight"]).max()), 2e-3)


def test_softmax_ce_kernel_matches_torch_and_reference_accuracy():
    """nn.CrossEntropyLoss() (mean) + eval/util.py:26-40 top-1/top-5 in one launch, and its autograd."""
    from subreg_hip import functional as HF
    rs = np.random.RandomState(5)
    for B, N in ((64, 60), (7, 100), (128, 351), (3, 5)):
        z = torch.from_numpy(rs.standard_normal((B, N)).astype(np.float32) * 3).cuda().requires_grad_(True)
        y = torch.from_numpy(rs.randint(0, N, size=B)).cuda()
        cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
        loss = HF.cross_entropy(z, y, cnt, 5)
        (loss * 1.7).backward()
        zr = z.detach().clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(zr, y)
        (ref * 1.7).backward()
        _cmp("ce loss", loss.item(), ref.item(), 1e-5, 1e-5)
        _cmp("ce grad", z.grad.cpu().numpy(), zr.grad.cpu().numpy(), 1e-6, 1e-4)
        _, pred = zr.detach().topk(min(5, N), 1, True, True)                  # the reference's accuracy()
        correct = pred.t().eq(y.view(1, -1).expand_as(pred.t()))
        assert cnt.tolist() == [int(correct[:1].sum()), int(correct[:5].sum())], (B, N, cnt.tolist())


@pytest.mark.parametrize("hw", [32, 84])
def test_pretrain_driver_runs_the_reference_routine(tmp_path, hw):
    """train_supervised.py:150-202 over the HIP train step: LR schedule applied, loss falls on a small fixed batch set,
    periodic + last checkpoints are written in the reference's format and reload into an identical model.  84x84 is the
    reference's image size (the streaming dW kernel, 42x42 / 21x21 maps and the 21 -> 10 floor pooling only occur there)."""
    import argparse
    from subreg_hip import checkpoint as ck, pretrain as pt
    from subreg_hip.resnet_language import create_model
    opt = argparse.Namespace(no_dropblock=True, linear_bias=False, hip_dtype="bf16", model="resnet18", learning_rate=0.004,
                             momentum=0.9, weight_decay=5e-4, lr_decay_epochs=[4], lr_decay_rate=0.1, epochs=5, cosine=False,
                             print_freq=100, save_freq=2, model_path=str(tmp_path / "models"), continual=True, adam=False,
                             label_pull=None, eval_only=False, dataset="miniImageNet")
    torch.manual_seed(0)
    net = create_model("resnet18", 10, opt)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in syn.make_state_dict(3, n_cls=10, randomize_bn=False).items()})
    net = net.cuda()
    rs = np.random.RandomState(2)
    ys = [torch.from_numpy(rs.randint(0, 10, size=16)) for _ in range(3)]
    batches = [(torch.from_numpy((rs.standard_normal((16, 3, hw, hw)) + y.numpy()[:, None, None, None] * 0.5).astype(np.float32)), y,
                torch.arange(16)) for y in ys]

    class _DS(list):
        dataset = SimpleNamespace(basec_map={i: i for i in range(10)}, label2human=["c%d" % i for i in range(10)])
    loader = _DS(batches)
    hist = pt.fit(net, opt, loader, val_loader=loader, log=lambda *a: None)
    assert [round(h["lr"], 6) for h in hist] == [0.004] * 4 + [0.0004]          # milestone 4 passes at epoch 5 (strictly greater)
    assert all(np.isfinite(h["train_loss"]) and np.isfinite(h["test_loss"]) for h in hist)
    assert hist[-1]["train_loss"] < hist[0]["train_loss"]
    assert 0.0 <= hist[-1]["test_acc"] <= hist[-1]["test_acc_top5"] <= 100.0
    files = sorted(os.listdir(opt.model_path))
    assert files == ["ckpt_epoch_2.pth", "ckpt_epoch_4.pth", "resnet18_last.pth"]
    last = ck.load_checkpoint(os.path.join(opt.model_path, "resnet18_last.pth"))
    assert sorted(last.keys()) == ["label2human", "model", "opt", "training_classes"]
    net2 = ck.model_from_checkpoint(last, "resnet18", 10, argparse.Namespace(no_dropblock=True, hip_dtype="bf16")).cuda()
    for (k, a), (_k, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k


def test_fused_multi_tensor_sgd_matches_oracle_over_two_steps():
    """subreg_sgd_momentum_multi (gradients = views of one flat buffer, as BackboneTrainFn.backward returns them) against the
    oracle's torch.optim.SGD restatement: first step (buffer = d) and second step (momentum)."""
    from subreg_hip.train import SGD
    rs = np.random.RandomState(9)
    shapes = [(64, 3, 3, 3), (64,), (64,), (160, 64, 3, 3), (7,), (320, 160, 1, 1)]
    ps = [torch.nn.Parameter(torch.from_numpy(rs.standard_normal(s).astype(np.float32)).cuda()) for s in shapes]
    opt = SGD(ps, lr=0.05, momentum=0.9, weight_decay=5e-4)
    ref_p = [p.detach().cpu().numpy().astype(np.float64) for p in ps]
    ref_b = [None] * len(ps)
    for step in range(2):
        sizes = [int(np.prod(s)) for s in shapes]
        flat = torch.from_numpy(rs.standard_normal(sum(sizes)).astype(np.float32)).cuda()
        off = 0
        for p, n, s in zip(ps, sizes, shapes):
            p.grad = flat[off:off + n].view(s)
            off += n
        opt.step()
        off = 0
        for i, (n, s) in enumerate(zip(sizes, shapes)):
            g = flat[off:off + n].view(s).cpu().numpy().astype(np.float64)
            ref_p[i], ref_b[i] = br.sgd_momentum_step(ref_p[i], g, ref_b[i], 0.05, 0.9, 5e-4)
            off += n
    assert opt._multi, "the fused path was not taken"
    for p, r in zip(ps, ref_p):
        _cmp("multi sgd", p.detach().cpu().numpy(), r, 1e-6, 1e-5)


def test_fused_sgd_and_repack_equals_separate_update_and_repack():
    """subreg_sgd_pack_train (conv weights: SGD + raw / dX re-packing in one launch, taken when .grad are views of the train
    stash's flat buffer) against the per-tensor SGD kernel + subreg_backbone_pack_train: after one optimiser step the
    parameters and momentum buffers agree to 1e-5 of their scale (same formula; the gradients themselves carry atomics noise) and
    the next train-mode forward - which reads the packed copies each path wrote - to bf16 rounding; a second step runs."""
    from subreg_hip.train import SGD
    nets, opts, feats = [], [], []
    x = torch.from_numpy(syn.make_images(5, 6, 32)).cuda()
    y = torch.from_numpy(np.random.RandomState(6).randint(0, 60, 6)).cuda()
    for fused in (True, False):
        net = _train_net("bf16")
        opt = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)

        def step():
            net.train()
            loss = torch.nn.CrossEntropyLoss()(net(x), y)
            opt.zero_grad()
            loss.backward()
            if not fused:                                     # unrelated tensors: per-tensor kernels, repack at the next forward
                for p in net.parameters():
                    p.grad = p.grad.clone()
            opt.step()
        net.mask_source = MaskSource(3)
        step()
        assert bool(opt._stash_mom) == fused, "fused path %staken" % ("not " if fused else "")
        torch.cuda.synchronize()
        nets.append({n: p.detach().clone() for n, p in net.named_parameters()})
        opts.append([b.clone() for b in opt.bufs])
        net.mask_source = MaskSource(4)
        feats.append(net.features(x).detach().cpu().numpy())
        net.mask_source = MaskSource(5)
        step()                                                # a second step through the same path (momentum term)
        assert all(torch.isfinite(p).all() for p in net.parameters())
    # (the two runs' GRADIENTS already differ in the last bits: the 1x1 / first-layer dW kernels accumulate with float atomics)
    for n in nets[0]:
        a, b = nets[0][n].cpu().numpy(), nets[1][n].cpu().numpy()
        _cmp(n, a, b, 1e-5 * max(float(np.abs(b).max()), 1e-3), 1e-5)
    for ba, bb in zip(opts[0], opts[1]):
        a, b = ba.cpu().numpy(), bb.cpu().numpy()
        _cmp("momentum buffer", a, b, 1e-5 * max(float(np.abs(b).max()), 1e-6), 1e-5)
    _cmp("features after the step", feats[0], feats[1], 2e-3 * np.abs(feats[1]).max(), 2e-3)


def _grads_of(net, x, y, zero=True):
    hb = net.hip_backbone()
    hb.nbt = [0] * len(hb.nbt)                      # same DropBlock rate and ...
    net.mask_source = MaskSource(74)                # ... the same masks for every forward of this test
    if zero:
        for p in net.parameters():
            p.grad = None
    net.train()
    torch.nn.CrossEntropyLoss()(net(x), y).backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters()}


def test_second_backward_accumulates_like_autograd():
    """loss.backward() twice without zero_grad() must ADD the second gradient (torch autograd semantics, which
    train_supervised.py:242-244 relies on only through zero_grad); `.grad` is a view of the stash's flat buffer, which the
    second backward overwrites - the old values have to be taken out first.  Also zero_grad(set_to_none=False)."""
    net = _train_net("f32")
    rs = np.random.RandomState(3)
    x1, x2 = (torch.from_numpy(syn.make_images(80 + i, 6, 32)).cuda() for i in range(2))
    y1, y2 = (torch.from_numpy(rs.randint(0, 60, 6)).cuda() for _ in range(2))
    g1 = _grads_of(net, x1, y1)
    g2 = _grads_of(net, x2, y2)
    _grads_of(net, x1, y1)
    both = _grads_of(net, x2, y2, zero=False)                        # accumulate on top of g1
    for n in g1:
        want = (g1[n] + g2[n]).cpu().numpy()
        _cmp("accumulated " + n, both[n].cpu().numpy(), want, 1e-5 * max(float(np.abs(want).max()), 1e-6), 1e-5)
        assert float((g1[n] - g2[n]).abs().max()) > 0                # the two gradients really differ
    for p in net.parameters():                                        # torch's zero_grad(set_to_none=False): zeroed IN PLACE
        p.grad.zero_()
    again = _grads_of(net, x2, y2, zero=False)
    for n in g2:
        want = g2[n].cpu().numpy()
        _cmp("after in-place zero " + n, again[n].cpu().numpy(), want, 1e-5 * max(float(np.abs(want).max()), 1e-6), 1e-5)
    # with the data-parallel stage hook the flat buffer is all-reduced in place: accumulation is refused loudly
    net.hip_backbone().grad_stage_hook = lambda t: None
    try:
        with pytest.raises(RuntimeError, match="accumulation"):
            _grads_of(net, x1, y1, zero=False)
    finally:
        net.hip_backbone().grad_stage_hook = None


def test_fused_sgd_keeps_momentum_of_earlier_unfused_steps():
    """A step through the per-tensor kernels (gradients not views of the stash) followed by a step through the fused
    subreg_sgd_pack_train launch: the flat momentum buffer the fused path creates must start from the existing momentum."""
    from subreg_hip.train import SGD
    x = torch.from_numpy(syn.make_images(5, 6, 32)).cuda()
    y = torch.from_numpy(np.random.RandomState(6).randint(0, 60, 6)).cuda()
    after = []
    for second_fused in (True, False):
        # The float atomics of the 1x1 / first-layer dW kernels make two runs differ in the last bit of some step-0 gradients, and
        # that can flip ONE MaxPool argmax / LeakyReLU side in the second forward: in f32 mode a handful of weight-gradient elements
        # then take one of two values (measured over 24 runs: 7-111 elements of one tensor off by 3e-5...7e-5, either way round); in
        # bf16 mode a weight can also move by a bf16 ulp and whole tensors shift by per cents.  So: f32, and the gate is each
        # tensor's L2 difference against what LOSING the momentum would change, lr * 0.9 * |momentum buffer after step 0|.
        net = _train_net("f32")
        opt = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
        for step in range(2):
            _grads_of(net, x, y)
            if step == 0 or not second_fused:
                for p in net.parameters():
                    p.grad = p.grad.clone()
            opt.step()
            if step == 0:
                torch.cuda.synchronize()
                lost = {n: 0.05 * 0.9 * float(torch.linalg.vector_norm(opt.bufs[i].double()))
                        for i, (n, _p) in enumerate(net.named_parameters())}
        assert bool(opt._stash_mom) == second_fused
        torch.cuda.synchronize()
        after.append({n: p.detach().cpu().numpy() for n, p in net.named_parameters()})
    worst = 0.0
    for n in after[0]:
        diff = float(np.linalg.norm(after[0][n].astype(np.float64) - after[1][n].astype(np.float64)))
        ratio = diff / max(lost[n], 1e-30)
        worst = max(worst, ratio)
        assert ratio < 0.05, ("momentum carried", n, diff, lost[n])       # measured over 14 runs: <= 8e-4 (kink flips); lost momentum: 1.0
    print("momentum carry-over: worst (difference / effect of losing the momentum) %.2e" % worst)


def _hip_stash_as_oracle_input(net, B, hw):
    """The HIP train step's own forward stash (bf16 NHWC device buffers of subreg_hip.train.TrainStash + the keep masks) as
    the dict oracle/backward_ref.py::backward_from_stash takes."""
    hb = net.hip_backbone()
    st = hb._train_stash
    out, h = {}, hw
    for bi, (name, _cin, cout, stride, ds, _db) in enumerate(hb.blocks):
        def grab(key, c=cout, hh=None):
            t = st.named[key].float().cpu().numpy()
            return t.reshape(B, hh or h, hh or h, c) if t.size == B * (hh or h) ** 2 * c else t
        d = {}
        for slot, tag in (("conv1", "1"), ("conv2", "2"), ("conv3", "3")) + ((("down", "d"),) if ds else ()):
            d["raw" + tag] = grab((bi, slot, "raw"))
            d["mean" + tag], d["invstd" + tag] = grab((bi, slot, "mean")), grab((bi, slot, "invstd"))
            d["scale" + tag], d["shift" + tag] = grab((bi, slot, "bscale")), grab((bi, slot, "bshift"))
        d["act1"], d["act2"] = grab((bi, "conv1", "act")), grab((bi, "conv2", "act"))
        ho = h // stride
        d["out"] = grab((bi, "out"), hh=ho)
        d["keep"] = hb._keep[bi].cpu().numpy().reshape(B, ho, ho, cout).astype(np.float64) * hb.mask_scale(bi)
        out[name] = d
        h = ho
    return out


# (84, 64): the batch bench.py's pretraining leg times (configs.py:124).  B = 128 (its second timed batch) runs the same kernel
# selection as 64 and costs 9 minutes of NumPy oracle on the GPU box's host: not in the suite (SUBREG_TEST_B128=1 adds it).
@pytest.mark.parametrize("replay", [False, True], ids=["eager", "graph_replay"])
@pytest.mark.parametrize("case", [(32, 0), (84, 0), (84, 64)] + ([(84, 128)] if os.environ.get("SUBREG_TEST_B128") == "1" else []),
                         ids=["hw32_B8", "hw84_B6", "hw84_B64"] + (["hw84_B128"] if os.environ.get("SUBREG_TEST_B128") == "1" else []))
def test_bf16_backward_from_its_own_forward_stash(case, replay):
    """bf16 is the dtype of BASELINE.json configs[2] / [4]: pin its BACKWARD kernels tightly.  The bf16 forward's own stash
    (raw conv outputs, activations, batch statistics, block outputs, keep masks - exactly what the HIP backward reads) goes into
    the oracle's backward (oracle/backward_ref.py::backward_from_stash, pinned on CPU against train_step and the reference's
    autograd), which rounds gradient tensors to bf16 where the HIP path stores them.  Forward rounding can no longer flip a
    LeakyReLU side or a MaxPool argmax between the two (that is what kept the whole-step bf16 gate at cosine 0.9), so every
    conv weight gradient (dW kernels), BatchNorm affine gradient (BN backward), and through them every dX / block-tail kernel
    must agree to 5e-2 relative L2 per tensor (measured 0.5-2e-2)."""
    hw, B = case
    g = np.load(os.path.join(GOLDEN, "train_step.npz"))
    key = "hw%d" % hw
    if B == 0:
        B, labels = int(g[key + ".B"]), g[key + ".labels"]
    else:                                  # the timed batches of bench.py's pretraining leg (84x84, 64 and 128 images)
        labels = np.random.RandomState(73).randint(0, 60, B)
    net = _train_net("bf16")
    x_np = syn.make_images(72, B, hw)
    x = torch.from_numpy(x_np).cuda()
    y = torch.from_numpy(labels).cuda()
    net.train()
    if replay:
        # the step as ONE replayed hipGraph (train.GraphedStep): the stash and the gradients the THIRD call - a replay - leaves behind.
        # Free-running device masks (an injected mask source keeps the step eager) and an optimiser that moves nothing (lr = 0,
        # no decay), so the parameters the oracle reads are the ones the replayed forward used.
        from subreg_hip.train import SGD, GraphedStep
        net.mask_source = None
        crit = torch.nn.CrossEntropyLoss()
        stepper = GraphedStep(net, SGD(net.parameters(), lr=0.0, momentum=0.0, weight_decay=0.0), lambda a, b: crit(net(a), b))
        for _ in range(4):
            loss = stepper(x, y)
        assert stepper.replays == 2
    else:
        loss = torch.nn.CrossEntropyLoss()(net(x), y)
        loss.backward()
    torch.cuda.synchronize()
    stash = _hip_stash_as_oracle_input(net, B, hw)
    sd = {k: v.detach().float().cpu().numpy() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    rb = lambda a: _round_bf16(np.asarray(a, np.float32)).astype(np.float64)      # noqa: E731
    loss_o, go = br.backward_from_stash(sd, stash, _round_bf16(x_np), labels, round_fn=rb, conv_dtype=np.float32 if B > 16 else np.float64)
    _cmp("loss", loss.item(), loss_o, 1e-3, 1e-3)
    worst = ("", 0.0)
    for name, p in net.named_parameters():
        a, b = p.grad.detach().cpu().numpy().astype(np.float64).ravel(), np.asarray(go[name], np.float64).ravel()
        l2 = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        worst = max(worst, (name, l2), key=lambda t: t[1])
        assert l2 < 5e-2, ("bf16 backward vs the oracle on the same stash", name, l2)
    print("worst tensor:", worst)


def _plain_net(dtype, dropblock=False, seed=71):
    from subreg_hip.resnet_language import create_model
    from test_hip_loop import make_opt
    net = create_model("resnet18", 60, make_opt(hip_dtype=dtype, no_dropblock=not dropblock))
    sd = syn.make_state_dict(seed)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return net.cuda()


@pytest.mark.parametrize("case", [("f32", 32, 8, False), ("bf16", 84, 6, True), ("bf16", 84, 64, False)],
                         ids=["f32_hw32_B8", "bf16_hw84_B6_dropblock5", "bf16_hw84_B64"])
def test_graphed_step_replay_equals_the_eager_step_from_the_same_state(case):
    """train.GraphedStep (the pretraining driver's step as ONE replayed hipGraph, train_supervised.py:229-244).  After two eager
    warm-up calls and the capture, the state (parameters, buffers, momentum, forward counters) is saved, ONE step is made by replay,
    the state is restored and the SAME step (same batch, same host generator state: the masks' seeds are drawn from it in the same
    order by both) is made eagerly.  The loss, every BatchNorm parameter / statistic / momentum and the classifier must be IDENTICAL -
    fresh dropout / DropBlock masks with DropBlock's gamma at the forward counter's value, batch statistics, the two backward
    streams, the fused SGD + re-pack all behave as in the eager step; the conv weights, some of whose dW kernels accumulate with
    float atomics (every f32 dW, the bf16 1x1 shortcut and first-layer dW: csrc/backward.hip), agree to that run-to-run noise
    (1e-5 of the tensor; a missed re-pack or a stale mask would be 1e-2).  So every parity statement about the eager step (reference goldens, stash-fed oracle
    backward) holds for the replayed one."""
    from subreg_hip.train import SGD, GraphedStep
    dtype, hw, B, dropblock = case
    crit = torch.nn.CrossEntropyLoss()
    xs = [torch.from_numpy(syn.make_images(300 + i, B, hw)).cuda() for i in range(4)]
    ys = [torch.from_numpy(np.random.RandomState(400 + i).randint(0, 60, B)).cuda() for i in range(4)]
    net = _plain_net(dtype, dropblock).train()
    hb = net.hip_backbone()
    opt = SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    stepper = GraphedStep(net, opt, lambda a, b: crit(net(a), b))
    for i in range(3):
        torch.manual_seed(1000 + i)
        stepper(xs[i], ys[i])
    torch.cuda.synchronize()
    assert stepper.replays == 1 and not any(e["failed"] for e in stepper.entries.values())

    def snapshot():
        return ({k: v.detach().clone() for k, v in net.state_dict().items()}, [b.clone() for b in opt.bufs], list(hb.nbt))
    saved = snapshot()
    torch.manual_seed(1003)
    loss_g = float(stepper(xs[3], ys[3]).item())
    assert stepper.replays == 2
    after_g = snapshot()
    with torch.no_grad():                                                  # back to the saved state, in place
        sd = net.state_dict()
        for k, v in saved[0].items():
            sd[k].copy_(v)
        for b, v in zip(opt.bufs, saved[1]):
            b.copy_(v)
    hb.nbt = list(saved[2])
    torch.manual_seed(1003)
    loss = crit(net(xs[3]), ys[3])
    opt.zero_grad()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    after_e = snapshot()
    assert loss_g == float(loss.item()), (loss_g, float(loss.item()))
    assert after_g[2] == after_e[2]
    worst = 0.0
    for (k, g), e in list(zip(after_g[0].items(), after_e[0].values())) + [(("momentum %d" % i, g), e) for i, (g, e) in enumerate(zip(after_g[1], after_e[1]))]:
        # conv weights / their momentum: some dW kernels accumulate with float atomics (every f32 one, the bf16 1x1 and K = 32 first-layer
        # ones, the fallback for shapes the streaming kernel does not take) - those agree to their run-to-run noise, the rest exactly
        atomics = g.dim() == 4
        if atomics:
            rel = float((g.double() - e.double()).norm() / max(float(e.double().norm()), 1e-30))
            worst = max(worst, rel)
            assert rel < 1e-5, ("replayed vs eager step (atomic dW)", k, rel)
        else:
            assert torch.equal(g, e), ("replayed vs eager step", k, float((g.double() - e.double()).abs().max()))
    print("atomic-dW tensors: worst relative difference %.2e" % worst)
    # a learning-rate change makes a second graph (two eager calls, then its capture); the first stays cached.  `loss` - the eager
    # step's, with its autograd graph and the leaves' AccumulateGrad nodes of the DEFAULT stream - is still alive here on purpose:
    # a capture that went through loss.backward() re-used those nodes and crashed in hipStreamEndCapture (GraphedStep._eager)
    assert loss.grad_fn is not None
    opt.lr = 0.02
    for i in range(3):
        stepper(xs[i], ys[i])
    torch.cuda.synchronize()
    assert len(stepper.entries) == 2 and stepper.replays == 3 and not any(e["failed"] for e in stepper.entries.values())


@pytest.mark.gpu
def test_pretrain_epoch_with_the_step_as_one_hipgraph_equals_the_eager_epoch():
    """subreg_hip.pretrain.train (train_supervised.py:205-268) over one epoch of five batches (four of 8 images and a short last one of
    5: two graph keys), f32, once with the eager launches (the default) and once with opt.hip_graph = True (train.GraphedStep: loss +
    accuracy counters inside the graph, two eager warm-up calls, the capture, two replays; the short batch runs once, eagerly).  Same
    start, same batches, same host generator state (the dropout masks' seeds are drawn from it in the same order by both forms).
    This is the driver-level check - counters, meters, graph keys, the optimiser's state across the two forms of the step.  It is NOT
    tight: the float atomics of the f32 dW kernels leave last bits run-dependent, and chained steps carry that across LeakyReLU sides
    and MaxPool argmaxes - two EAGER epochs from the same seeds already differ by 7e-4 (3 steps) ... 2e-2 (7 steps) on the 1728
    elements of layer1.0.conv1.weight and by 1e-5 ... 2e-4 on layer 4 (tools/probes/pretrain_modes_probe.py).  The tight statement
    is test_graphed_step_replay_equals_the_eager_step_from_the_same_state."""
    from types import SimpleNamespace
    from subreg_hip import pretrain as pt
    from subreg_hip.train import SGD
    sizes = [8, 8, 8, 8, 5]
    batches = [(torch.from_numpy(syn.make_images(500 + i, n, 32)), torch.from_numpy(np.random.RandomState(600 + i).randint(0, 60, n)))
               for i, n in enumerate(sizes)]
    results = []
    for use_graph in (False, True):
        net = _plain_net("f32").train()
        opt = SimpleNamespace(print_freq=1000, hip_graph=use_graph, label_pull=None)
        sgd = SGD(net.parameters(), lr=0.002, momentum=0.9, weight_decay=5e-4)
        torch.manual_seed(77)
        acc, loss = pt.train(1, batches, net, None, sgd, opt, log=lambda *_a: None)
        torch.cuda.synchronize()
        if use_graph:
            stepper = sgd._subreg_graphed_step
            assert stepper.replays == 2 and len(stepper.entries) == 2 and not any(e["failed"] for e in stepper.entries.values())
        else:
            assert not hasattr(sgd, "_subreg_graphed_step")
        results.append((float(acc), float(loss), {k: v.detach().clone() for k, v in net.state_dict().items()}, torch.get_rng_state()))
    (acc_e, loss_e, sd_e, rng_e), (acc_g, loss_g, sd_g, rng_g) = results
    assert torch.equal(rng_e, rng_g)                                       # both forms drew the same number of mask seeds
    assert abs(acc_e - acc_g) <= 100.0 / sum(sizes) + 1e-6, (acc_e, acc_g)
    assert abs(loss_e - loss_g) <= 5e-3 * abs(loss_e), (loss_e, loss_g)
    worst = ("", 0.0)
    for k in sd_e:
        a, b = sd_e[k].double(), sd_g[k].double()
        if a.dim() == 0:
            assert torch.equal(a, b), k                                    # num_batches_tracked
            continue
        rel = float((a - b).norm() / a.norm().clamp_min(1e-30))
        worst = max(worst, (k, rel), key=lambda t: t[1])
        tight = k == "classifier.weight" or (k.startswith("layer4.") and "conv" in k)      # (running means sit near zero: loose gate)
        assert rel < (2e-3 if tight else 5e-2), (k, rel)
    print("pretrain epoch, eager vs graph: accuracy %.3f / %.3f, loss %.6f / %.6f, worst tensor %s %.2e" % (acc_e, acc_g, loss_e, loss_g, *worst))


def test_a_step_of_another_batch_shape_makes_the_first_shape_repack_its_weights():
    """A stash's dX weight copies are per stash (TrainStash w_dgrad); the raw forward copies are shared.  A step of ANOTHER batch shape
    (the short last batch of an epoch) moves the weights under the first shape's stash, whose next forward - eager or the one inside a
    replayed graph - must rebuild its copies: train.SGD._fused_conv_step tells every other stash, BackboneTrainFn.forward /
    GraphedStep then launch subreg_backbone_pack_train.  Checked on the mechanism itself (eight chaotic steps of a 5- / 8-image batch
    amplify the atomics' rounding noise to 1e-1 on layer 1, so end-to-end weights say nothing): the copies a forward leaves behind
    must equal a fresh packing of the current weights, and the emulated old behaviour (the mark kept across the other shape's step)
    must NOT - so the check sees the bug."""
    from subreg_hip.train import SGD, GraphedStep, conv_weight_versions
    crit = torch.nn.CrossEntropyLoss()
    xs = {b: torch.from_numpy(syn.make_images(500 + b, b, 32)).cuda() for b in (5, 8)}
    ys = {b: torch.from_numpy(np.random.RandomState(600 + b).randint(0, 60, b)).cuda() for b in (5, 8)}

    def eager_step(net, opt, b):
        loss = crit(net(xs[b]), ys[b])
        opt.zero_grad()
        loss.backward()
        opt.step()

    def copies_are_current(stash):
        """the stash's dX copies against a fresh packing of the parameters as they are now"""
        torch.cuda.synchronize()
        have = [wd.clone() for _c, _co, _ci, _k, wd in stash.dgrad]
        stash.repack_dgrad()
        torch.cuda.synchronize()
        return all(torch.equal(h, wd) for h, (_c, _co, _ci, _k, wd) in zip(have, stash.dgrad))

    for emulate_bug in (False, True):
        net = _plain_net("f32").train()
        hb = net.hip_backbone()
        opt = SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
        eager_step(net, opt, 8)
        stash8 = hb._train_stash
        eager_step(net, opt, 8)
        assert copies_are_current(stash8)                  # (the fused optimiser step wrote them)
        mark = stash8.opt_packed
        assert mark is not None
        eager_step(net, opt, 5)                            # another shape: another stash; the weights move
        assert hb._train_stash is not stash8
        if emulate_bug:
            stash8.opt_packed = conv_weight_versions(hb)   # what the optimiser step used to leave behind
        else:
            assert stash8.opt_packed is None
        eager_step(net, opt, 5)
        if emulate_bug:
            stash8.opt_packed = conv_weight_versions(hb)
        torch.manual_seed(1)
        out = net(xs[8])                                   # the forward of the next 8-image step
        assert hb._train_stash is stash8
        current = copies_are_current(stash8)
        assert current != emulate_bug, ("dX weight copies after another shape's step", emulate_bug, current)
        crit(out, ys[8]).backward()
        opt.zero_grad()

    # the replayed graph: its captured forward reads the copies as they are, so GraphedStep re-packs BEFORE the replay when another
    # shape's step has moved the weights - counted on the library entry point
    net = _plain_net("f32").train()
    hb = net.hip_backbone()
    opt = SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
    stepper = GraphedStep(net, opt, lambda a, b, _n=net: crit(_n(a), b))
    for _ in range(4):
        stepper(xs[8], ys[8])
    assert stepper.replays == 2
    calls, real = [], hb.lib.subreg_backbone_pack_train

    class Counting:
        def __call__(self, *a):
            calls.append(1)
            return real(*a)
    hb.lib.subreg_backbone_pack_train = Counting()
    try:
        stepper(xs[8], ys[8])
        assert not calls                                   # nothing moved the weights but the graph's own optimiser step
        eager_step(net, opt, 5)                            # first sight of the 5-image shape: packs for ITS stash
        n_before = len(calls)
        stepper(xs[8], ys[8])
        assert len(calls) == n_before + 1, (n_before, len(calls))
        stepper(xs[8], ys[8])
        assert len(calls) == n_before + 1
    finally:
        hb.lib.subreg_backbone_pack_train = real
    torch.cuda.synchronize()
    assert stepper.replays == 5


def test_adam_step_matches_torch_optim_adam():
    """train_supervised.py:128-131 (`--adam`): Adam(lr, weight_decay 5e-4) - the HIP update against torch.optim.Adam on CPU over
    four steps (bias corrections, L2 decay added to the gradient), tensors of the sizes a backbone has."""
    from subreg_hip.train import Adam
    rs = np.random.RandomState(5)
    shapes = [(64, 3, 3, 3), (160,), (351, 640), (320, 160, 1, 1)]
    ref = [torch.nn.Parameter(torch.from_numpy((rs.standard_normal(sh) * 0.1).astype(np.float32))) for sh in shapes]
    dut = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=1e-3, weight_decay=0.0005)
    o_dut = Adam(dut, lr=1e-3, weight_decay=0.0005)
    for step in range(4):
        for pr, pd in zip(ref, dut):
            g = (rs.standard_normal(tuple(pr.shape)) * (0.5 if step % 2 else 5e-4)).astype(np.float32)   # large and tiny gradients
            pr.grad = torch.from_numpy(g)
            pd.grad = torch.from_numpy(g).cuda()
        o_ref.step()
        o_dut.step()
    torch.cuda.synchronize()
    for k, (pr, pd) in enumerate(zip(ref, dut)):
        _cmp("adam tensor %d" % k, pd.detach().cpu().numpy(), pr.detach().numpy(), 2e-7, 2e-6)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_eval_mode_backbone_gradients_against_oracle(dtype):
    """Whole-network fine-tuning with the model in EVAL mode (eval/language_eval.py:242-295 before `freeze_backbone_at`, after the
    first validate() put the model into eval mode): BatchNorm normalises with its running statistics (and does not update them),
    there is no dropout, every parameter still gets a gradient.  Against torch autograd on CPU (oracle/torch_ref.py, eval_mode)."""
    from oracle import torch_ref
    net = _train_net(dtype)
    sd = syn.make_state_dict(71)
    B, hw = 5, 32
    x = syn.make_images(91, B, hw)
    labels = np.random.RandomState(92).randint(0, 60, B).astype(np.int64)
    net.eval()
    rm0 = net.state_dict()["layer2.0.bn2.running_mean"].clone()
    logits = net(torch.from_numpy(x).cuda())
    loss = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(labels).cuda())
    loss.backward()
    f32 = dtype == "f32"
    loss_o, go = torch_ref.train_step_grads(sd, x, labels, None, bf16=not f32, eval_mode=True)
    _cmp("loss", loss.item(), loss_o, 2e-4 if f32 else 2e-2, 2e-4 if f32 else 5e-3)
    assert torch.equal(net.state_dict()["layer2.0.bn2.running_mean"], rm0)                  # eval mode: statistics untouched
    assert int(net.state_dict()["layer1.0.bn1.num_batches_tracked"]) == 0
    for name, p in net.named_parameters():
        a, b = p.grad.detach().cpu().numpy().astype(np.float64).ravel(), go[name].astype(np.float64).ravel()
        l2 = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        if f32:
            assert l2 < 1.5e-2, ("l2", name, l2)
        else:
            cos = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))
            weight_like = ".conv" in name or "downsample.0" in name or name.startswith("classifier")
            assert cos > (0.91 if weight_like else 0.87) and l2 < (0.45 if weight_like else 0.55), ("bf16 oracle", name, cos, l2)
    # the eval-mode forward with a stash gives the features the plain eval-mode forward gives
    with torch.no_grad():
        ref_logits = net(torch.from_numpy(x).cuda())
    _cmp("logits", logits.detach().cpu().numpy(), ref_logits.cpu().numpy(), 1e-4 if f32 else 5e-2, 1e-4 if f32 else 2e-2)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_is_feat_with_a_trainable_backbone_returns_the_stage_features(dtype):
    """forward(x, is_feat=True) (models/resnet_language.py:170-192) while the backbone requires gradients: the four stage outputs
    come from the stash of the gradient-carrying forward (detached copies, announced once by a RuntimeWarning), the logits still
    back-propagate into every parameter.  Model in eval mode, so the no-grad eval forward is the comparison."""
    import warnings
    net = _train_net(dtype).eval()
    x = torch.from_numpy(syn.make_images(93, 4, 32)).cuda()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        feats, logits = net(x, is_feat=True)
    assert any("detached" in str(w.message) for w in rec)
    with torch.no_grad():
        feats_ref, logits_ref = net(x, is_feat=True)
    f32 = dtype == "f32"
    assert len(feats) == len(feats_ref) == 5
    for i, (a_, b_) in enumerate(zip(feats, feats_ref)):
        assert a_.shape == b_.shape and (i == 4 or not a_.requires_grad)
        scale = max(1.0, float(b_.abs().max()))
        _cmp("stage %d" % i, a_.detach().cpu().numpy(), b_.cpu().numpy(), (2e-4 if f32 else 6e-2) * scale, 2e-4 if f32 else 3e-2)
    _cmp("logits", logits.detach().cpu().numpy(), logits_ref.cpu().numpy(), 1e-4 if f32 else 5e-2, 1e-4 if f32 else 2e-2)
    logits.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_two_forwards_before_the_backward_keep_their_own_stash_masks_and_input(mode):
    """eval/language_eval.py:252-258 with a trainable backbone and replay memory: net(support) and net(memory) both run BEFORE
    loss.backward().  Each forward owns its stash, its keep masks and its packed input; the gradients of the joint backward must
    equal the sum of two separate forward -> backward rounds over the same batches and masks (same batch sizes on purpose: the
    case in which the second forward used to overwrite the first one's stash)."""
    import warnings
    xa = torch.from_numpy(syn.make_images(75, 6, 32)).cuda()
    xb = torch.from_numpy(syn.make_images(76, 6, 32)).cuda()
    ga = torch.from_numpy(np.random.RandomState(77).randn(6, 60).astype(np.float32)).cuda()
    gb = torch.from_numpy(np.random.RandomState(78).randn(6, 60).astype(np.float32)).cuda()

    def make():
        net = _train_net("f32")
        net.mask_source = MaskSource(74)
        return net.train() if mode == "train" else net.eval()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        net = make()                                              # joint: A, B, then both backward passes
        oa, ob = net(xa), net(xb)
        assert net.hip_backbone()._train_stashes[0] is not net.hip_backbone()._train_stashes[1]
        torch.autograd.backward([oa, ob], [ga, gb])
        joint = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        stats = {k: v.clone() for k, v in net.state_dict().items() if "running" in k}
        ref = make()                                              # sequential: A forward + backward, then B (accumulating)
        ref(xa).backward(ga)
        ref(xb).backward(gb)
    for n, p in ref.named_parameters():
        a, b = joint[n].double().flatten(), p.grad.double().flatten()
        l2 = float((a - b).norm() / b.norm().clamp_min(1e-30))
        assert l2 < 2e-5, (n, l2)                                 # (dW sums are atomic: not bit-identical)
    for k, v in ref.state_dict().items():
        if "running" in k:
            assert torch.allclose(stats[k], v, rtol=1e-6, atol=1e-7), k


def test_a_stash_taken_over_by_later_forwards_makes_its_backward_fail_loudly():
    """At most two stashes per shape are kept (a forward whose backward never comes must not pile up buffers): the third pending
    forward of one shape takes over the oldest stash, and THAT forward's backward must raise instead of differentiating through
    another batch's activations."""
    net = _train_net("f32").eval()
    x = torch.from_numpy(syn.make_images(79, 3, 32)).cuda()
    outs = [net(x) for _ in range(3)]
    assert len([st for st in net.hip_backbone()._train_stashes if st.shape == (3, 32, 32)]) == 2
    outs[2].sum().backward()
    outs[1].sum().backward()
    with pytest.raises(RuntimeError, match="taken over"):
        outs[0].sum().backward()
