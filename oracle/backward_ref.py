"""NumPy restatement of one pretraining step's forward+backward (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/train_supervised.py:205-268 (`train`: model.train(); output = model(input);
loss = criterion(output, target); loss.backward()) over models/resnet_language.py (ResNet.forward :170-192,
BasicBlock.forward :268-301) with the autograd rules of the torch ops behind it: Conv2d (cross-correlation),
BatchNorm2d in training mode, LeakyReLU(0.1) (slope at x <= 0), MaxPool2d(2) (gradient to the FIRST maximum of the
window in scan order), F.dropout / DropBlock (mask * scale), AdaptiveAvgPool2d(1), Linear, CrossEntropyLoss(mean).
Parity pinned by tests/golden/train_step.npz (tools/make_golden.py, the reference's own autograd).
"""
import numpy as np

from . import resnet_ref as rr
from .loop_ref import cross_entropy


def conv_backward(x, w_oihw, dy, dtype=np.float64):
    """x [B,H,W,C], w [O,C,k,k], dy [B,H,W,O] -> (dx, dw) in float64 (dtype = np.float32: the BLAS calls in single precision -
    half the time at the pretraining batch; for comparisons gated at 5e-2, never for the pinned CPU tests)."""
    B, H, W, C = x.shape
    O, _, k, _ = w_oihw.shape
    x, dy, w = x.astype(dtype), dy.astype(dtype), w_oihw.astype(dtype)
    dw = np.zeros_like(w)
    if k == 1:
        dw[:, :, 0, 0] = dy.reshape(-1, O).T @ x.reshape(-1, C)
        return (dy.reshape(-1, O) @ w[:, :, 0, 0]).reshape(B, H, W, C), dw
    xp = np.zeros((B, H + 2, W + 2, C), dtype)
    xp[:, 1:-1, 1:-1] = x
    dxp = np.zeros_like(xp)
    d2 = dy.reshape(-1, O)
    for ky in range(3):
        for kx in range(3):
            xs = np.ascontiguousarray(xp[:, ky:ky + H, kx:kx + W]).reshape(-1, C)
            dw[:, :, ky, kx] = d2.T @ xs
            dxp[:, ky:ky + H, kx:kx + W] += (d2 @ w[:, :, ky, kx]).reshape(B, H, W, C)
    return dxp[:, 1:-1, 1:-1], dw


def bn_train_forward(x, gamma, beta):
    x = x.astype(np.float64)
    mean = x.mean(axis=(0, 1, 2))
    var = x.var(axis=(0, 1, 2))
    invstd = 1.0 / np.sqrt(var + rr.BN_EPS)
    xhat = (x - mean) * invstd
    return xhat * gamma.astype(np.float64) + beta.astype(np.float64), (xhat, invstd, mean, var, x.shape[0] * x.shape[1] * x.shape[2])


def bn_train_backward(dy, cache, gamma):
    xhat, invstd, _mean, _var, n = cache
    dgamma = (dy * xhat).sum(axis=(0, 1, 2))
    dbeta = dy.sum(axis=(0, 1, 2))
    dx = gamma.astype(np.float64) * invstd / n * (n * dy - dbeta - xhat * dgamma)
    return dx, dgamma, dbeta


def lrelu_backward(dy, pre):
    return dy * np.where(pre > 0, 1.0, rr.LEAK)


def maxpool_backward(dy, z, stride):
    """Gradient to the first maximum of every window (scan order dy-major), zero to the floor-dropped edge."""
    if stride == 1:
        return dy
    B, H, W, C = z.shape
    Ho, Wo = H // 2, W // 2
    dz = np.zeros_like(z, dtype=np.float64)
    win = z[:, :Ho * 2, :Wo * 2].reshape(B, Ho, 2, Wo, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, Ho, Wo, C, 4)
    arg = np.argmax(win, axis=-1)                        # first maximum
    b, i, j, c = np.meshgrid(np.arange(B), np.arange(Ho), np.arange(Wo), np.arange(C), indexing="ij")
    dz[b, 2 * i + arg // 2, 2 * j + arg % 2, c] = dy
    return dz


def train_step(sd, x_nchw, labels, masks=None, n_blocks=(1, 1, 2, 2), block_size=1, nbt=None, stash_out=None):
    """One train-mode forward + backward.  Returns (loss, logits, grads) with grads keyed like the state_dict.
    Updates the BN running statistics in `sd` like the forward does.  stash_out: a dict that receives the forward stash in
    the format `backward_from_stash` takes."""
    masks = masks if masks is not None else rr.OnesMaskSource()
    specs = rr.block_specs(n_blocks)
    nbt = nbt if nbt is not None else {s["name"]: 0 for s in specs}
    f64 = np.float64
    x = rr._nhwc(np.asarray(x_nchw)).astype(f64)
    tape = []
    raws = {}

    def conv_bn(inp, cname, bname):
        w = sd[cname + ".weight"]
        raw = rr.conv_nhwc(inp, w.astype(f64))
        y, cache = bn_train_forward(raw, sd[bname + ".weight"], sd[bname + ".bias"])
        _xh, _inv, mean, var, n = cache
        raws[cname] = (raw, mean, _inv)
        sd[bname + ".running_mean"] = ((1 - rr.BN_MOMENTUM) * sd[bname + ".running_mean"] + rr.BN_MOMENTUM * mean).astype(np.float32)
        sd[bname + ".running_var"] = ((1 - rr.BN_MOMENTUM) * sd[bname + ".running_var"] + rr.BN_MOMENTUM * var * n / (n - 1)).astype(np.float32)
        return y, (inp, cname, bname, cache)

    for spec in specs:
        name = spec["name"]
        nbt[name] += 1
        y1, c1 = conv_bn(x, name + ".conv1", name + ".bn1")
        t1 = rr.leaky_relu(y1)
        y2, c2 = conv_bn(t1, name + ".conv2", name + ".bn2")
        t2 = rr.leaky_relu(y2)
        y3, c3 = conv_bn(t2, name + ".conv3", name + ".bn3")
        cd = None
        if spec["downsample"]:
            res, cd = conv_bn(x, name + ".downsample.0", name + ".downsample.1")
        else:
            res = x
        v = y3 + res
        z = rr.leaky_relu(v)
        out = rr.maxpool_nhwc(z, spec["stride"])
        B, H, W, C = out.shape
        if spec["drop_block"]:
            gamma = rr.dropblock_gamma(nbt[name], H, block_size)
            sample = masks.bernoulli((B, C, H - (block_size - 1), W - (block_size - 1)), gamma)
            bm = rr.dropblock_block_mask(sample, block_size)
            m = rr._nhwc(bm).astype(f64) * (bm.size / bm.sum())
        else:
            m = rr._nhwc(masks.dropout_keep((B, C, H, W), rr.DROP_RATE)).astype(f64) / (1.0 - rr.DROP_RATE)
        out = out * m
        tape.append((spec, c1, y1, c2, y2, c3, cd, v, z, m))
        if stash_out is not None:
            st = dict(act1=t1, act2=t2, out=out, keep=m)
            for slot, cn, bn in (("1", ".conv1", ".bn1"), ("2", ".conv2", ".bn2"), ("3", ".conv3", ".bn3"), ("d", ".downsample.0", ".downsample.1")):
                if name + cn in raws:
                    raw, mean, inv = raws[name + cn]
                    st["raw" + slot], st["mean" + slot], st["invstd" + slot] = raw, mean, inv
                    st["scale" + slot] = sd[name + bn + ".weight"].astype(f64) * inv
                    st["shift" + slot] = sd[name + bn + ".bias"].astype(f64) - mean * st["scale" + slot]
            stash_out[name] = st
        x = out
    B, H, W, C = x.shape
    feat = x.mean(axis=(1, 2))
    Wc = sd["classifier.weight"].astype(f64)
    logits = feat @ Wc.T
    loss, dlogits = cross_entropy(logits.astype(np.float32), np.asarray(labels))
    grads = {"classifier.weight": dlogits.T @ feat}
    dx = np.broadcast_to((dlogits @ Wc)[:, None, None, :] / (H * W), (B, H, W, C)).copy()
    for spec, c1, y1, c2, y2, c3, cd, v, z, m in reversed(tape):
        name = spec["name"]
        dz = maxpool_backward(dx * m, z, spec["stride"])
        dv = lrelu_backward(dz, v)

        def back(dy, c):
            inp, cname, bname, cache = c
            draw, dg, db = bn_train_backward(dy, cache, sd[bname + ".weight"])
            dinp, dw = conv_backward(inp, sd[cname + ".weight"], draw)
            grads[cname + ".weight"], grads[bname + ".weight"], grads[bname + ".bias"] = dw, dg, db
            return dinp
        d_t2 = back(dv, c3)
        d_t1 = back(lrelu_backward(d_t2, y2), c2)
        d_in = back(lrelu_backward(d_t1, y1), c1)
        d_in = d_in + (back(dv, cd) if cd is not None else dv)
        dx = d_in
    return float(loss), logits, grads


def sgd_momentum_step(p, g, buf, lr, momentum, weight_decay):
    """torch.optim.SGD (dampening 0, no nesterov): returns (new_p, new_buf); buf None on the first step."""
    d = g + weight_decay * p
    buf = d.copy() if buf is None else momentum * buf + d
    return p - lr * buf, buf


def backward_from_stash(sd, stash, x_nchw, labels, n_blocks=(1, 1, 2, 2), round_fn=None, conv_dtype=np.float64):
    """The backward half of `train_step` started from a GIVEN forward stash instead of its own forward (same autograd rules,
    models/resnet_language.py:268-301 / train_supervised.py:229-244): every decision the backward takes from forward values -
    LeakyReLU side, MaxPool argmax, keep masks, BatchNorm batch statistics - comes from `stash`, so an implementation whose
    forward stash is passed in is compared on its BACKWARD arithmetic alone (a one-ulp forward difference can flip a LeakyReLU
    side or an argmax and re-route gradient discretely; that effect is excluded here by construction).
    stash[name] = dict(raw1, act1, mean1, invstd1, raw2, act2, mean2, invstd2, raw3, mean3, invstd3, scale3, shift3,
                       [rawd, meand, invstdd, scaled, shiftd], out, keep) with NHWC float arrays; keep = mask * scale of the block
    output.  round_fn (e.g. bf16 rounding) is applied where the implementation under test stores a gradient tensor in its
    compute dtype: d(block output), d(pre-activation sum), every BatchNorm dx and every conv dX.  Returns (loss, grads)."""
    r = round_fn if round_fn is not None else (lambda a: a)
    specs = rr.block_specs(n_blocks)
    f64 = np.float64
    last = stash[specs[-1]["name"]]["out"].astype(f64)
    B, H, W, C = last.shape
    feat = last.mean(axis=(1, 2))
    Wc = sd["classifier.weight"].astype(f64)
    loss, dlogits = cross_entropy((feat @ Wc.T).astype(np.float32), np.asarray(labels))
    grads = {"classifier.weight": dlogits.T @ feat}
    g = r(np.broadcast_to((dlogits @ Wc)[:, None, None, :] / (H * W), (B, H, W, C)).copy())

    def bn_bwd(dy, raw, mean, invstd, gamma, act):
        gg = dy * np.where(act > 0, 1.0, rr.LEAK) if act is not None else dy
        n = raw.shape[0] * raw.shape[1] * raw.shape[2]
        xhat = (raw.astype(f64) - mean.astype(f64)) * invstd.astype(f64)
        dgamma, dbeta = (gg * xhat).sum(axis=(0, 1, 2)), gg.sum(axis=(0, 1, 2))
        dx = gamma.astype(f64) * invstd.astype(f64) / n * (n * gg - dbeta - xhat * dgamma)
        return r(dx), dgamma, dbeta

    for bi in range(len(specs) - 1, -1, -1):
        spec = specs[bi]
        name, st = spec["name"], stash[spec["name"]]
        xin = stash[specs[bi - 1]["name"]]["out"].astype(f64) if bi > 0 else rr._nhwc(np.asarray(x_nchw)).astype(f64)
        res = st["rawd"].astype(f64) * st["scaled"] + st["shiftd"] if spec["downsample"] else xin
        v = st["raw3"].astype(f64) * st["scale3"] + st["shift3"] + res
        dv = r(lrelu_backward(maxpool_backward(g * st["keep"], rr.leaky_relu(v), spec["stride"]), v))

        def conv_back(slot, cname, bname, dy, inp, act):
            draw, dg, db = bn_bwd(dy, st["raw" + slot], st["mean" + slot], st["invstd" + slot], sd[bname + ".weight"], act)
            dinp, dw = conv_backward(inp, r(sd[cname + ".weight"].astype(f64)), draw, conv_dtype)
            dinp, dw = dinp.astype(f64), dw.astype(f64)
            grads[cname + ".weight"], grads[bname + ".weight"], grads[bname + ".bias"] = dw, dg, db
            return dinp
        d_t2 = r(conv_back("3", name + ".conv3", name + ".bn3", dv, st["act2"].astype(f64), None))
        d_t1 = r(conv_back("2", name + ".conv2", name + ".bn2", d_t2, st["act1"].astype(f64), st["act2"].astype(f64)))
        d_in = conv_back("1", name + ".conv1", name + ".bn1", d_t1, xin, st["act1"].astype(f64))
        if spec["downsample"]:
            d_in = d_in + conv_back("d", name + ".downsample.0", name + ".downsample.1", dv, xin, None)
        else:
            d_in = d_in + dv
        g = r(d_in)
    return float(loss), grads
