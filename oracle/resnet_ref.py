"""NumPy restatement of the reference backbone (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/models/resnet_language.py:
  ResNet.__init__/_make_layer  :101-167   (stage widths 64/160/320/640, block placement)
  ResNet.forward               :170-192
  BasicBlock.forward           :268-301
  DropBlock                    :303-357
  conv3x3                      :402-405
and torch.nn.{Conv2d,BatchNorm2d,LeakyReLU,MaxPool2d,AdaptiveAvgPool2d,Linear}
semantics (cross-correlation, biased batch variance for normalisation, unbiased
for the running estimate, momentum 0.1, eps 1e-5, floor max-pooling).

Public tensors are NCHW like the reference; internally activations are NHWC so a
3x3 convolution is nine [pixels,Cin]x[Cin,Cout] GEMMs (tap accumulation).
Parity pinned by tests/golden/* (tools/make_golden.py).
"""
import numpy as np

LEAK = 0.1          # nn.LeakyReLU(0.1)              resnet_language.py:251
BN_EPS = 1e-5       # nn.BatchNorm2d default
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default
DROP_RATE = 0.1     # models/util.py:15-18 (drop_rate=0.1 hard-wired by create_model)
WIDTHS = (64, 160, 320, 640)   # resnet_language.py:112-122


def block_specs(n_blocks=(1, 1, 2, 2)):
    """(name, cin, cout, stride, has_downsample, drop_block) per BasicBlock.

    resnet_language.py:142-167: the first block of each stage has stride 2 and a
    1x1 conv+BN shortcut; `drop_block=True` reaches only the LAST block of a
    stage, and only for stages 3 and 4 (:119-122).  For a multi-block stage the
    first block receives `use_se` in the drop_block slot (:155) => False.
    """
    specs = []
    cin = 3
    for si, (nb, planes) in enumerate(zip(n_blocks, WIDTHS)):
        stage_db = si >= 2
        for bi in range(nb):
            first = bi == 0
            if nb == 1:
                db = stage_db
            else:
                db = stage_db and (bi == nb - 1) and not first
            specs.append(dict(name="layer%d.%d" % (si + 1, bi), cin=cin if first else planes,
                              cout=planes, stride=2 if first else 1, downsample=first,
                              drop_block=db))
        cin = planes
    return specs


# ----------------------------------------------------------------------------- primitives (NHWC)
def conv_nhwc(x, w_oihw):
    """Bias-free stride-1 'same' cross-correlation.  x [B,H,W,Cin]; w [O,Cin,k,k] (k in {1,3})."""
    B, H, W, C = x.shape
    O, C2, k, _ = w_oihw.shape
    assert C == C2
    if k == 1:
        return (x.reshape(-1, C) @ w_oihw.reshape(O, C).T.astype(x.dtype)).reshape(B, H, W, O)
    assert k == 3
    xp = np.zeros((B, H + 2, W + 2, C), dtype=x.dtype)
    xp[:, 1:-1, 1:-1, :] = x
    out = np.zeros((B * H * W, O), dtype=x.dtype)
    for dy in range(3):
        for dx in range(3):
            a = np.ascontiguousarray(xp[:, dy:dy + H, dx:dx + W, :]).reshape(-1, C)
            out += a @ np.ascontiguousarray(w_oihw[:, :, dy, dx].T).astype(x.dtype)
    return out.reshape(B, H, W, O)


def bn_eval_nhwc(x, weight, bias, running_mean, running_var):
    inv = 1.0 / np.sqrt(running_var.astype(x.dtype) + x.dtype.type(BN_EPS))
    return (x - running_mean.astype(x.dtype)) * (inv * weight.astype(x.dtype)) + bias.astype(x.dtype)


def bn_train_nhwc(x, weight, bias, running_mean, running_var):
    """Returns (y, new_running_mean, new_running_var); statistics over (B,H,W)."""
    n = x.shape[0] * x.shape[1] * x.shape[2]
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=(0, 1, 2))
    var = x64.var(axis=(0, 1, 2))                       # biased, used to normalise
    inv = 1.0 / np.sqrt(var + BN_EPS)
    y = ((x64 - mean) * (inv * weight.astype(np.float64)) + bias.astype(np.float64)).astype(x.dtype)
    unbiased = var * (n / max(n - 1, 1))
    new_rm = ((1 - BN_MOMENTUM) * running_mean.astype(np.float64) + BN_MOMENTUM * mean).astype(np.float32)
    new_rv = ((1 - BN_MOMENTUM) * running_var.astype(np.float64) + BN_MOMENTUM * unbiased).astype(np.float32)
    return y, new_rm, new_rv


def leaky_relu(x):
    return np.where(x >= 0, x, x * x.dtype.type(LEAK))


def maxpool_nhwc(x, stride):
    """nn.MaxPool2d(stride): kernel=stride, floor mode (21 -> 10 drops the last row/col)."""
    if stride == 1:
        return x
    B, H, W, C = x.shape
    Ho, Wo = H // stride, W // stride
    v = x[:, :Ho * stride, :Wo * stride, :].reshape(B, Ho, stride, Wo, stride, C)
    return v.max(axis=(2, 4))


def dropblock_block_mask(sample_nchw, block_size):
    """DropBlock._compute_block_mask (resnet_language.py:327-357). sample [B,C,H-bs+1,W-bs+1] in {0,1}.

    Reference quirk restated on purpose: the n non-zero indices are tiled as
    nz.repeat(bs^2, 1) (:345) while the bs^2 offsets are tiled as
    offsets.repeat(n, 1) (:346), so row i pairs nz[i % n] with offsets[i % bs^2].
    Every (seed, offset) pair is produced only when gcd(n, bs^2) == 1; otherwise the
    seed of rank r only receives the offsets o with o % g == r % g, g = gcd(n, bs^2).
    """
    bs = block_size
    lp, rp = int((bs - 1) / 2), int(bs / 2)
    padded = np.pad(sample_nchw, ((0, 0), (0, 0), (lp, rp), (lp, rp))).astype(np.float32)
    nz = np.argwhere(sample_nchw != 0)                     # lexicographic (b,c,i,j) like Tensor.nonzero()
    n = nz.shape[0]
    if n:
        i = np.arange(bs * bs * n)
        seeds = nz[i % n]
        o = i % (bs * bs)
        padded[seeds[:, 0], seeds[:, 1], seeds[:, 2] + o // bs, seeds[:, 3] + o % bs] = 1.0
    return 1.0 - padded


def dropblock_gamma(num_batches_tracked, feat_size, block_size, drop_rate=DROP_RATE):
    """BasicBlock.forward :294-296 (python-float arithmetic)."""
    keep_rate = max(1.0 - drop_rate / (20 * 2000) * num_batches_tracked, 1.0 - drop_rate)
    return (1 - keep_rate) / block_size ** 2 * feat_size ** 2 / (feat_size - block_size + 1) ** 2


class MaskSource:
    """Deterministic stand-in for the two RNG streams of the train-mode forward.

    The reference draws dropout masks from the device Philox stream (F.dropout,
    :299) and DropBlock samples from the CPU default generator (:317-318); neither
    is reproducible across devices, so goldens, oracle and HIP path all take their
    masks from this source (NCHW draw order, one RandomState).
    """

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def dropout_keep(self, shape_nchw, p):
        return (self.rs.random_sample(shape_nchw) >= p).astype(np.float32)

    def bernoulli(self, shape_nchw, gamma):
        return (self.rs.random_sample(shape_nchw) < gamma).astype(np.float32)


class OnesMaskSource:
    """No stochastic dropping (keep everything)."""

    def dropout_keep(self, shape_nchw, p):
        return np.ones(shape_nchw, np.float32)

    def bernoulli(self, shape_nchw, gamma):
        return np.zeros(shape_nchw, np.float32)


def _nchw(a_nhwc):
    return np.ascontiguousarray(a_nhwc.transpose(0, 3, 1, 2))


def _nhwc(a_nchw):
    return np.ascontiguousarray(a_nchw.transpose(0, 2, 3, 1))


# ----------------------------------------------------------------------------- model
class ResNetRef:
    """State-dict driven restatement of models.resnet_language.ResNet (+BasicBlock).

    `sd` maps the reference's state_dict key names to numpy arrays (133 keys for
    resnet18 without classifier bias).  BN running stats in `sd` are UPDATED IN
    PLACE by train-mode forwards, like the reference's buffers.
    """

    def __init__(self, sd, n_blocks=(1, 1, 2, 2), block_size=1, dtype=np.float32):
        self.sd = sd
        self.specs = block_specs(n_blocks)
        self.block_size = block_size      # --no_dropblock => 1 (resnet_language.py:116-118), else 5
        self.dtype = dtype
        self.training = False
        self.nbt = {s["name"]: 0 for s in self.specs}   # BasicBlock.num_batches_tracked (python int, :260,269)

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    # -- one BN layer in the current mode
    def _bn(self, x, prefix):
        sd = self.sd
        if self.training:
            y, rm, rv = bn_train_nhwc(x, sd[prefix + ".weight"], sd[prefix + ".bias"],
                                      sd[prefix + ".running_mean"], sd[prefix + ".running_var"])
            sd[prefix + ".running_mean"] = rm
            sd[prefix + ".running_var"] = rv
            if prefix + ".num_batches_tracked" in sd:
                sd[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
            return y
        return bn_eval_nhwc(x, sd[prefix + ".weight"], sd[prefix + ".bias"],
                            sd[prefix + ".running_mean"], sd[prefix + ".running_var"])

    def block_forward(self, x, spec, masks=None):
        """BasicBlock.forward (resnet_language.py:268-301); x, result NHWC."""
        name = spec["name"]
        sd = self.sd
        self.nbt[name] += 1
        out = conv_nhwc(x, sd[name + ".conv1.weight"].astype(self.dtype))
        out = leaky_relu(self._bn(out, name + ".bn1"))
        out = conv_nhwc(out, sd[name + ".conv2.weight"].astype(self.dtype))
        out = leaky_relu(self._bn(out, name + ".bn2"))
        out = conv_nhwc(out, sd[name + ".conv3.weight"].astype(self.dtype))
        out = self._bn(out, name + ".bn3")
        if spec["downsample"]:
            res = conv_nhwc(x, sd[name + ".downsample.0.weight"].astype(self.dtype))
            res = self._bn(res, name + ".downsample.1")
        else:
            res = x
        out = leaky_relu(out + res)
        out = maxpool_nhwc(out, spec["stride"])
        if self.training:
            masks = masks if masks is not None else OnesMaskSource()
            B, H, W, C = out.shape
            if spec["drop_block"]:
                bs = self.block_size
                gamma = dropblock_gamma(self.nbt[name], H, bs)
                sample = masks.bernoulli((B, C, H - (bs - 1), W - (bs - 1)), gamma)
                bm = dropblock_block_mask(sample, bs)                       # NCHW
                scale = bm.size / bm.sum()
                out = (_nhwc(bm).astype(self.dtype) * out * self.dtype(scale)).astype(self.dtype)
            else:
                keep = masks.dropout_keep((B, C, H, W), DROP_RATE)
                out = (out * _nhwc(keep).astype(self.dtype) / self.dtype(1.0 - DROP_RATE)).astype(self.dtype)
        return out

    def features(self, x_nchw, masks=None, return_stages=False):
        """[B,3,H,W] -> feat [B,640] (AdaptiveAvgPool2d(1) + view, :179-182)."""
        x = _nhwc(np.asarray(x_nchw).astype(self.dtype))
        stages = []
        for spec in self.specs:
            x = self.block_forward(x, spec, masks)
            stages.append(x)
        feat = x.mean(axis=(1, 2), dtype=np.float64).astype(self.dtype)
        if return_stages:
            return feat, stages
        return feat

    def forward(self, x_nchw, masks=None):
        feat = self.features(x_nchw, masks)
        return linear(feat, self.sd["classifier.weight"], self.sd.get("classifier.bias"))


def linear(feat, weight, bias=None):
    out = feat @ weight.astype(feat.dtype).T
    if bias is not None:
        out = out + bias.astype(feat.dtype)
    return out


def copy_state_dict(sd):
    return {k: np.array(v, copy=True) for k, v in sd.items()}


FLOPS_PER_IMAGE_84 = 8.1219e9   # 22 convs, 2*MAC, 84x84 input (probed)


def conv_flops_per_image(hw=84, n_blocks=(1, 1, 2, 2)):
    """Forward conv FLOPs (2*MAC) of one image; 8.1219e9 for hw=84 resnet18."""
    total = 0
    h = hw
    for s in block_specs(n_blocks):
        total += 2 * h * h * s["cout"] * s["cin"] * 9
        total += 2 * 2 * h * h * s["cout"] * s["cout"] * 9
        if s["downsample"]:
            total += 2 * h * h * s["cout"] * s["cin"]
        h = h // s["stride"]
    return total
