"""Seeded synthetic inputs with the shapes of the miniImageNet 5-way 5-shot FSCIL episodes.

There is no dataset and no pretrained checkpoint in the build container or on the
GPU box, so weights, images and episode label layouts are drawn from NumPy
RandomStates (identical on every machine).  Shapes/label layouts follow
/root/reference/dataset/mini_imagenet.py:273-350 (MetaImageNet episode: 5 ways x
5 shots x 5 augmented copies = 125 support rows ordered tile(repeat(cls,5),5);
5 x 25 = 125 query rows ordered repeat(cls,25)) and
eval_incremental.py:53-77 (base test batch; one base exemplar per class).
"""
import numpy as np

WIDTHS = (64, 160, 320, 640)


def backbone_blocks(n_blocks=(1, 1, 2, 2)):
    """[(name, cin, cout, stride, has_downsample, drop_block)], models/resnet_language.py:142-167."""
    out, cin = [], 3
    for si, (nb, planes) in enumerate(zip(n_blocks, WIDTHS)):
        for bi in range(nb):
            first = bi == 0
            drop_block = si >= 2 and (nb == 1 or (bi == nb - 1 and not first))
            out.append((("layer%d.%d" % (si + 1, bi)), cin if first else planes, planes,
                        2 if first else 1, first, drop_block))
        cin = planes
    return out


def make_state_dict(seed, n_cls=60, n_blocks=(1, 1, 2, 2), randomize_bn=True, classifier_std=0.05):
    """Synthetic backbone with the reference's state_dict key names (133 keys for resnet18).

    Conv: kaiming-normal(fan_out, leaky_relu) like resnet_language.py:131-136
    (gain sqrt(2/(1+0.01^2)), std = gain/sqrt(Cout*k*k)).  BN affine and running
    statistics are randomised so BN folding is exercised non-trivially.
    """
    rs = np.random.RandomState(seed)
    sd = {}
    gain = np.sqrt(2.0 / (1.0 + 0.01 ** 2))

    def conv(name, o, c, k):
        sd[name] = (rs.standard_normal((o, c, k, k)) * (gain / np.sqrt(o * k * k))).astype(np.float32)

    def bn(prefix, c):
        if randomize_bn:
            sd[prefix + ".weight"] = rs.uniform(0.5, 1.5, c).astype(np.float32)
            sd[prefix + ".bias"] = (rs.standard_normal(c) * 0.1).astype(np.float32)
            sd[prefix + ".running_mean"] = (rs.standard_normal(c) * 0.1).astype(np.float32)
            sd[prefix + ".running_var"] = rs.uniform(0.5, 1.5, c).astype(np.float32)
        else:
            sd[prefix + ".weight"] = np.ones(c, np.float32)
            sd[prefix + ".bias"] = np.zeros(c, np.float32)
            sd[prefix + ".running_mean"] = np.zeros(c, np.float32)
            sd[prefix + ".running_var"] = np.ones(c, np.float32)
        sd[prefix + ".num_batches_tracked"] = np.array(0, np.int64)

    for name, cin, cout, _stride, ds, _db in backbone_blocks(n_blocks):
        conv(name + ".conv1.weight", cout, cin, 3)
        bn(name + ".bn1", cout)
        conv(name + ".conv2.weight", cout, cout, 3)
        bn(name + ".bn2", cout)
        conv(name + ".conv3.weight", cout, cout, 3)
        bn(name + ".bn3", cout)
        if ds:
            conv(name + ".downsample.0.weight", cout, cin, 1)
            bn(name + ".downsample.1", cout)
    sd["classifier.weight"] = (rs.standard_normal((n_cls, 640)) * classifier_std).astype(np.float32)
    return sd


def make_images(seed, n, hw=84):
    """[n,3,hw,hw] fp32, per-channel-normalised-like (mean 0 / std 1, dataset/transform_cfg.py:8-10)."""
    return np.random.RandomState(seed).standard_normal((n, 3, hw, hw)).astype(np.float32)


def session_labels(s, n_ways=5, n_shots=5, n_aug=5, n_queries=25, first_novel=60):
    """Original (pre-remap) labels of session s: classes first_novel+5s .. +4."""
    cls = first_novel + n_ways * s + np.arange(n_ways)
    support_ys = np.tile(np.repeat(cls, n_shots), n_aug)
    query_ys = np.repeat(cls, n_queries)
    return support_ys.astype(np.int64), query_ys.astype(np.int64)


def class_prototype(c, hw, grid=0):
    """Seeded mean image of class c.  grid = 0: white noise (every pixel independent); grid = g > 0: a g x g checker of
    constant cells per channel - low-frequency content that survives a randomly initialised backbone's pooling stages, so
    that classes are separable in FEATURE space (white-noise prototypes mostly average out)."""
    rs = np.random.RandomState(777000 + int(c))
    if not grid:
        return rs.standard_normal((3, hw, hw)).astype(np.float32)
    cell = -(-hw // grid)
    return np.kron(rs.standard_normal((3, grid, grid)).astype(np.float32), np.ones((cell, cell), np.float32))[:, :hw, :hw]


def make_sessions(seed, n_sessions, hw=84, class_signal=0.0, first_novel=60, proto_grid=0, hard_queries=0):
    """Episodes for `n_sessions` incremental sessions.

    class_signal > 0 adds a seeded per-class mean image so that accuracies are not
    all at chance (useful for the loop goldens; pure noise gives ~0 % novel accuracy).
    hard_queries = h > 0: the LAST h query images of every class carry the prototype of the session's next class
    (cyclically) under their own label - examples a correct classifier gets wrong by a wide margin, which caps the query
    accuracy at (25 - h) / 25 without putting any image near a decision boundary.
    """
    sessions = []
    for s in range(n_sessions):
        sy, qy = session_labels(s, first_novel=first_novel)
        sx = make_images(seed * 1000 + 2 * s, len(sy), hw)
        qx = make_images(seed * 1000 + 2 * s + 1, len(qy), hw)
        if class_signal:
            cls = np.unique(sy)
            for i, c in enumerate(cls):
                proto = class_prototype(c, hw, proto_grid)
                sx[sy == c] += class_signal * proto
                rows = np.nonzero(qy == c)[0]
                own = rows[:len(rows) - hard_queries] if hard_queries else rows
                qx[own] += class_signal * proto
                if hard_queries:
                    qx[rows[len(rows) - hard_queries:]] += class_signal * class_prototype(cls[(i + 1) % len(cls)], hw, proto_grid)
        sessions.append(dict(support_xs=sx, support_ys=sy, query_xs=qx, query_ys=qy))
    return sessions


def make_base_batch(seed, n, hw=84, n_base=60, class_signal=0.0, proto_grid=0):
    rs = np.random.RandomState(seed + 500000)
    y = rs.randint(0, n_base, n).astype(np.int64)
    x = make_images(seed + 600000, n, hw)
    if class_signal:
        for c in np.unique(y):
            x[y == c] += class_signal * class_prototype(c, hw, proto_grid)
    return x, y


def make_base_support(seed, hw=84, n_base=60, class_signal=0.0, proto_grid=0):
    """One exemplar per base class (--n_base_support_samples 1)."""
    y = np.arange(n_base, dtype=np.int64)
    x = make_images(seed + 700000, n_base, hw)
    if class_signal:
        for c in y:
            x[c] += class_signal * class_prototype(c, hw, proto_grid)
    return x, y


def make_novel_inits(seed, n_sessions, n_ways=5, dim=640):
    """Init rows for augment_base_classifier_(novel_weight=...): U(-1/sqrt(dim), 1/sqrt(dim)).

    Same distribution as nn.Linear's default init (resnet_language.py:216-217) but
    from a NumPy stream, because the reference's draw is coupled to the global
    torch CPU generator state ('RNG-coupled initial novel rows').
    """
    rs = np.random.RandomState(seed + 900000)
    b = 1.0 / np.sqrt(dim)
    return [rs.uniform(-b, b, (n_ways, dim)).astype(np.float32) for _ in range(n_sessions)]


def make_novel_bias_inits(seed, n_sessions, n_ways=5, dim=640):
    """Init values for augment_base_classifier_(novel_bias=...) of a classifier WITH bias: nn.Linear's default bias init
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (resnet_language.py:216-219), from a NumPy stream like make_novel_inits."""
    rs = np.random.RandomState(seed + 910000)
    b = 1.0 / np.sqrt(dim)
    return [rs.uniform(-b, b, (n_ways,)).astype(np.float32) for _ in range(n_sessions)]


def make_classifier_bias(seed, n_cls=60, scale=0.1):
    """classifier.bias of a backbone pretrained without --no_linear_bias (configs.py:177,213): N(0, scale)."""
    return (np.random.RandomState(seed + 920000).standard_normal(n_cls) * scale).astype(np.float32)


def make_linear_map(seed, indim=500, outdim=640, scale=0.002):
    """A synthetic LinearMap state (learn_mapping.py's product, ckpt['mapping_linear_label2image']): (map.weight, map.bias)."""
    rs = np.random.RandomState(seed)
    return (rs.standard_normal((outdim, indim)) * scale).astype(np.float32), (rs.standard_normal((outdim,)) * scale).astype(np.float32)
