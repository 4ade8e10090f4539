#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; CPU, torch 2.10).  The
reference's Python never travels to the GPU box: only the .npz outputs of this
script (data: inputs-as-seeds + expected outputs) are committed.

Shims: Tensor.cuda / Module.cuda -> identity,
torch.cuda.is_available -> True, cwd=/root/reference for the word-embedding pickle
that LangPuller.__init__ loads even in distance2subspace mode.  The two RNG
streams of the train-mode forward (F.dropout, Bernoulli) are replaced by
oracle.resnet_ref.MaskSource so that masks are reproducible anywhere, and
augment_base_classifier_ receives explicit `novel_weight` rows (its own API).

Usage:  python tools/make_golden.py [blocks backbone reg loop32 loop84 ...]
"""
import os
import sys
import time
import types
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "subspace-reg_amd"))
sys.path.insert(0, REF)

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
torch.cuda.is_available = lambda: True
torch.cuda.is_current_stream_capturing = lambda: False    # torch.optim.Adam asks this whenever is_available() says True (--adam golden)
os.chdir(REF)

import models.resnet_language as rl          # noqa: E402  (the reference)
import eval.language_eval as le              # noqa: E402  (the reference)

from oracle.resnet_ref import MaskSource     # noqa: E402
from subreg_hip import synthetic as syn      # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
torch.set_num_threads(8)


# ------------------------------------------------------------------ RNG shims for the reference
class _MaskedF:
    """Stand-in for `F` inside models.resnet_language: dropout from the MaskSource."""

    def __init__(self):
        self.masks = None
        self.pad = torch.nn.functional.pad

    def dropout(self, x, p=0.5, training=True, inplace=False):
        if not training:
            return x
        keep = torch.from_numpy(self.masks.dropout_keep(tuple(x.shape), p))
        return x.mul_(keep * (1.0 / (1.0 - p))) if inplace else x * keep * (1.0 / (1.0 - p))


_F = _MaskedF()
rl.F = _F


class _Bern:
    def __init__(self, gamma):
        self.gamma = gamma

    def sample(self, shape):
        return torch.from_numpy(_F.masks.bernoulli(tuple(shape), self.gamma))


rl.Bernoulli = _Bern


def set_masks(seed):
    _F.masks = MaskSource(seed)


def ref_opt(**kw):
    """Flag values of scripts/continual/slurm_subspace_reg.sh:33-54 + configs.py defaults."""
    o = SimpleNamespace(
        no_dropblock=True, linear_bias=False, temperature=1, word_embed_size=500,
        word_embed_path="word_embeds", dataset="miniImageNet", use_synonyms=False, glove=False,
        track_weights=False, track_label_inspired_weights=False, save_preds_0=False, set_seed=1,
        memory_replay=0, neval_episodes=8, continual=False, n_ways=5, n_shots=5, n_queries=25,
        label_pull=1.0, pulling="regularize", attraction_override="distance2subspace",
        classifier="linear", attention=None, lmbd_reg_transform_w=0.2, lmbd_reg_novel=0.1,
        target_train_loss=0.0, convergence_epsilon=1e-4, stable_epochs=10, max_novel_epochs=1000,
        min_novel_epochs=20, learning_rate=0.002, momentum=0.9, weight_decay=5e-4, adam=False,
        freeze_backbone_at=1, eval_mode="few-shot-incremental-fine-tune", verbose=False)
    o.__dict__.update(kw)
    return o


def ref_net(sd, opt, n_cls=60):
    net = rl.resnet18(avg_pool=True, drop_rate=0.1, dropblock_size=5, num_classes=n_cls, vocab=None, opt=opt)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return net


def t2n(t):
    return t.detach().cpu().numpy().copy()


# ------------------------------------------------------------------ G1: block level
def gen_blocks():
    opt = ref_opt()
    sd = syn.make_state_dict(11)
    out = {}
    hw_in = {"layer1.0": 84, "layer2.0": 42, "layer3.0": 21, "layer3.1": 10, "layer4.0": 10, "layer4.1": 5}
    for bs_name, no_db in (("bs1", True), ("bs5", False)):
        opt.no_dropblock = no_db
        net = ref_net(sd, opt)
        for li, (name, cin, cout, stride, ds, db) in enumerate(syn.backbone_blocks()):
            if bs_name == "bs5" and not db:
                continue                      # block_size only matters for the DropBlock blocks
            blk = dict(net.named_modules())[name]
            B = 1 if name in ("layer1.0", "layer2.0") else 2
            hw = hw_in[name]
            x = np.random.RandomState(100 + li).standard_normal((B, cin, hw, hw)).astype(np.float32)
            key = "%s.%s" % (name, bs_name)
            out[key + ".in_seed"] = np.array(100 + li)
            out[key + ".in_shape"] = np.array(x.shape)
            if bs_name == "bs1":
                blk.eval()
                with torch.no_grad():
                    out[key + ".eval_out"] = t2n(blk(torch.from_numpy(x)))
            # train mode with injected masks; force a large forward counter so DropBlock's gamma is at its cap
            blk.train()
            blk.num_batches_tracked = 39999
            set_masks(200 + li)
            with torch.no_grad():
                out[key + ".train_out"] = t2n(blk(torch.from_numpy(x)))
            out[key + ".mask_seed"] = np.array(200 + li)
            for bn in ("bn1", "bn2", "bn3") + (("downsample.1",) if ds else ()):
                m = dict(blk.named_modules())[bn]
                out["%s.%s.running_mean" % (key, bn)] = t2n(m.running_mean)
                out["%s.%s.running_var" % (key, bn)] = t2n(m.running_var)
            # restore the buffers for the next variant
            net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    np.savez_compressed(os.path.join(GOLD, "blocks.npz"), **out)
    print("blocks.npz", sum(v.nbytes for v in out.values()) / 1e6, "MB")


# ------------------------------------------------------------------ G2: backbone
def gen_backbone():
    opt = ref_opt()
    out = {}
    for hw in (84, 32):
        sd = syn.make_state_dict(21)
        net = ref_net(sd, opt)
        x = syn.make_images(31, 4, hw)
        net.eval()
        with torch.no_grad():
            feats, logits = net(torch.from_numpy(x), is_feat=True)
        out["hw%d.eval_feat" % hw] = t2n(feats[-1])
        out["hw%d.eval_logits" % hw] = t2n(logits)
        out["hw%d.eval_f0_b0" % hw] = t2n(feats[0][0])         # stage-1 output of image 0 (localises a mismatch)
        net.train()
        set_masks(41)
        with torch.no_grad():
            logits = net(torch.from_numpy(x))
        out["hw%d.train_logits" % hw] = t2n(logits)
        for k in ("layer1.0.bn1", "layer2.0.downsample.1", "layer3.1.bn2", "layer4.1.bn3"):
            m = dict(net.named_modules())[k]
            out["hw%d.%s.running_mean" % (hw, k)] = t2n(m.running_mean)
            out["hw%d.%s.running_var" % (hw, k)] = t2n(m.running_var)
        net.eval()
        with torch.no_grad():
            out["hw%d.eval2_logits" % hw] = t2n(net(torch.from_numpy(x)))   # eval AFTER the stats moved
    np.savez_compressed(os.path.join(GOLD, "backbone.npz"), **out)
    print("backbone.npz", sum(v.nbytes for v in out.values()) / 1e6, "MB")


# ------------------------------------------------------------------ G3: regularizers
def gen_reg():
    opt = ref_opt()
    rs = np.random.RandomState(51)
    out = {}
    puller = rl.LangPuller(opt, ["a"], ["b"])
    net = ref_net(syn.make_state_dict(21), opt)
    for case in ("rand", "trained"):
        wb = (rs.standard_normal((60, 640)) * 0.05).astype(np.float32)
        if case == "trained":
            wb += (rs.standard_normal((60, 1)) * rs.standard_normal((1, 640)) * 0.05).astype(np.float32)
        for k in (5, 40):
            w = torch.from_numpy((rs.standard_normal((k, 640)) * 0.04).astype(np.float32)).requires_grad_(True)
            p = puller.get_projected_weight(torch.from_numpy(wb), w)
            loss = puller.loss1(0.7, p, w)
            loss.backward()
            key = "%s.k%d" % (case, k)
            out[key + ".w_base"], out[key + ".w"] = wb, t2n(w)
            out[key + ".P"], out[key + ".loss1"], out[key + ".grad"] = t2n(p), t2n(loss), t2n(w.grad)
    # regloss / reglossnovel incl. the exact-zero case
    base_w, _ = net._get_base_weights()
    net.augment_base_classifier_(10, novel_weight=torch.from_numpy((rs.standard_normal((10, 640)) * 0.03).astype(np.float32)))
    out["frob.W"] = t2n(net.classifier.weight)
    out["frob.base"] = t2n(base_w)
    l0 = net.regloss(0.2, base_w)
    l0.backward()
    out["frob.regloss_zero"], out["frob.regloss_zero_grad"] = t2n(l0), t2n(net.classifier.weight.grad)
    net.classifier.weight.grad = None
    with torch.no_grad():
        net.classifier.weight += torch.from_numpy((rs.standard_normal((70, 640)) * 0.01).astype(np.float32))
    out["frob.W2"] = t2n(net.classifier.weight)
    prev = torch.from_numpy((rs.standard_normal((10, 640)) * 0.03).astype(np.float32))
    out["frob.prev"] = t2n(prev)
    l1 = net.regloss(0.2, base_w) + net.reglossnovel(0.1, prev)
    l1.backward()
    out["frob.loss"], out["frob.grad"] = t2n(l1), t2n(net.classifier.weight.grad)
    np.savez_compressed(os.path.join(GOLD, "reg.npz"), **out)
    print("reg.npz", sum(v.nbytes for v in out.values()) / 1e6, "MB")


# ------------------------------------------------------------------ G4: the loop
class _Loader(list):
    def __init__(self, items, label2human):
        super().__init__(items)
        self.dataset = SimpleNamespace(label2human=label2human)


def calibrate_bn(net, hw, signal, proto_grid=0):
    """One train-mode pass (momentum 1.0, no dropping) so the running stats match the synthetic data."""
    from oracle.resnet_ref import OnesMaskSource
    x, _ = syn.make_base_batch(99, 64, hw, class_signal=signal, proto_grid=proto_grid)
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    net.train()
    _F.masks = OnesMaskSource()
    with torch.no_grad():
        net(torch.from_numpy(x))
    for m in bns:
        m.momentum = 0.1
        m.num_batches_tracked.zero_()
    for blk in net.modules():
        if isinstance(blk, rl.BasicBlock):
            blk.num_batches_tracked = 0
    net.eval()


def centre_features(net, hw, signal, proto_grid, iters=8):
    """Make the backbone's output features zero-mean per channel on the synthetic data by calibrating layer4.1.bn3.bias
    (Newton steps through the monotone LeakyReLU + average pool).  A randomly initialised backbone's post-activation features
    are dominated by one common direction (|mean| ~ 7 x the class-specific part), which turns the fine-tuning dynamics into a
    race along that direction: every old class is forgotten completely within ~15 epochs and accuracies sit at 0 / 100 %.
    With centred features the synthetic episodes behave like the paper's: novel classes are learned, base accuracy decays
    slowly.  The calibrated bias is part of the fixture (`param.layer4.1.bn3.bias`)."""
    x, _ = syn.make_base_batch(98, 64, hw, class_signal=signal, proto_grid=proto_grid)
    xt = torch.from_numpy(x)
    bn = net.layer4[1].bn3
    net.eval()
    with torch.no_grad():
        for it in range(iters):
            f, _ = net(xt, is_feat=True)
            m = f[-1].mean(0)
            bn.bias -= m / 0.55                                   # slope of LeakyReLU(0.1) averaged over positions ~ 0.55
        f, _ = net(xt, is_feat=True)
        print("centred features: |mean| %.4f, mean |f| %.3f" % (float(f[-1].mean(0).norm()), float(f[-1].norm(dim=1).mean())))
    for blk in net.modules():
        if isinstance(blk, rl.BasicBlock):
            blk.num_batches_tracked = 0


def ncm_classifier(net, hw, signal, per_class=8, proto_grid=0, norm=0.5):
    """Base classifier rows = nearest-class-mean directions of noisy class samples, centred and made
    orthogonal to the global mean feature (there is no bias), scaled to norm 0.5."""
    ys = np.repeat(np.arange(60), per_class)
    xs = syn.make_images(4242, len(ys), hw)
    for c in range(60):
        xs[ys == c] += signal * syn.class_prototype(c, hw, proto_grid)
    net.eval()
    with torch.no_grad():
        f, _ = net(torch.from_numpy(xs), is_feat=True)
    f = t2n(f[-1]).astype(np.float64)
    g = f.mean(0)
    w = np.stack([f[ys == c].mean(0) for c in range(60)]) - g
    gh = g / np.linalg.norm(g)
    w = w - (w @ gh)[:, None] * gh[None, :]
    w = norm * w / np.linalg.norm(w, axis=1, keepdims=True)
    # the centred class means sum to zero (rank 59); a real base classifier has full rank and torch.qr has no
    # rank handling (its last basis vector would be rounding noise), so add a small seeded full-rank component
    w = w + 0.004 * np.random.RandomState(31337).standard_normal(w.shape)
    for blk in net.modules():
        if isinstance(blk, rl.BasicBlock):
            blk.num_batches_tracked = 0
    return w.astype(np.float32)


def embed_table():
    """The reference's own word vectors (word_embeds/miniImageNet_dim500.pickle, read-only data file)."""
    import pickle
    with open(os.path.join(REF, "word_embeds", "miniImageNet_dim500.pickle"), "rb") as f:
        return pickle.load(f)


def class_names(seed):
    """Synthetic label2human lists over the reference's vocabulary: 60 base names (+40 blanks, like the base loader of
    mini_imagenet.py:127-136) and 100 novel names, one to three words each."""
    words = sorted(embed_table().keys())
    rs = np.random.RandomState(seed)

    def name():
        return " ".join(rs.choice(words, size=rs.randint(1, 4), replace=False))
    return [name() for _ in range(60)] + [""] * 40, [name() for _ in range(100)]


def table_arrays(out):
    t = embed_table()
    words = sorted(t.keys())
    out["embed.words"] = np.array(words)
    out["embed.vecs"] = np.stack([np.asarray(t[w], np.float32) for w in words], 0)


def gen_semantic():
    """LangPuller.forward in its two modes + update_novel_embeds + loss1 on the result (resnet_language.py:20-90)."""
    rs = np.random.RandomState(91)
    out = {}
    table_arrays(out)
    names_base, names_novel = class_names(92)
    vocab_base = [n for n in names_base if n != ""]
    sess = [names_novel[0:5], names_novel[5:10]]
    out["vocab_base"], out["vocab_novel0"], out["vocab_novel1"] = np.array(vocab_base), np.array(sess[0]), np.array(sess[1])
    wb = (rs.standard_normal((60, 640)) * 0.05).astype(np.float32)
    out["w_base"] = wb
    for temp in (1.0, 3.0):
        opt = ref_opt(attraction_override=None, temperature=temp)
        puller = rl.LangPuller(opt, vocab_base, sess[0])
        key = "t%g" % temp
        out[key + ".E_base"], out[key + ".E_novel0"] = t2n(puller.base_embeds), t2n(puller.novel_embeds)
        wbt = torch.from_numpy(wb).requires_grad_(True)
        pl = puller(wbt)
        out[key + ".pullers0"] = t2n(pl)
        go = torch.from_numpy(rs.standard_normal((5, 640)).astype(np.float32))
        pl.backward(go)
        out[key + ".grad_out"], out[key + ".grad_w_base"] = t2n(go), t2n(wbt.grad)
        out[key + ".pullers0_masked"] = t2n(puller(torch.from_numpy(wb), mask=True))
        w = torch.from_numpy((rs.standard_normal((5, 640)) * 0.04).astype(np.float32)).requires_grad_(True)
        loss = puller.loss1(0.7, puller(torch.from_numpy(wb)), w)
        loss.backward()
        out[key + ".w"], out[key + ".loss1"], out[key + ".loss1_grad"] = t2n(w), t2n(loss), t2n(w.grad)
        puller.update_novel_embeds(sess[1])
        out[key + ".E_novel1"], out[key + ".pullers1"] = t2n(puller.novel_embeds), t2n(puller(torch.from_numpy(wb)))
    # linear-mapping variant
    opt = ref_opt(attraction_override="mapping_linear_label2image")
    puller = rl.LangPuller(opt, vocab_base, sess[0])
    mw = (rs.standard_normal((640, 500)) * 0.02).astype(np.float32)
    mb = (rs.standard_normal((640,)) * 0.02).astype(np.float32)
    puller.create_pulling_mapping({"map.weight": torch.from_numpy(mw), "map.bias": torch.from_numpy(mb)})
    out["map.weight"], out["map.bias"], out["map.pullers0"] = mw, mb, t2n(puller(torch.from_numpy(wb)))
    np.savez_compressed(os.path.join(GOLD, "semantic.npz"), **out)
    print("semantic.npz", sum(v.nbytes for v in out.values()) / 1e6, "MB")


def gen_loop(tag, hw, n_sessions, memory, n_base_batch, seed=1, real_names=False, mapping_seed=None, signal=3.0, proto_grid=0,
             hard_queries=0, centre=False, base_norm=0.5, save=True, **optkw):
    opt = ref_opt(set_seed=seed, neval_episodes=n_sessions, memory_replay=1 if memory else 0, **optkw)
    sd = syn.make_state_dict(21 + seed)
    with_bias = bool(getattr(opt, "linear_bias", False))
    if with_bias:                                # eval_incremental.py:96-103: the checkpoint holds classifier.bias => linear_bias
        sd["classifier.bias"] = syn.make_classifier_bias(21 + seed)
    net = ref_net(sd, opt)
    calibrate_bn(net, hw, signal, proto_grid)
    params0 = {}
    if centre:
        centre_features(net, hw, signal, proto_grid)
        params0["layer4.1.bn3.bias"] = t2n(net.layer4[1].bn3.bias)
    wcls = ncm_classifier(net, hw, signal, proto_grid=proto_grid, norm=base_norm)
    bn0 = {k: t2n(v) for k, v in net.state_dict().items() if "running_" in k}
    with torch.no_grad():
        net.classifier.weight.copy_(torch.from_numpy(wcls))
    sessions = syn.make_sessions(seed, n_sessions, hw, class_signal=signal, proto_grid=proto_grid, hard_queries=hard_queries)
    base_x, base_y = syn.make_base_batch(seed, n_base_batch, hw, class_signal=signal, proto_grid=proto_grid)
    inits = syn.make_novel_inits(seed, n_sessions)
    bias_inits = syn.make_novel_bias_inits(seed, n_sessions)
    names_base = ["b%d" % i for i in range(60)] + [""] * 40
    names_novel = ["n%d" % i for i in range(100)]
    ckpt = {}
    if real_names:                               # semantic / mapping variants read the class names' word vectors
        names_base, names_novel = class_names(200 + seed)
    if mapping_seed is not None:
        mw, mb = syn.make_linear_map(mapping_seed)
        ckpt = {"mapping_linear_label2image": {"map.weight": torch.from_numpy(mw), "map.bias": torch.from_numpy(mb)}}
    base_loader = _Loader([(torch.from_numpy(base_x), torch.from_numpy(base_y), torch.arange(len(base_y)))], names_base)
    meta = _Loader([(torch.from_numpy(s["support_xs"])[None], torch.from_numpy(s["support_ys"])[None],
                     torch.from_numpy(s["query_xs"])[None], torch.from_numpy(s["query_ys"])[None]) for s in sessions],
                   names_novel)
    bsl = None
    if memory:
        bx, by = syn.make_base_support(seed, hw, class_signal=signal, proto_grid=proto_grid)
        bsl = _Loader([(torch.from_numpy(bx)[None], torch.from_numpy(by)[None],
                        torch.zeros(1, 1, 3, hw, hw), torch.zeros(1, 1, dtype=torch.long))], names_base)
    # explicit init rows through the reference's own novel_weight= argument; one call == one session start
    rec = dict(loss=[], val=[], picks=[], base=[])
    counter = {"i": 0}
    orig_aug = net.augment_base_classifier_

    def aug(n, novel_weight=None, novel_bias=None):
        r = orig_aug(n, novel_weight=torch.from_numpy(inits[counter["i"]]),
                     novel_bias=torch.from_numpy(bias_inits[counter["i"]]) if with_bias else None)
        counter["i"] += 1
        rec["loss"].append([])
        rec["val"].append([])
        return r
    net.augment_base_classifier_ = aug
    # record per-epoch losses (at loss.backward()), per-epoch validation accuracies, memory picks
    orig_validate, orig_choice, orig_backward, orig_eval_base = le.validate, np.random.choice, torch.Tensor.backward, le.eval_base

    def eval_base(*a, **k):
        r = orig_eval_base(*a, **k)
        rec["base"].append(float(r[0] if isinstance(r, tuple) else r))
        return r

    def validate(query_xs, query_ys_id, net_, criterion, opt_, epoch):
        r = orig_validate(query_xs, query_ys_id, net_, criterion, opt_, epoch)
        if isinstance(query_xs, list):
            rec["val"][-1].append([float(a) for a in r[0]])
        return r

    def choice(a, size=None, *args, **kw):
        r = orig_choice(a, size, *args, **kw)
        rec["picks"].append(np.array(r))
        return r

    def backward(self, *a, **k):
        rec["loss"][-1].append(float(self.item()))
        return orig_backward(self, *a, **k)
    le.validate, np.random.choice, torch.Tensor.backward, le.eval_base = validate, choice, backward, eval_base
    crit = torch.nn.CrossEntropyLoss()
    set_masks(61 + seed)
    t0 = time.time()
    try:
        novel_avg, base_avg = le.few_shot_finetune_incremental_test(net, ckpt, crit, meta, base_loader, opt,
                                                                    vis=False, base_support_loader=bsl)
    finally:
        le.validate, np.random.choice, torch.Tensor.backward, le.eval_base = orig_validate, orig_choice, orig_backward, orig_eval_base
    took = time.time() - t0
    print(tag, "reference loop took %.1f s" % took)
    if not save:                                 # (timing runs: `timeref`)
        return took, [len(l) for l in rec["loss"]]
    out = dict(hw=np.array(hw), n_sessions=np.array(n_sessions), memory=np.array(int(memory)), seed=np.array(seed),
               n_base_batch=np.array(n_base_batch), signal=np.array(signal), sd_seed=np.array(21 + seed),
               mask_seed=np.array(61 + seed), base_classifier=wcls,
               final_classifier=t2n(net.classifier.weight), novel_avg=np.array(novel_avg), base_avg=np.array(base_avg),
               picks=np.array([p.reshape(-1) for p in rec["picks"]]) if rec["picks"] else np.zeros((0, 1), np.int64))
    for k, v in optkw.items():
        if v is not None:
            out["opt." + k] = np.array(v)
    out["attraction_override"] = np.array(str(opt.attraction_override))
    if with_bias:                                # base bias = subreg_hip.synthetic.make_classifier_bias(sd_seed)
        out["final_bias"] = t2n(net.classifier.bias)
        out["opt.lmbd_reg_novel_is_none"] = np.array(int(opt.lmbd_reg_novel is None))
    if real_names:
        table_arrays(out)
        out["names_base"], out["names_novel"] = np.array(names_base), np.array(names_novel)
    if mapping_seed is not None:
        out["mapping_seed"] = np.array(mapping_seed)          # weights = subreg_hip.synthetic.make_linear_map(seed)
    for k, v in bn0.items():
        out["bn0." + k] = v
    for k, v in params0.items():
        out["param." + k] = v                                 # backbone parameters that differ from make_state_dict(sd_seed)
    if proto_grid or hard_queries:
        out["proto_grid"], out["hard_queries"] = np.array(proto_grid), np.array(hard_queries)
    out["acc_base_sessions"] = np.array(rec["base"], np.float64)   # eval_base: before session 1, then after every session
    for s in range(n_sessions):
        out["s%d.loss" % s] = np.array(rec["loss"][s], np.float64)
        out["s%d.last_val" % s] = np.array(rec["val"][s][-1], np.float64)
        out["s%d.epochs" % s] = np.array(len(rec["loss"][s]))
    for k in ("layer1.0.bn1", "layer4.1.bn3"):
        m = dict(net.named_modules())[k]
        out[k + ".running_mean"], out[k + ".running_var"] = t2n(m.running_mean), t2n(m.running_var)
    if int(opt.freeze_backbone_at) != 1:         # the backbone was fine-tuned before the freeze: pin what it became
        sdf = net.state_dict()
        for k in ("layer1.0.conv1.weight", "layer2.0.bn2.weight", "layer2.0.bn2.bias", "layer3.1.conv2.weight", "layer4.0.downsample.0.weight",
                  "layer4.1.conv3.weight", "layer4.1.bn3.bias"):
            full = t2n(sdf[k])
            out["final." + k] = full if full.size <= 50000 else full[:2]          # (large tensors: the first two output channels ...)
            out["final_delta_norm." + k] = np.array(np.linalg.norm((full - np.asarray(sd[k])).astype(np.float64)))   # ... and how far all of it moved
        out["final.requires_grad"] = np.array([int(p.requires_grad) for n, p in net.named_parameters() if not n.startswith("classifier")])
    np.savez_compressed(os.path.join(GOLD, "loop_%s.npz" % tag), **out)
    print("loop_%s.npz" % tag, {k: v for k, v in out.items() if k.endswith("epochs")})


# ------------------------------------------------------------------ G6: episode sampler (dataset/mini_imagenet.py)
def gen_episodes():
    """Which images feed which session: runs the reference's ImageNet / MetaImageNet classes themselves on a synthetic
    all.pickle (100 classes x 600 two-by-two 'images' whose pixels spell their own index) and records the indices.
    dataset/mini_imagenet.py imports torchvision.transforms (absent in this image) at module level; the module object below
    only lets that import succeed - its names are never CALLED, because explicit transforms (index decoders) are passed to
    every dataset.  The fixture therefore pins the numpy index sampling only, not the image augmentation."""
    import pickle
    import tempfile
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    for name in ("Normalize", "Compose", "RandomCrop", "ColorJitter", "RandomHorizontalFlip", "ToTensor"):
        setattr(tvt, name, lambda *a, **k: None)
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    import dataset.mini_imagenet as mi                                   # the reference
    rs = np.random.RandomState(123)
    labels = rs.permutation(np.repeat(np.arange(100), 600))
    n = labels.shape[0]
    idx = np.arange(n)
    imgs = np.zeros((n, 2, 2, 3), np.uint8)
    imgs[:, 0, 0, 0], imgs[:, 0, 0, 1], imgs[:, 0, 0, 2] = idx & 255, (idx >> 8) & 255, (idx >> 16) & 255
    decode = lambda x: torch.tensor(float(int(x[0, 0, 0]) | (int(x[0, 0, 1]) << 8) | (int(x[0, 0, 2]) << 16)))   # noqa: E731
    out = {"labels": labels.astype(np.int64)}
    with tempfile.TemporaryDirectory() as root:
        with open(os.path.join(root, "all.pickle"), "wb") as f:
            pickle.dump({"data": imgs, "labels": labels.tolist(), "catname2label": {"n%04d" % c: c for c in range(100)}}, f)
        with open(os.path.join(root, "class_labels.txt"), "w") as f:
            f.write("".join("n%04d class_%d\n" % (c, c) for c in range(100)))
        for seed in (1, 7):
            args = SimpleNamespace(data_aug=True, set_seed=seed, continual=True, data_root=root, n_ways=5, n_shots=5, n_queries=25,
                                   n_test_runs=8, eval_mode="few-shot-incremental-fine-tune", n_aug_support_samples=5,
                                   n_base_aug_support_samples=0, n_base_support_samples=1)
            key = "seed%d" % seed
            base_test = mi.ImageNet(args=args, split="train", phase="test", transform=decode)
            out[key + ".basec"] = np.array(sorted(base_test.basec_map.keys()))
            out[key + ".base_test_len"] = np.array(len(base_test))
            items = [base_test[i] for i in range(0, len(base_test), 37)]
            out[key + ".base_test_items"] = np.array([[int(x), int(t), int(i)] for x, t, i in items])
            out[key + ".label2human_nonempty"] = np.array(sum(1 for h in base_test.label2human if h != ""))
            base_sup = mi.MetaImageNet(args=args, split="train", phase="train", train_transform=decode, test_transform=decode,
                                       fix_seed=True, use_episodes=False)
            for item in (0, 3):
                sx, sy, _qx, _qy = base_sup[item]
                out["%s.base_support%d.pos" % (key, item)] = sx.numpy().astype(np.int64)
                out["%s.base_support%d.ys" % (key, item)] = np.asarray(sy).astype(np.int64)
            meta = mi.MetaImageNet(args=args, split="val", train_transform=decode, test_transform=decode, fix_seed=True,
                                   use_episodes=False, disjoint_classes=True)
            for item in range(8):
                sx, sy, qx, qy = meta[item]
                for nm, v in (("sup", sx.numpy()), ("sup_ys", sy), ("qry", qx.numpy()), ("qry_ys", qy)):
                    out["%s.s%d.%s" % (key, item, nm)] = np.asarray(v).astype(np.int64)
    np.savez_compressed(os.path.join(GOLD, "episodes.npz"), **out)
    print("episodes.npz", sum(v.nbytes for v in out.values()) / 1e6, "MB")


# ------------------------------------------------------------------ G5: one pretraining step (train_supervised.py:205-268)
TRAIN_SLICES = {"layer1.0.conv1.weight": None, "layer1.0.downsample.0.weight": None, "layer1.0.conv2.weight": 8,
                "layer2.0.conv2.weight": 8, "layer2.0.downsample.0.weight": 16, "layer3.0.conv1.weight": 4,
                "layer3.1.conv3.weight": 4, "layer4.0.downsample.0.weight": 8, "layer4.1.conv1.weight": 2,
                "layer4.1.conv3.weight": 2, "classifier.weight": None}


def gen_train_step(cases=((84, 6), (32, 8)), fname="train_step.npz"):
    """cases: (image size, batch).  train_step.npz holds the two small cases; train_step_b64.npz (`train64`) the batch
    train_supervised.py really runs (configs.py:124: batch_size 64, 84x84) - the one bench.py's pretraining leg times, where the
    HIP path selects its split-K workspace, the dW split targets and the two-stream schedule."""
    opt = ref_opt()
    out = {}
    for hw, B in cases:
        sd = syn.make_state_dict(71)
        net = ref_net(sd, opt)
        x = syn.make_images(72, B, hw)
        y = np.random.RandomState(73).randint(0, 60, B)
        net.train()
        set_masks(74)
        logits = net(torch.from_numpy(x))
        loss = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(y))
        loss.backward()
        key = "hw%d" % hw
        out[key + ".B"], out[key + ".labels"] = np.array(B), y
        out[key + ".loss"], out[key + ".logits"] = t2n(loss), t2n(logits)
        for name, prm in net.named_parameters():
            g = t2n(prm.grad)
            out["%s.gnorm.%s" % (key, name)] = np.array(np.linalg.norm(g.astype(np.float64)))
            if name in TRAIN_SLICES:
                k = TRAIN_SLICES[name]
                out["%s.grad.%s" % (key, name)] = g if k is None else g[:k]
            elif ".bn" in name or "downsample.1" in name:
                out["%s.grad.%s" % (key, name)] = g
        for k in ("layer1.0.bn1", "layer4.1.bn3"):
            m = dict(net.named_modules())[k]
            out["%s.%s.running_mean" % (key, k)], out["%s.%s.running_var" % (key, k)] = t2n(m.running_mean), t2n(m.running_var)
        # one SGD step with train_supervised.py's hyper-parameters (configs.py:124-134: lr 0.05, momentum 0.9, wd 5e-4)
        o = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
        o.step()
        out[key + ".after_step.classifier.weight"] = t2n(net.classifier.weight)
        out[key + ".after_step.layer1.0.conv1.weight"] = t2n(dict(net.named_parameters())["layer1.0.conv1.weight"])
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    print(fname, sum(v.nbytes for v in out.values()) / 1e6, "MB")


def main():
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["blocks", "backbone", "reg", "train", "train64", "loop32", "adam", "bias", "semantic", "episodes", "loop84", "freeze_opts"]
    if "blocks" in what:
        gen_blocks()
    if "backbone" in what:
        gen_backbone()
    if "reg" in what:
        gen_reg()
    if "train" in what:
        gen_train_step()
    if "train64" in what:
        gen_train_step(cases=((84, 64),), fname="train_step_b64.npz")
    if "loop32" in what:
        gen_loop("hw32_noM", 32, 2, False, 40, seed=1, max_novel_epochs=4)
        gen_loop("hw32_M", 32, 3, True, 40, seed=2, max_novel_epochs=3)
        # data-dependent stop: loose epsilon + short stable window so the stable rule fires before the cap
        gen_loop("hw32_stop", 32, 2, True, 40, seed=3, max_novel_epochs=40, stable_epochs=3,
                 convergence_epsilon=2e-2)
    if "adam" in what:
        # --adam (eval/util.py:92-97: torch.optim.Adam(lr, weight_decay=0.0005) instead of SGD), +M, three sessions
        gen_loop("hw32_adam", 32, 3, True, 40, seed=13, max_novel_epochs=5, adam=True)
    if "freeze" in what:
        # freeze_backbone_at = 3 (language_eval.py:243-249, eval/util.py:62-69): epochs 1-2 of the first session fine-tune the WHOLE
        # network (epoch 1 in train mode, epoch 2 in eval mode: validate() leaves the model there), the backbone freezes at epoch 3
        gen_loop("hw32_freeze3", 32, 2, False, 40, seed=15, max_novel_epochs=6, freeze_backbone_at=3)
    if "timeref" in what:
        # the REFERENCE's own loop on this container's CPU cores, config 1 (one -M session of a miniImageNet-shaped
        # 5-way 5-shot run, 84x84, 125 support + 125 query images, 1000-image base evaluation), E fixed: 5 (smoke) and 100 (headline)
        import json
        res = {"threads": torch.get_num_threads(), "cores": os.cpu_count()}
        for E in (5, 100):
            took, epochs = gen_loop("timeref_E%d" % E, 84, 1, False, 1000, seed=1, max_novel_epochs=E, stable_epochs=E + 1, save=False)
            imgs = E * 250 + 2 * 1000                   # per epoch: support + one query set; base evaluation before and after the session
            res["E%d" % E] = {"wall_s": round(took, 1), "epochs": epochs, "episodes_per_s": 1.0 / took, "images_per_s": imgs / took}
            print(json.dumps(res), flush=True)
    if "freeze_opts" in what:
        # the same with --adam: get_optim's ONE torch.optim.Adam(lr, weight_decay=0.0005) over net.parameters() (eval/util.py:92-97)
        # also steps the backbone in epochs 1-2.  lr 2e-5: Adam moves EVERY element by ~lr per step whatever its gradient; at the
        # loop's default 0.002 the backbone of this random-weight fixture is destroyed in one step (loss 6.5 -> 35 -> 340) and the run
        # pins nothing but chaos
        gen_loop("hw32_freeze3_adam", 32, 2, False, 40, seed=17, max_novel_epochs=6, freeze_backbone_at=3, adam=True, learning_rate=2e-5)
        # a backbone that NEVER freezes (every session ends at epoch 3 < freeze_backbone_at = 5) with replay memory: from session 2
        # on every step makes TWO gradient-carrying forwards (support, then memory: language_eval.py:252-258), train mode in epoch 1
        # lr 2e-4: at the loop's default 0.002 whole-network SGD on this random-weight fixture amplifies a 1e-7 difference tenfold per
        # step (loss 5.7 -> 4.1 in two steps), and by session 3 two correct fp32 implementations differ by 1e-3
        gen_loop("hw32_freeze5_M", 32, 3, True, 40, seed=18, max_novel_epochs=3, freeze_backbone_at=5, learning_rate=2e-4)
    if "bias" in what:
        # classifier WITH bias (a backbone pretrained without --no_linear_bias; eval_incremental.py:96-103 reads it off the
        # checkpoint), +M, three sessions.  --lmbd_reg_novel must be absent: with a bias the reference's reglossnovel
        # (resnet_language.py:238) raises IndexError from session 2 on.
        gen_loop("hw32_bias", 32, 3, True, 40, seed=14, max_novel_epochs=5, linear_bias=True, lmbd_reg_novel=None)
    if "semantic" in what:
        gen_semantic()
        gen_loop("hw32_sem", 32, 3, True, 40, seed=5, max_novel_epochs=4, real_names=True, attraction_override=None,
                 temperature=3.0)
        gen_loop("hw32_map", 32, 2, False, 40, seed=6, max_novel_epochs=4, real_names=True, mapping_seed=77,
                 attraction_override="mapping_linear_label2image")
    if "episodes" in what:
        gen_episodes()
    if "loop84" in what:
        gen_loop("hw84_M", 84, 3, True, 40, seed=4, max_novel_epochs=4)
    if "loop84d" in what:
        # DISCRIMINATING bench-scale goldens (84x84, -M): low-frequency class prototypes + centred features (centre_features) put
        # the session accuracies at 40-95 % and keep the base accuracy well above zero; 5 of every class's 25 query images are
        # wrong by construction (hard_queries), so no accuracy saturates and no image sits near a decision boundary.
        # The learning rate is 75 x the scripts' 0.002: centred synthetic features have |f| ~ 2 (real ones ~ 20-30), and the
        # CE gradient scales with |f|^2.  (~10 + ~5 min of torch-CPU)
        gen_loop("hw84_noM_disc", 84, 2, False, 200, seed=8, signal=0.3, proto_grid=3, hard_queries=5, centre=True, base_norm=2.0,
                 max_novel_epochs=30, learning_rate=0.15)
        # ... and one that ENDS ON THE STABLE-EPOCHS RULE (language_eval.py:298-318) at 84x84
        gen_loop("hw84_stop", 84, 1, False, 100, seed=9, signal=0.3, proto_grid=3, hard_queries=5, centre=True, base_norm=2.0,
                 max_novel_epochs=80, learning_rate=0.15, stable_epochs=3, convergence_epsilon=4e-2)
    if "loop84sem" in what:
        # the semantic subspace regularizer and the linear-mapping target (scripts/continual/slurm_semantic_subspace_reg.sh,
        # slurm_linear_mapping.sh) at 84x84 on the discriminating episodes, with the reference's own word vectors (~10 + ~3 min).
        # label_pull 0.03 (the scripts sweep it): at 1.0 the constant semantic target holds the novel rows and nothing is learned
        # in 30 epochs on these episodes (novel accuracy 0-2 %)
        gen_loop("hw84_sem", 84, 2, False, 100, seed=10, signal=0.3, proto_grid=3, hard_queries=5, centre=True, base_norm=2.0,
                 max_novel_epochs=30, learning_rate=0.15, label_pull=0.03, real_names=True, attraction_override=None, temperature=3.0)
        gen_loop("hw84_map", 84, 1, False, 100, seed=11, signal=0.3, proto_grid=3, hard_queries=5, centre=True, base_norm=2.0,
                 max_novel_epochs=30, learning_rate=0.15, label_pull=0.03, real_names=True, mapping_seed=78,
                 attraction_override="mapping_linear_label2image")
    if "loop84s8" in what:
        # bench-scale case (BASELINE.json configs[1]): 8 sessions, -M, 84x84, 1000-image base batch; 6 epochs per
        # session so that the build's loop captures and replays its per-epoch hipGraph (~25 min of torch-CPU here)
        gen_loop("hw84_noM_s8", 84, 8, False, 1000, seed=7, max_novel_epochs=6)


if __name__ == "__main__":
    main()
