"""Pin the CPU oracle against the golden vectors produced by the reference itself.

The fixtures under tests/golden/ were written by tools/make_golden.py, which
imports /root/reference (PyTorch CPU) in the build container.  These tests run
without a GPU and without the reference.
Tolerance: the path is fp32; north_star asks for 1e-4 on weights/accuracies.  The
oracle differs from torch only in summation order, so block outputs (|x| ~ 1..10)
are held to 2e-4 absolute+relative and the regularizer/loop scalars tighter.
"""
import os

import numpy as np
import pytest

from oracle import loop_ref, subspace_ref as sr
from oracle.resnet_ref import MaskSource, ResNetRef, _nchw, _nhwc, block_specs, copy_state_dict
from subreg_hip import synthetic as syn

from conftest import GOLDEN

ATOL = RTOL = 2e-4


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _close(a, b, atol=ATOL, rtol=RTOL, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert err.max() <= 0, "%s: max abs diff %.3e (ref max %.3e)" % (what, np.abs(a - b).max(), np.abs(b).max())


# ---------------------------------------------------------------- G1 blocks
@pytest.mark.parametrize("spec", block_specs(), ids=lambda s: s["name"])
def test_block_eval_and_train(spec):
    g = _load("blocks.npz")
    name = spec["name"]
    li = [s["name"] for s in block_specs()].index(name)
    for bs_name, bs in (("bs1", 1), ("bs5", 5)):
        key = "%s.%s" % (name, bs_name)
        if key + ".in_shape" not in g:
            continue
        x = np.random.RandomState(int(g[key + ".in_seed"])).standard_normal(tuple(g[key + ".in_shape"])).astype(np.float32)
        sd = syn.make_state_dict(11)
        net = ResNetRef(sd, block_size=bs)
        if bs_name == "bs1":
            net.eval()
            y = _nchw(net.block_forward(_nhwc(x), spec))
            _close(y, g[key + ".eval_out"], what=key + " eval")
        net.train()
        net.nbt[name] = 39999
        y = _nchw(net.block_forward(_nhwc(x), spec, MaskSource(int(g[key + ".mask_seed"]))))
        _close(y, g[key + ".train_out"], what=key + " train")
        for bn in ("bn1", "bn2", "bn3") + (("downsample.1",) if spec["downsample"] else ()):
            _close(sd["%s.%s.running_mean" % (name, bn)], g["%s.%s.running_mean" % (key, bn)], 1e-5, 1e-5, key + bn)
            _close(sd["%s.%s.running_var" % (name, bn)], g["%s.%s.running_var" % (key, bn)], 1e-5, 1e-5, key + bn)


def test_dropblock_mask_matches_reference_semantics():
    # one seed blanks a bs x bs block anchored at the seed index (resnet_language.py:327-357) ...
    from oracle.resnet_ref import dropblock_block_mask
    s = np.zeros((1, 1, 6, 6), np.float32)
    s[0, 0, 2, 3] = 1
    bm = dropblock_block_mask(s, 5)
    assert bm.shape == (1, 1, 10, 10)
    assert bm[0, 0, 2:7, 3:8].sum() == 0 and bm.sum() == 100 - 25
    # ... but with gcd(n, bs^2) = 5 the reference's repeat/tile pairing gives each seed only 5 of 25 offsets
    s = np.zeros((5, 1, 1, 1), np.float32)
    s[:, 0, 0, 0] = 1
    bm = dropblock_block_mask(s, 5)
    assert bm.shape == (5, 1, 5, 5)
    for r in range(5):
        dropped = {(int(i), int(j)) for i, j in np.argwhere(bm[r, 0] == 0)}
        # column r from the paired offsets + the seed itself, which F.pad leaves at the padded centre (2,2)
        assert dropped == {(i, r) for i in range(5)} | {(2, 2)}


# ---------------------------------------------------------------- G2 backbone
@pytest.mark.parametrize("hw", [32, 84])
def test_backbone(hw):
    g = _load("backbone.npz")
    sd = syn.make_state_dict(21)
    net = ResNetRef(sd)
    x = syn.make_images(31, 4, hw)
    net.eval()
    feat, stages = net.features(x, return_stages=True)
    _close(_nchw(stages[0])[0], g["hw%d.eval_f0_b0" % hw], what="stage1")
    _close(feat, g["hw%d.eval_feat" % hw], what="feat")
    from oracle.resnet_ref import linear
    _close(linear(feat, sd["classifier.weight"]), g["hw%d.eval_logits" % hw], what="logits")
    net.train()
    _close(net.forward(x, MaskSource(41)), g["hw%d.train_logits" % hw], 5e-4, 5e-4, what="train logits")
    for k in ("layer1.0.bn1", "layer2.0.downsample.1", "layer3.1.bn2", "layer4.1.bn3"):
        _close(sd[k + ".running_mean"], g["hw%d.%s.running_mean" % (hw, k)], 1e-5, 1e-4, k)
        _close(sd[k + ".running_var"], g["hw%d.%s.running_var" % (hw, k)], 1e-5, 1e-4, k)
    net.eval()
    _close(net.forward(x), g["hw%d.eval2_logits" % hw], 5e-4, 5e-4, what="eval after stats moved")


# ---------------------------------------------------------------- G3 regularizers
@pytest.mark.parametrize("case", ["rand.k5", "rand.k40", "trained.k5", "trained.k40"])
def test_subspace_projection_loss_grad(case):
    g = _load("reg.npz")
    wb, w = g[case + ".w_base"], g[case + ".w"]
    _close(sr.get_projected_weight(wb, w), g[case + ".P"], 1e-6, 1e-5, "P")
    loss, grad = sr.loss1_and_grad(0.7, wb, w)
    _close(loss, g[case + ".loss1"], 1e-6, 1e-5, "loss1")
    _close(grad, g[case + ".grad"], 1e-6, 1e-5, "grad")
    # closed form of SURVEY.md section 0 item 4: grad = 2*gamma*(w - w Q Q^T)
    q = sr.orthonormal_basis(wb)
    _close(2 * 0.7 * (w - w @ q @ q.T), g[case + ".grad"], 1e-6, 1e-5, "closed form")


def test_frobenius_regs_and_zero_subgradient():
    g = _load("reg.npz")
    l0, g0 = sr.frob_reg_and_grad(0.2, g["frob.W"][:60], g["frob.base"])
    assert l0 == 0.0 and float(g["frob.regloss_zero"]) == 0.0
    assert not g0.any() and not g["frob.regloss_zero_grad"].any()
    l1, g1 = sr.frob_reg_and_grad(0.2, g["frob.W2"][:60], g["frob.base"])
    l2, g2 = sr.frob_reg_and_grad(0.1, g["frob.W2"][60:70], g["frob.prev"])
    _close(l1 + l2, g["frob.loss"], 1e-6, 1e-5, "loss")
    full = np.zeros((70, 640))
    full[:60] += g1
    full[60:70] += g2
    _close(full, g["frob.grad"], 1e-7, 1e-5, "grad")


# ---------------------------------------------------------------- G4 loop
def _loop_setup(g):
    hw, ns, seed = int(g["hw"]), int(g["n_sessions"]), int(g["seed"])
    signal, memory = float(g["signal"]), bool(int(g["memory"]))
    sd = syn.make_state_dict(int(g["sd_seed"]))
    for k in g.files:
        if k.startswith("bn0."):
            sd[k[4:]] = g[k].copy()
    sd["classifier.weight"] = g["base_classifier"].copy()
    from types import SimpleNamespace
    opt = SimpleNamespace(n_ways=5, n_shots=5, learning_rate=0.002, momentum=0.9, weight_decay=5e-4,
                          lmbd_reg_transform_w=0.2, lmbd_reg_novel=0.1, label_pull=1.0, max_novel_epochs=1000,
                          min_novel_epochs=20, target_train_loss=0.0, convergence_epsilon=1e-4, stable_epochs=10,
                          memory_replay=1 if memory else 0, adam=False)
    for k in g.files:
        if k.startswith("opt."):
            setattr(opt, k[4:], g[k].item())
    if "final_bias" in g.files:                     # classifier with bias (tools/make_golden.py `bias`)
        sd["classifier.bias"] = syn.make_classifier_bias(int(g["sd_seed"]))
        if int(g["opt.lmbd_reg_novel_is_none"]):
            opt.lmbd_reg_novel = None
    sessions = syn.make_sessions(seed, ns, hw, class_signal=signal)
    base = syn.make_base_batch(seed, int(g["n_base_batch"]), hw, class_signal=signal)
    bsup = syn.make_base_support(seed, hw, class_signal=signal) if memory else None
    inits = syn.make_novel_inits(seed, ns)
    if "attraction_override" in g.files:
        ao = str(g["attraction_override"])
        opt.attraction_override = None if ao == "None" else ao
    else:
        opt.attraction_override = "distance2subspace"
    opt.word_embed_size = 500
    opt.temperature = float(getattr(opt, "temperature", 1))
    return sd, opt, sessions, base, bsup, inits, int(g["mask_seed"]), [p for p in g["picks"]]


def _embed_kwargs(g):
    """embeds / names / mapping arguments of the semantic and linear-mapping loop goldens."""
    if "embed.words" not in g.files:
        return {}
    kw = dict(embeds={str(w): v for w, v in zip(g["embed.words"], g["embed.vecs"])},
              names=([str(n) for n in g["names_base"]], [str(n) for n in g["names_novel"]]))
    if "mapping_seed" in g.files:
        kw["mapping"] = syn.make_linear_map(int(g["mapping_seed"]))
    return kw


@pytest.mark.parametrize("tag", ["hw32_noM", "hw32_M", "hw32_stop", "hw32_sem", "hw32_map", "hw32_adam", "hw32_bias"])
def test_loop_against_reference(tag):
    g = _load("loop_%s.npz" % tag)
    sd, opt, sessions, base, bsup, inits, mseed, picks = _loop_setup(g)
    net = ResNetRef(sd)
    out = loop_ref.run_incremental(net, sessions, base, opt, inits, base_support=bsup,
                                   masks=MaskSource(mseed), memory_picks=picks,
                                   novel_bias_inits=syn.make_novel_bias_inits(int(g["seed"]), len(sessions)), **_embed_kwargs(g))
    if "final_bias" in g.files:
        _close(out["classifier_bias"], g["final_bias"], 1e-5, 1e-4, "final bias")
    for s in range(len(sessions)):
        assert out["epochs"][s] == int(g["s%d.epochs" % s]), (s, out["epochs"], g["s%d.epochs" % s])
        _close(out["loss"][s], g["s%d.loss" % s], 2e-4, 2e-4, "loss s%d" % s)
        _close(out["test_acc"][s], np.round(g["s%d.last_val" % s], 2), 1e-6, 0, "val acc s%d" % s)
    _close(out["classifier_weight"], g["final_classifier"], 1e-4, 1e-4, "final classifier")
    _close(np.mean(out["novel_vals"]), g["novel_avg"], 1e-5, 1e-6, "novel avg")
    _close(np.mean(out["base_vals"]), g["base_avg"], 1e-5, 1e-6, "base avg")
    for k in ("layer1.0.bn1", "layer4.1.bn3"):
        _close(sd[k + ".running_mean"], g[k + ".running_mean"], 1e-5, 1e-4, k)
        _close(sd[k + ".running_var"], g[k + ".running_var"], 1e-5, 1e-4, k)


def test_loop_with_bias_and_novel_reg_fails_like_the_reference():
    """resnet_language.py:238 indexes the 1-D bias with two indices: with a bias and --lmbd_reg_novel the reference raises
    IndexError in the first epoch of session 2 (checked against the reference itself when the golden was generated:
    `net.reglossnovel(0.1, w, b)` -> IndexError: too many indices for tensor of dimension 1)."""
    g = _load("loop_hw32_bias.npz")
    sd, opt, sessions, base, bsup, inits, mseed, picks = _loop_setup(g)
    opt.lmbd_reg_novel, opt.max_novel_epochs = 0.1, 1
    with pytest.raises(IndexError, match="too many indices"):
        loop_ref.run_incremental(ResNetRef(sd), sessions, base, opt, inits, base_support=bsup, masks=MaskSource(mseed),
                                 memory_picks=picks, novel_bias_inits=syn.make_novel_bias_inits(int(g["seed"]), len(sessions)))


def test_memory_index_formula():
    # language_eval.py:354-358: one shot index per class, its 5 augmented copies
    inds = loop_ref.memory_indices([2])
    sy, _ = syn.session_labels(0)
    assert len(inds) == 25 and len(set(inds)) == 25
    assert sorted(np.bincount(sy[inds] - 60)) == [5] * 5


# ---------------------------------------------------------------- G5 pretraining step (forward + backward)
@pytest.mark.parametrize("hw", [32] + ([84] if os.environ.get("SUBREG_SLOW_TESTS") else []))   # hw=84: 3 min in fp64 NumPy
def test_train_step_backward(hw):
    from oracle import backward_ref as br
    g = _load("train_step.npz")
    key = "hw%d" % hw
    sd = syn.make_state_dict(71)
    x = syn.make_images(72, int(g[key + ".B"]), hw)
    loss, logits, grads = br.train_step(sd, x, g[key + ".labels"], MaskSource(74))
    _close(loss, g[key + ".loss"], 1e-5, 1e-5, "loss")
    _close(logits, g[key + ".logits"], 5e-4, 5e-4, "logits")
    n_full = 0
    for k in g.files:
        if k.startswith(key + ".gnorm."):
            name = k[len(key) + 7:]
            _close(np.linalg.norm(grads[name]), g[k], 1e-5, 2e-3, "gnorm " + name)
        elif k.startswith(key + ".grad."):
            name = k[len(key) + 6:]
            want = g[k]
            got = grads[name][:want.shape[0]]
            # the reference accumulates weight gradients over B*H*W pixels in fp32: 1e-3 of the tensor's magnitude
            _close(got, want, 1e-3 * max(1.0, float(np.abs(want).max())), 2e-3, "grad " + name)
            n_full += 1
    assert n_full > 50
    for k in ("layer1.0.bn1", "layer4.1.bn3"):
        _close(sd[k + ".running_mean"], g["%s.%s.running_mean" % (key, k)], 1e-5, 1e-4, k)
    w = sd["classifier.weight"]
    p, _ = br.sgd_momentum_step(w.astype(np.float64), grads["classifier.weight"], None, 0.05, 0.9, 5e-4)
    _close(p, g[key + ".after_step.classifier.weight"], 1e-6, 1e-5, "sgd step")


def test_semantic_regularizer_against_reference():
    """LangPuller.forward (:75-87) in both modes, get_embeds (models/util.py:50-66), update_novel_embeds, loss1 on the result."""
    g = _load("semantic.npz")
    table = {str(w): v for w, v in zip(g["embed.words"], g["embed.vecs"])}
    vb = [str(n) for n in g["vocab_base"]]
    v0, v1 = [str(n) for n in g["vocab_novel0"]], [str(n) for n in g["vocab_novel1"]]
    assert any(" " in n for n in vb + v0 + v1)                       # multi-word names exercise the mean
    for temp in (1.0, 3.0):
        key = "t%g" % temp
        eb, e0, e1 = sr.get_embeds(table, vb), sr.get_embeds(table, v0), sr.get_embeds(table, v1)
        _close(eb, g[key + ".E_base"], 1e-6, 1e-6, "E_base")
        _close(e0, g[key + ".E_novel0"], 1e-6, 1e-6, "E_novel0")
        _close(e1, g[key + ".E_novel1"], 1e-6, 1e-6, "E_novel1")
        t0, p0 = sr.semantic_target(e0, eb, g["w_base"], temp)
        _close(t0, g[key + ".pullers0"], 1e-5, 1e-5, "pullers0")
        _close(p0.T @ g[key + ".grad_out"].astype(np.float64), g[key + ".grad_w_base"], 1e-5, 1e-5, "d W_base")
        _close(sr.semantic_target(e0, eb, g["w_base"], temp, mask=True)[0], g[key + ".pullers0_masked"], 1e-5, 1e-5, "masked")
        _close(sr.semantic_target(e1, eb, g["w_base"], temp)[0], g[key + ".pullers1"], 1e-5, 1e-5, "pullers1")
        loss, grad = sr.loss1_to_target_and_grad(0.7, t0, g[key + ".w"])
        _close(loss, g[key + ".loss1"], 1e-5, 1e-5, "loss1")
        _close(grad, g[key + ".loss1_grad"], 1e-5, 1e-5, "loss1 grad")
    _close(sr.linear_map_target(sr.get_embeds(table, v0), g["map.weight"], g["map.bias"]), g["map.pullers0"], 1e-5, 1e-5, "mapping")
    # the unknown-word quirk of get_embeds: the running sum is reset, later words still count, divisor = all words
    known = sorted(table)[0]
    e = sr.get_embeds(table, ["%s notaword %s" % (known, known), "notaword"])
    _close(e[0], table[known] / 3.0, 1e-6, 1e-6, "unknown word resets the sum")
    assert not e[1].any()


# ---------------------------------------------------------------- torch-CPU restatement (bench.py's cpu_baseline leg)
@pytest.mark.parametrize("hw", [32, 84])
def test_torch_cpu_ref_backbone_matches_reference_golden(hw):
    torch = pytest.importorskip("torch")
    from oracle.torch_ref import TorchCpuRef
    g = _load("backbone.npz")
    net = TorchCpuRef(syn.make_state_dict(21))
    feat = net.features(torch.from_numpy(syn.make_images(31, 4, hw))).numpy()
    _close(feat, g["hw%d.eval_feat" % hw], what="torch-cpu feat")


def test_torch_cpu_ref_epoch_matches_numpy_oracle_step():
    """finetune_epoch (autograd + hand-written SGD) against the NumPy restatement's loss / gradient / update."""
    torch = pytest.importorskip("torch")
    from oracle import torch_ref as tr
    rs = np.random.RandomState(4)
    n_base, k_prev, k_new, n, d = 60, 5, 5, 40, 640
    feat = rs.standard_normal((n, d)).astype(np.float32)
    wb = (rs.standard_normal((n_base, d)) * 0.05).astype(np.float32)
    prev = (rs.standard_normal((k_prev, d)) * 0.05).astype(np.float32)
    W = np.concatenate([wb + 0.01 * rs.standard_normal(wb.shape), prev + 0.01 * rs.standard_normal(prev.shape),
                        rs.standard_normal((k_new, d)) * 0.03]).astype(np.float32)
    y = rs.randint(0, n_base + k_prev + k_new, n)
    hp = dict(lmbd_base=0.2, lmbd_prev=0.1, pull=1.0, lr=0.002, momentum=0.9, wd=5e-4)

    class _Net:
        def features(self, x):
            return x
    Wt = torch.from_numpy(W.copy())
    qy = rs.randint(0, 70, 30)
    loss, accs, mom = tr.finetune_epoch(_Net(), Wt, None, torch.from_numpy(wb), torch.from_numpy(prev), torch.from_numpy(feat),
                                        torch.from_numpy(y), [(torch.from_numpy(feat[:30]), torch.from_numpy(qy))], hp)
    l_ce, dlog = loop_ref.cross_entropy(feat @ W.T, y)
    grad = dlog.T @ feat.astype(np.float64)
    l1, g1 = sr.frob_reg_and_grad(0.2, W[:60], wb)
    l2, g2 = sr.frob_reg_and_grad(0.1, W[60:65], prev)
    l3, g3 = sr.loss1_and_grad(1.0, wb, W[65:])
    grad[:60] += g1
    grad[60:65] += g2
    grad[65:] += g3
    _close(loss, l_ce + l1 + l2 + l3, 1e-5, 1e-5, "loss")
    want = W - 0.002 * (grad.astype(np.float32) + np.float32(5e-4) * W)
    _close(Wt.numpy(), want, 1e-6, 1e-5, "W after SGD")
    acc_ref, _ = loop_ref.accuracy_top1(feat[:30] @ want.T, qy)
    assert abs(accs[0] - acc_ref) < 1e-4


def test_torch_train_step_oracle_fp32_mode_matches_reference_autograd():
    """oracle/torch_ref.py::train_step_grads with its bf16 storage emulation switched OFF is the reference's own computation
    (train_supervised.py:229-244 through models/resnet_language.py:268-301): loss and every parameter gradient of the
    reference-generated golden (tests/golden/train_step.npz, hw 32) are reproduced.  The bf16 mode (same code with rounding
    at the HIP path's storage points) is what the GPU test gates the bf16 kernels against."""
    from oracle import torch_ref
    from oracle.resnet_ref import MaskSource
    from subreg_hip import synthetic as syn
    g = np.load(os.path.join(GOLDEN, "train_step.npz"))
    key = "hw32"
    loss, grads = torch_ref.train_step_grads(syn.make_state_dict(71), syn.make_images(72, int(g[key + ".B"]), 32), g[key + ".labels"],
                                             MaskSource(74), bf16=False)
    assert abs(loss - float(g[key + ".loss"])) < 1e-5
    n = 0
    for k in g.files:
        if k.startswith(key + ".grad."):
            name = k[len(key) + 6:]
            want = g[k].astype(np.float64)
            got = grads[name][:want.shape[0]].astype(np.float64)
            assert np.linalg.norm(got - want) <= 1e-3 * max(np.linalg.norm(want), 1e-20), name
            n += 1
    assert n > 40
    # the bf16 mode moves the loss by bf16-rounding noise only and keeps every gradient's direction
    loss_b, grads_b = torch_ref.train_step_grads(syn.make_state_dict(71), syn.make_images(72, int(g[key + ".B"]), 32), g[key + ".labels"],
                                                 MaskSource(74), bf16=True)
    assert abs(loss_b - loss) < 2e-2
    for name in ("classifier.weight", "layer4.1.conv3.weight", "layer1.0.conv1.weight"):
        a, b = grads_b[name].astype(np.float64).ravel(), grads[name].astype(np.float64).ravel()
        assert a @ b / (np.linalg.norm(a) * np.linalg.norm(b)) > 0.85, name


def test_backward_from_stash_equals_the_full_oracle_step():
    """oracle/backward_ref.py::backward_from_stash (the backward started from a GIVEN forward stash: what the GPU test feeds the
    HIP path's own bf16 stash into) reproduces train_step's gradients exactly when handed train_step's own stash - and
    train_step is pinned by the reference's autograd golden above."""
    from oracle import backward_ref as br
    sd = syn.make_state_dict(71)
    x = syn.make_images(72, 4, 32)
    y = np.array([1, 5, 7, 30])
    stash = {}
    from oracle import resnet_ref as rr
    loss, _logits, grads = br.train_step(rr.copy_state_dict(sd), x, y, masks=MaskSource(74), stash_out=stash)
    loss2, grads2 = br.backward_from_stash(sd, stash, x, y)
    assert abs(loss - loss2) < 1e-6
    assert set(grads) == set(grads2)
    for k in grads:
        _close(grads2[k], grads[k], 1e-9 * max(float(np.abs(grads[k]).max()), 1e-12), 1e-9, k)
