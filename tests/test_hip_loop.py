"""GPU parity of the whole incremental loop against the goldens produced by the reference loop itself
(eval/language_eval.py::few_shot_finetune_incremental_test run on CPU by tools/make_golden.py).

f32 mode: per-epoch losses, per-session accuracies, stop epochs and the learned classifier rows must match
the reference within the north_star's fp32 tolerance (1e-4 on weights; losses ~5 => 2e-4 abs+rel).
bf16 mode (ALWAYS compared; fixed-epoch goldens must run exactly the golden's epochs): per-session accuracy within
+-2 query images = 1.6 points at these 125-image sets (the north_star's +-0.1 % is for the 10-seed average),
per-epoch loss 5e-2 / 2e-2, 5e-3 on the learned weights.
The module-surface test drives the REFERENCE's own loop body (net(x), criterion, regloss, LangPuller,
torch SGD) over the drop-in modules and checks it against the fused loop.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle.resnet_ref import MaskSource                                  # noqa: E402
from subreg_hip import synthetic as syn                                   # noqa: E402

from conftest import GOLDEN                                               # noqa: E402
from test_hip_kernels import _cmp                                         # noqa: E402


class _Loader(list):
    def __init__(self, items, label2human):
        super().__init__(items)
        self.dataset = SimpleNamespace(label2human=label2human)


def make_opt(**kw):
    o = SimpleNamespace(no_dropblock=True, linear_bias=False, temperature=1, word_embed_size=500,
                        word_embed_path="word_embeds", dataset="miniImageNet", use_synonyms=False, glove=False,
                        track_weights=False, track_label_inspired_weights=False, save_preds_0=False, set_seed=1,
                        memory_replay=0, neval_episodes=8, continual=False, n_ways=5, n_shots=5, n_queries=25,
                        label_pull=1.0, pulling="regularize", attraction_override="distance2subspace",
                        classifier="linear", attention=None, lmbd_reg_transform_w=0.2, lmbd_reg_novel=0.1,
                        target_train_loss=0.0, convergence_epsilon=1e-4, stable_epochs=10, max_novel_epochs=1000,
                        min_novel_epochs=20, learning_rate=0.002, momentum=0.9, weight_decay=5e-4, adam=False,
                        freeze_backbone_at=1, hip_dtype="f32")
    o.__dict__.update(kw)
    return o


def write_embed_pickle(g, dirpath):
    """The word vectors a semantic / mapping golden was generated with, as the pickle LangPuller reads
    (<word_embed_path>/<dataset>_dim<size>.pickle, resnet_language.py:31)."""
    import pickle
    os.makedirs(dirpath, exist_ok=True)
    with open(os.path.join(dirpath, "miniImageNet_dim500.pickle"), "wb") as f:
        pickle.dump({str(w): np.asarray(v) for w, v in zip(g["embed.words"], g["embed.vecs"])}, f)
    return dirpath


def build_case(g, dtype, embed_dir=None):
    """Model + loaders of a loop golden (inputs regenerated from seeds, BN stats / classifier from the fixture)."""
    from subreg_hip.resnet_language import create_model
    hw, ns, seed = int(g["hw"]), int(g["n_sessions"]), int(g["seed"])
    signal, memory = float(g["signal"]), bool(int(g["memory"]))
    kw = {k[4:]: g[k].item() for k in g.files if k.startswith("opt.")}
    if "attraction_override" in g.files:
        ao = str(g["attraction_override"])
        kw["attraction_override"] = None if ao == "None" else ao
    if "embed.words" in g.files:
        kw["word_embed_path"] = write_embed_pickle(g, embed_dir)
    if "final_bias" in g.files and int(g["opt.lmbd_reg_novel_is_none"]):
        kw["lmbd_reg_novel"] = None
    kw.pop("lmbd_reg_novel_is_none", None)
    opt = make_opt(set_seed=seed, neval_episodes=ns, memory_replay=1 if memory else 0, hip_dtype=dtype, **kw)
    sd = syn.make_state_dict(int(g["sd_seed"]))
    if "final_bias" in g.files:                      # classifier with bias (eval_incremental.py:96-103: read off the checkpoint)
        sd["classifier.bias"] = syn.make_classifier_bias(int(g["sd_seed"]))
    for k in g.files:
        if k.startswith("bn0."):
            sd[k[4:]] = g[k].copy()
        if k.startswith("param."):                   # backbone parameters the generator calibrated (tools/make_golden.py::centre_features)
            sd[k[6:]] = g[k].copy()
    sd["classifier.weight"] = g["base_classifier"].copy()
    net = create_model("resnet18", 60, opt)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net = net.cuda()
    net.mask_source = MaskSource(int(g["mask_seed"]))
    pg = int(g["proto_grid"]) if "proto_grid" in g.files else 0
    hq = int(g["hard_queries"]) if "hard_queries" in g.files else 0
    sessions = syn.make_sessions(seed, ns, hw, class_signal=signal, proto_grid=pg, hard_queries=hq)
    bx, by = syn.make_base_batch(seed, int(g["n_base_batch"]), hw, class_signal=signal, proto_grid=pg)
    names_base = ["b%d" % i for i in range(60)] + [""] * 40
    names_novel = ["n%d" % i for i in range(100)]
    if "names_base" in g.files:
        names_base, names_novel = [str(n) for n in g["names_base"]], [str(n) for n in g["names_novel"]]
    base_loader = _Loader([(torch.from_numpy(bx), torch.from_numpy(by), torch.arange(len(by)))], names_base)
    meta = _Loader([(torch.from_numpy(s["support_xs"])[None], torch.from_numpy(s["support_ys"])[None],
                     torch.from_numpy(s["query_xs"])[None], torch.from_numpy(s["query_ys"])[None]) for s in sessions],
                   names_novel)
    bsl = None
    if memory:
        sx, sy = syn.make_base_support(seed, hw, class_signal=signal, proto_grid=pg)
        bsl = _Loader([(torch.from_numpy(sx)[None], torch.from_numpy(sy)[None], torch.zeros(1, 1, 3, hw, hw),
                        torch.zeros(1, 1, dtype=torch.long))], names_base)
    inits = syn.make_novel_inits(seed, ns)
    picks = [p for p in g["picks"]]
    return net, opt, meta, base_loader, bsl, inits, picks


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("tag", ["hw32_noM", "hw32_M", "hw32_stop", "hw84_M", "hw32_sem", "hw32_map", "hw84_noM_s8", "hw84_noM_disc",
                                 "hw84_stop", "hw84_sem", "hw84_map", "hw32_adam", "hw32_bias", "hw32_freeze3",
                                 "hw32_freeze3_adam", "hw32_freeze5_M"])
def test_fused_loop_against_reference_golden(tag, dtype, tmp_path):
    """hw84_noM_s8 is the bench-scale case (BASELINE.json configs[1]: 8 sessions, -M, 84x84, 1000-image base batch) with 6
    epochs per session, so the per-epoch hipGraph is captured and replayed and up to 1125 images go through one launch
    sequence - the code path bench.py times.
    hw84_noM_disc / hw84_stop are the DISCRIMINATING 84x84 goldens (tools/make_golden.py loop84d): 30 epochs x 2 sessions and a
    run that ends on the stable-epochs rule (language_eval.py:298-318) at epoch 48 of 80; their session accuracies sit at
    58-80 % and the base accuracy falls 95 -> 40 -> 20 %, with the wrong answers wrong by construction (hard_queries), so the
    bf16 gate on them is ONE query image.  hw84_sem / hw84_map: the semantic subspace regularizer and the linear-mapping target
    (scripts/continual/slurm_semantic_subspace_reg.sh, slurm_linear_mapping.sh) on the same kind of episodes at 84x84, 30 epochs,
    with the reference's own word vectors.  hw32_adam: `--adam` (eval/util.py:92-97, torch.optim.Adam instead of SGD), +M, 3 sessions.
    hw32_bias: a classifier WITH bias (backbone pretrained without --no_linear_bias), +M, 3 sessions, no --lmbd_reg_novel.
    hw32_freeze3: freeze_backbone_at = 3 (language_eval.py:243, eval/util.py:62-69): epochs 1-2 of the first session fine-tune
    the WHOLE network (train mode, then eval mode), the backbone freezes at epoch 3; the golden pins what the backbone became.
    hw32_freeze3_adam: the same under --adam (ONE Adam over net.parameters() steps the backbone too).  hw32_freeze5_M: the backbone
    never freezes (3 epochs per session, freeze_backbone_at = 5) and sessions 2-3 replay memory: two gradient-carrying forwards per
    step (support, memory; language_eval.py:252-258), train mode in epoch 1."""
    from subreg_hip.incremental import few_shot_finetune_incremental_test
    g = np.load(os.path.join(GOLDEN, "loop_%s.npz" % tag))
    net, opt, meta, base_loader, bsl, inits, picks = build_case(g, dtype, str(tmp_path / "word_embeds"))
    ckpt = {}
    if "mapping_seed" in g.files:                    # ckpt['mapping_linear_label2image'] = LinearMap state_dict (:225-226)
        mw, mb = syn.make_linear_map(int(g["mapping_seed"]))
        ckpt = {"mapping_linear_label2image": {"map.weight": torch.from_numpy(mw), "map.bias": torch.from_numpy(mb)}}
    novel_avg, base_avg = few_shot_finetune_incremental_test(net, ckpt, None, meta, base_loader, opt, base_support_loader=bsl,
                                                             novel_inits=inits, memory_picks=picks, epochs_per_sync=4,
                                                             verbose=False,
                                                             novel_bias_inits=syn.make_novel_bias_inits(int(g["seed"]), int(g["n_sessions"])))
    run = net.last_run
    ns = int(g["n_sessions"])
    f32 = dtype == "f32"
    data_dependent_stop = tag in ("hw32_stop", "hw84_stop")   # every other golden runs a fixed number of epochs (max_novel_epochs)
    discriminating = "hard_queries" in g.files        # accuracies far from chance, no image near a decision boundary
    one_image = 100.0 / 125 + 1e-6                    # one query image of a 125-image set, in accuracy points
    for s in range(ns):
        want_e = int(g["s%d.epochs" % s])
        want_acc = np.round(g["s%d.last_val" % s], 2)
        if f32:
            assert run["epochs"][s] == want_e, (tag, s, run["epochs"], want_e)
            _cmp("loss s%d" % s, run["loss"][s], g["s%d.loss" % s], 2e-4, 2e-4)
            _cmp("val acc s%d" % s, run["test_acc"][s], want_acc, 1e-6, 0)
            continue
        # bf16 (the dtype of BASELINE.json configs[1]): ALWAYS compared.  Fixed-epoch goldens must run exactly those epochs;
        # the data-dependent stop may move by at most two epochs (bf16 features shift the loss plateau slightly).
        if data_dependent_stop:
            assert abs(run["epochs"][s] - want_e) <= 2, (tag, s, run["epochs"], want_e)
        else:
            assert run["epochs"][s] == want_e, (tag, s, run["epochs"], want_e)
        n = min(run["epochs"][s], want_e)
        _cmp("loss s%d" % s, run["loss"][s][:n], g["s%d.loss" % s][:n], 5e-2, 2e-2)
        same = run["epochs"][s] == want_e
        # accuracy: within 2 query images per 125-image set.  These short goldens sit near chance level, where many
        # argmaxes are near-ties between freshly initialised novel rows; bf16 activations (8-bit mantissa, ~3e-3 relative
        # on the features) flip up to two of them (measured: 28.0 vs 26.4 on hw32_sem), which no kernel change can avoid
        _cmp("val acc s%d" % s, run["test_acc"][s], want_acc, (1 if discriminating else (2 if same else 3)) * one_image, 0)
    if f32 or not data_dependent_stop:
        # bf16: the learned rows are sums of (softmax weight x feature) over the epochs, so they inherit the features' relative
        # error (8-bit mantissa activations through 22 layers: ~1e-2 of the feature scale).  The short lr = 0.002 goldens move the
        # rows by < 0.05, hence 5e-3 absolute; the discriminating goldens (lr 0.15 x 30 epochs) learn rows of norm ~1-1.5: gated
        # on the matrix as a whole (relative L2 <= 4e-2; measured 1.4e-2 / 2.1e-2 / 1.6e-2 on hw84_noM_disc / _sem / _map: the
        # features' own bf16 error, ~1e-2 of their scale) plus 5e-2 of the largest weight on every element
        # (measured: <= 2.0e-2 on hw84_noM_disc, 3.5e-2 on ONE of 41600 elements of hw84_map)
        wmax = float(np.abs(g["final_classifier"]).max())
        tol_bf16 = 5e-2 * wmax if discriminating else 5e-3
        if "opt.adam" in g.files and bool(g["opt.adam"]):
            # Adam divides by sqrt(v): an element whose gradient is within bf16 feature noise of zero moves by +-lr per step
            # either way, so two correct implementations can differ by 2 * lr per epoch on it (measured: 23 of 48000 elements
            # up to 9.2e-3 after 5 epochs at lr 0.002); fp32 stays at 1e-4
            tol_bf16 = 2.0 * float(g["opt.learning_rate"] if "opt.learning_rate" in g.files else 0.002) * int(g["opt.max_novel_epochs"])
        _cmp("final classifier", run["classifier_weight"], g["final_classifier"], 1e-4 if f32 else tol_bf16, 1e-4 if f32 else 5e-3)
        if "final_bias" in g.files:
            _cmp("final bias", run["classifier_bias"], g["final_bias"], 1e-5 if f32 else 2e-3, 1e-4 if f32 else 5e-3)
        if discriminating and not f32:
            dw = run["classifier_weight"].astype(np.float64) - g["final_classifier"].astype(np.float64)
            rel = float(np.linalg.norm(dw) / np.linalg.norm(g["final_classifier"].astype(np.float64)))
            print("final classifier relative L2 (bf16, %s): %.4f" % (tag, rel))
            assert rel < 4e-2, ("final classifier, relative L2", tag, rel)
    if int(g["hw"]) == 84 and int(g["opt.max_novel_epochs"]) > 5:
        # the per-epoch forward was replayed as a hipGraph from epoch 3 on, in every session
        assert all(r >= e - 2 for r, e in zip(run["graph_replays"], run["epochs"])), (run["graph_replays"], run["epochs"])
        if not data_dependent_stop:
            assert all(r == int(g["opt.max_novel_epochs"]) - 2 for r in run["graph_replays"]), run["graph_replays"]
    if "acc_base_sessions" in g.files:
        # eval_base after every session (language_eval.py:363-367): fp32 exact, bf16 within two of the base batch's images
        want_b = np.round(g["acc_base_sessions"][1:], 2)
        assert (want_b > 0).all() or not discriminating
        # (a stop rule that fires an epoch or two apart under bf16 moves the base accuracy with it: three images there)
        _cmp("base acc per session", run["acc_base"], want_b,
             1e-6 if f32 else (300.0 if data_dependent_stop else 200.0) / int(g["n_base_batch"]) + 1e-6, 0)
    if "final.requires_grad" in g.files:
        # the backbone was fine-tuned before the freeze: its parameters must have moved exactly as far, and be frozen now
        assert [int(p.requires_grad) for n, p in net.named_parameters() if not n.startswith("classifier")] == list(g["final.requires_grad"])
        assert net.classifier.weight.requires_grad
        sd0, sdn = syn.make_state_dict(int(g["sd_seed"])), net.state_dict()
        for k in g.files:
            if not k.startswith("final_delta_norm."):
                continue
            name = k[len("final_delta_norm."):]
            got = sdn[name].detach().cpu().numpy()
            moved = float(np.linalg.norm((got - sd0[name]).astype(np.float64)))
            want_moved = float(g[k])
            assert want_moved > 0
            many_steps = tag in ("hw32_freeze3_adam", "hw32_freeze5_M")         # (see `loose` below)
            # (bf16, measured: within 4.5 % on every pinned tensor of the three cases)
            assert abs(moved - want_moved) <= ((1e-2 if many_steps else 2e-3) if f32 else 0.08) * want_moved, (name, moved, want_moved)
            want = g["final." + name]
            # f32: element-wise on the update itself (the weights moved by ~1e-5 per element: compare the DIFFERENCE to the start)
            d_got, d_want = (got[:want.shape[0]] - sd0[name][:want.shape[0]]).astype(np.float64), (want - sd0[name][:want.shape[0]]).astype(np.float64)
            l2 = float(np.linalg.norm(d_got - d_want) / max(np.linalg.norm(d_want), 1e-30))
            # hw32_freeze3 (two SGD steps): 1e-2.  Adam moves an element whose gradient is within rounding noise of zero by +-lr
            # either way, and hw32_freeze5_M makes nine whole-network steps over three sessions, each fed by the previous one's
            # LeakyReLU / MaxPool decisions: measured 2.6e-2 / 2.7e-2 on layer1.0.conv1.weight in f32 with every loss within 7e-6
            # of the reference's (the joint two-forward backward itself is pinned at 2e-5 by test_hip_train.py)
            loose = tag in ("hw32_freeze3_adam", "hw32_freeze5_M")
            # bf16, measured (the reference run is f32): 0.007-0.30 under SGD, growing from the last layer (its gradient is exact up to
            # the features' rounding) to the first (every LeakyReLU / MaxPool decision between flips some elements); Adam turns an
            # element whose gradient is within that noise of zero into +-lr: 0.07-0.57
            adam = tag == "hw32_freeze3_adam"
            assert l2 < ((5e-2 if loose else 1e-2) if f32 else (0.65 if adam else 0.4)), ("backbone update", name, l2)
            # direction of the update against the f32 reference run (round-5 advisor: the L2 gate alone pins little in bf16).
            # Measured in bf16: >= 0.955 under SGD (0.9994-1.0 on layer 4's tensors), >= 0.829 under Adam (>= 0.978 on layer4.1's)
            cos = float((d_got * d_want).sum() / max(np.linalg.norm(d_got) * np.linalg.norm(d_want), 1e-30))
            print("backbone update %s %s (%s): relative L2 %.4f, cosine %.4f, moved %.4g / %.4g" % (tag, name, "f32" if f32 else "bf16", l2, cos, moved, want_moved))
            last_block = name.startswith("layer4.1.")
            want_cos = 0.998 if f32 else ((0.96 if last_block else 0.78) if adam else (0.998 if last_block else 0.93))
            assert cos > want_cos, ("backbone update direction", name, cos, want_cos)
    if f32:
        _cmp("novel avg", novel_avg, g["novel_avg"], 1e-5, 1e-6)
        _cmp("base avg", base_avg, g["base_avg"], 1e-5, 1e-6)
        sd = net.state_dict()
        # (running statistics taken over a backbone that was itself trained for several steps: the weights' 1e-2 relative update
        #  noise shows up as a few 1e-5 on the deepest layer's means)
        st_atol = 1e-4 if tag in ("hw32_freeze3_adam", "hw32_freeze5_M") else 1e-5
        for k in ("layer1.0.bn1", "layer4.1.bn3"):
            _cmp(k + ".running_mean", sd[k + ".running_mean"].cpu().numpy(), g[k + ".running_mean"], st_atol, 1e-4)
            _cmp(k + ".running_var", sd[k + ".running_var"].cpu().numpy(), g[k + ".running_var"], st_atol, 1e-4)
    else:
        _cmp("novel avg", novel_avg, g["novel_avg"], (2 if not data_dependent_stop else 3) * one_image, 0)
        _cmp("base avg", base_avg, g["base_avg"],                                                       # <= 1 base image / 0.5 pt
             max(0.5, (300.0 if data_dependent_stop and discriminating else 100.0) / int(g["n_base_batch"]) + 1e-6), 0)


def test_bias_with_novel_reg_fails_like_the_reference():
    """resnet_language.py:238 indexes the 1-D bias with two indices: a classifier with bias and --lmbd_reg_novel dies with
    IndexError in the first epoch of session 2 in the reference; the fused loop raises the same error at the same place
    (session 1 runs), and the C entry point refuses the combination."""
    from subreg_hip.incremental import IncrementalRunner
    g = np.load(os.path.join(GOLDEN, "loop_hw32_bias.npz"))
    net, opt, meta, base_loader, bsl, inits, picks = build_case(g, "f32")
    opt.lmbd_reg_novel, opt.max_novel_epochs = 0.1, 2
    r = IncrementalRunner(net, meta, base_loader, opt, bsl, inits, picks, verbose=False,
                          novel_bias_inits=syn.make_novel_bias_inits(int(g["seed"]), 3)).start()
    assert r.run_session(0) == 2
    with pytest.raises(IndexError, match="too many indices"):
        r.run_session(1)


def test_feature_reuse_is_results_identical():
    """The opt-in frozen-feature reuse must not change a single result (it only skips constant recomputation)."""
    from subreg_hip.incremental import few_shot_finetune_incremental_test
    g = np.load(os.path.join(GOLDEN, "loop_hw32_M.npz"))
    outs = []
    for reuse in (False, True):
        net, opt, meta, base_loader, bsl, inits, picks = build_case(g, "f32")
        few_shot_finetune_incremental_test(net, {}, None, meta, base_loader, opt, base_support_loader=bsl, novel_inits=inits,
                                           memory_picks=picks, reuse_features=reuse, verbose=False)
        outs.append(net.last_run)
    assert outs[0]["epochs"] == outs[1]["epochs"] and outs[0]["test_acc"] == outs[1]["test_acc"]
    assert np.array_equal(outs[0]["classifier_weight"], outs[1]["classifier_weight"])


def test_feature_reuse_after_pre_freeze_epochs_is_results_identical():
    """freeze_backbone_at = 3 with the opt-in feature reuse: the ONE eval-mode forward the reuse keeps must come AFTER the last
    backbone update of the pre-freeze epochs (round-4 advisor finding: with K - 1 >= 2 pre-freeze epochs no forward ran at all and
    the classifier steps of the frozen epochs trained on the support features of the backbone before its last SGD step)."""
    from subreg_hip.incremental import few_shot_finetune_incremental_test
    g = np.load(os.path.join(GOLDEN, "loop_hw32_freeze3.npz"))
    assert int(g["opt.freeze_backbone_at"]) >= 3
    outs = []
    for reuse in (False, True):
        net, opt, meta, base_loader, bsl, inits, picks = build_case(g, "f32")
        few_shot_finetune_incremental_test(net, {}, None, meta, base_loader, opt, base_support_loader=bsl, novel_inits=inits,
                                           memory_picks=picks, reuse_features=reuse, verbose=False)
        outs.append(net.last_run)
    assert outs[0]["epochs"] == outs[1]["epochs"] and outs[0]["test_acc"] == outs[1]["test_acc"]
    # (not bitwise: the two runs fine-tune the backbone separately, and the float atomics of the first-layer / 1x1 dW kernels leave
    # their last bits run-dependent - measured 5e-6 ... 1.4e-5 relative on the losses of the frozen epochs.  The stale-feature bug
    # moved them in the third digit.)
    for la, lb in zip(outs[0]["loss"], outs[1]["loss"]):
        np.testing.assert_allclose(np.asarray(la), np.asarray(lb), rtol=1e-4, atol=0)
    np.testing.assert_allclose(outs[0]["classifier_weight"], outs[1]["classifier_weight"], rtol=0, atol=2e-5)


def test_module_surface_runs_reference_loop_body():
    """The drop-in modules under the reference's own loop statements (language_eval.py:242-326), torch autograd + SGD,
    give the same numbers as the fused loop: 1 session, 4 epochs."""
    from subreg_hip.incremental import few_shot_finetune_incremental_test
    from subreg_hip.resnet_language import LangPuller
    g = np.load(os.path.join(GOLDEN, "loop_hw32_noM.npz"))
    net, opt, meta, base_loader, bsl, inits, picks = build_case(g, "f32")
    criterion = torch.nn.CrossEntropyLoss()
    base_weight, base_bias = net._get_base_weights()
    sx, sy, qx, qy = meta[0]
    support_xs, query_xs = sx[0].cuda(), qx[0].cuda()
    ids = {int(c): 60 + r for r, c in enumerate(np.sort(np.unique(qy[0].numpy())))}
    support_ys_id = torch.tensor([ids[int(y)] for y in sy[0]]).cuda()
    query_ys_id = torch.tensor([ids[int(y)] for y in qy[0]]).cuda()
    net.eval()
    with torch.no_grad():
        net(torch.from_numpy(syn.make_base_batch(int(g["seed"]), int(g["n_base_batch"]), int(g["hw"]),
                                                 class_signal=float(g["signal"]))[0]).cuda())   # eval_base forward (:128)
    net.train()
    net.augment_base_classifier_(5, novel_weight=torch.from_numpy(inits[0]))
    puller = LangPuller(opt, ["b"] * 60, ["n"] * 5)
    for name, prm in net.named_parameters():                  # freeze_backbone_weights (eval/util.py:62-69)
        prm.requires_grad = name.startswith("classifier")
    optimizer = torch.optim.SGD(net.parameters(), lr=opt.learning_rate, momentum=opt.momentum, weight_decay=opt.weight_decay)
    losses, accs = [], []
    for epoch in range(1, 5):
        output = net(support_xs)
        loss = criterion(output, support_ys_id)
        loss = loss + net.regloss(opt.lmbd_reg_transform_w, base_weight, base_bias)
        pullers = puller.get_projected_weight(base_weight, net.classifier.weight[60:, :])
        loss = loss + puller.loss1(opt.label_pull, pullers, net.classifier.weight[60:, :])
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
        net.eval()
        with torch.no_grad():
            out = net(query_xs)
            accs.append((out.argmax(1) == query_ys_id).float().sum().item() * (100.0 / len(query_ys_id)))
    _cmp("module-surface loss", losses, g["s0.loss"], 2e-4, 2e-4)
    _cmp("module-surface acc", round(accs[-1], 2), np.round(g["s0.last_val"], 2)[0], 1e-6, 0)


def test_semantic_regularizer_module_against_reference_golden(tmp_path):
    """LangPuller.forward on the fused HIP kernel (+ autograd to W_base), update_novel_embeds, the mapping variant, loss1."""
    from subreg_hip.resnet_language import LangPuller
    g = np.load(os.path.join(GOLDEN, "semantic.npz"))
    path = write_embed_pickle(g, str(tmp_path / "word_embeds"))
    vb = [str(n) for n in g["vocab_base"]]
    v0, v1 = [str(n) for n in g["vocab_novel0"]], [str(n) for n in g["vocab_novel1"]]
    wb = torch.from_numpy(g["w_base"]).cuda()
    for temp in (1.0, 3.0):
        key = "t%g" % temp
        opt = make_opt(attraction_override=None, temperature=temp, word_embed_path=path)
        puller = LangPuller(opt, vb, v0)
        wbt = wb.clone().requires_grad_(True)
        pl = puller(wbt)
        _cmp("pullers0", pl.detach().cpu().numpy(), g[key + ".pullers0"], 1e-5, 1e-4)
        pl.backward(torch.from_numpy(g[key + ".grad_out"]).cuda())
        _cmp("d W_base", wbt.grad.cpu().numpy(), g[key + ".grad_w_base"], 1e-5, 1e-4)
        _cmp("masked", puller(wb, mask=True).cpu().numpy(), g[key + ".pullers0_masked"], 1e-5, 1e-4)
        w = torch.from_numpy(g[key + ".w"]).cuda().requires_grad_(True)
        loss = puller.loss1(0.7, puller(wb), w)
        loss.backward()
        _cmp("loss1", loss.item(), g[key + ".loss1"], 1e-5, 1e-4)
        _cmp("loss1 grad", w.grad.cpu().numpy(), g[key + ".loss1_grad"], 1e-5, 1e-4)
        puller.update_novel_embeds(v1)
        _cmp("pullers1", puller(wb).cpu().numpy(), g[key + ".pullers1"], 1e-5, 1e-4)
    opt = make_opt(attraction_override="mapping_linear_label2image", word_embed_path=path)
    puller = LangPuller(opt, vb, v0)
    puller.create_pulling_mapping({"map.weight": torch.from_numpy(g["map.weight"]), "map.bias": torch.from_numpy(g["map.bias"])})
    _cmp("mapping pullers", puller(wb).cpu().numpy(), g["map.pullers0"], 1e-5, 1e-4)


_ORACLE_351 = {}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("scale", [(32, 2, 3)], ids=["hw32"])     # ((84, 1, 2) works too: ~6 minutes of NumPy oracle for accuracies at chance)
def test_fused_loop_351_base_classes_against_oracle(dtype, scale):
    """BASELINE.json configs[4]: tieredImageNet-sized base set (351 classes, train_supervised.py:94), +M, 2 sessions x 3 epochs at
    32x32.  The reference cannot run this configuration (eval_incremental.py:82-83 raises), so parity is against the NumPy
    restatement of its loop (oracle/loop_ref.py, pinned by the 60-class goldens) with n_base = 351.  Both weightings of the
    running average are checked: the reference's hard-coded 200/60 (language_eval.py:383-386) and the explicit
    avg_weights_follow_n_base flag."""
    from oracle import loop_ref
    from oracle.resnet_ref import ResNetRef, copy_state_dict
    from subreg_hip.incremental import few_shot_finetune_incremental_test
    from subreg_hip.resnet_language import create_model
    # hw84: the same configuration at the reference's image size (one session, 476 support + exemplar images per forward, the
    # production 84x84 kernels incl. the fused layer 1); one weighting only (the NumPy oracle needs ~2 minutes for it)
    hw, ns, n_epochs = scale
    NB, seed, signal = 351, 9, 3.0
    for follow in ((False, True) if hw == 32 else (True,)):
        opt = make_opt(set_seed=seed, neval_episodes=ns, memory_replay=1, hip_dtype=dtype, max_novel_epochs=n_epochs,
                       dataset="tieredImageNet", avg_weights_follow_n_base=follow)
        sd = syn.make_state_dict(40, n_cls=NB)
        sessions = syn.make_sessions(seed, ns, hw, class_signal=signal, first_novel=NB)
        bx, by = syn.make_base_batch(seed, 64, hw, n_base=NB, class_signal=signal)
        sx, sy = syn.make_base_support(seed, hw, n_base=NB, class_signal=signal)
        inits = syn.make_novel_inits(seed, ns)
        picks = [np.array([1]), np.array([3])][:ns]
        # ONE oracle run (~100 s of NumPy) for both HIP dtypes and both weightings: the weighting only enters the running average
        # (language_eval.py:383-393: the rounded (w1 * base + w2 * novel) / (w1 + w2)), which is recomputed here from the oracle's
        # un-rounded per-session values for the other weighting
        if hw not in _ORACLE_351:
            _ORACLE_351[hw] = loop_ref.run_incremental(ResNetRef(copy_state_dict(sd)), sessions, (bx, by), opt, inits,
                                                       base_support=(sx, sy), masks=MaskSource(77), memory_picks=picks, n_base=NB)
            _ORACLE_351[hw]["_follow"] = follow
        want = dict(_ORACLE_351[hw])
        if want.pop("_follow") != follow:
            wa = [want["weighted_avg"][0]]                            # (the entry before session 1 is the base accuracy alone)
            for k in range(ns):
                w1, w2 = (NB, 5 * (k + 1)) if follow else (200, NB + 5 * (k + 1) - 60)
                wa.append(round((w1 * want["base_vals"][k] + w2 * want["novel_vals"][k]) / (w1 + w2), 2))
            want["weighted_avg"] = wa
        net = create_model("resnet18", NB, opt, dataset="tieredImageNet")
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        net = net.cuda()
        net.mask_source = MaskSource(77)
        names_base = ["b%d" % i for i in range(NB)]
        base_loader = _Loader([(torch.from_numpy(bx), torch.from_numpy(by), torch.arange(len(by)))], names_base)
        meta = _Loader([(torch.from_numpy(s["support_xs"])[None], torch.from_numpy(s["support_ys"])[None],
                         torch.from_numpy(s["query_xs"])[None], torch.from_numpy(s["query_ys"])[None]) for s in sessions],
                       ["n%d" % i for i in range(NB + 100)])
        bsl = _Loader([(torch.from_numpy(sx)[None], torch.from_numpy(sy)[None], torch.zeros(1, 1, 3, hw, hw),
                        torch.zeros(1, 1, dtype=torch.long))], names_base)
        few_shot_finetune_incremental_test(net, {}, None, meta, base_loader, opt, base_support_loader=bsl, novel_inits=inits,
                                           memory_picks=picks, verbose=False)
        run = net.last_run
        f32 = dtype == "f32"
        assert run["classifier_weight"].shape == (NB + 5 * ns, 640)
        for s in range(ns):
            assert run["epochs"][s] == want["epochs"][s] == n_epochs
            _cmp("loss s%d" % s, run["loss"][s], want["loss"][s], 2e-4 if f32 else 5e-2, 2e-4 if f32 else 2e-2)
            _cmp("val acc s%d" % s, run["test_acc"][s], want["test_acc"][s], 1e-6 if f32 else 200.0 / 125 + 1e-6, 0)
            # validate's top-5 (language_eval.py:40); among 351+ classes it separates from top-1
            # (bf16: the 5th/6th-place gaps among 350+ logits are much denser than the 1st/2nd-place ones - measured 4 of 125
            # rows flipping - so the bf16 gate is 5 images; fp32 is exact)
            _cmp("val top-5 s%d" % s, run["test_acc_top5"][s], want["test_acc_top5"][s], 1e-4 if f32 else 500.0 / 125 + 1e-4, 0)
            assert all(a5 >= a1 - 0.01 for a5, a1 in zip(run["test_acc_top5"][s], run["test_acc"][s]))
        _cmp("final classifier", run["classifier_weight"], want["classifier_weight"], 1e-4 if f32 else 5e-3, 1e-4 if f32 else 5e-3)
        if f32:
            _cmp("weighted avg", run["weighted_avg"], want["weighted_avg"], 1e-6, 0)
            _cmp("acc base", run["acc_base"], want["acc_base"], 1e-6, 0)
        # the two weightings differ exactly as stated: 200/(351+5k-60) vs 351/(5k)
        ab, an = want["acc_base"][-1], want["novel_acc"][-1]
        w1, w2 = (NB, 5 * ns) if follow else (200, NB + 5 * ns - 60)
        assert abs(want["weighted_avg"][-1] - round((w1 * want["base_vals"][-1] + w2 * want["novel_vals"][-1]) / (w1 + w2), 2)) < 1e-9, (ab, an)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_loop_351_base_classes_hw84_against_cached_oracle(dtype):
    """BASELINE.json configs[4], incremental leg at the reference's image size: 351 base classes, +M, 2 sessions x 3 epochs at
    84x84 (476-601 images per forward through the production kernels incl. the fused layer 1).  Expected values: the build's NumPy
    oracle, computed once in the build container (tools/make_oracle_fixture.py, ~11 minutes) and committed as
    tests/golden/oracle_loop351_hw84.npz - NOT a reference golden (the reference cannot run tieredImageNet,
    eval_incremental.py:82-83); the oracle itself is pinned by the reference-generated 60-class goldens."""
    from subreg_hip.incremental import few_shot_finetune_incremental_test
    from subreg_hip.resnet_language import create_model
    g = np.load(os.path.join(GOLDEN, "oracle_loop351_hw84.npz"))
    c = {k[5:]: g[k].item() for k in g.files if k.startswith("case.")}
    NB, seed, signal, hw, ns, n_epochs = int(c["NB"]), int(c["seed"]), float(c["signal"]), int(c["hw"]), int(c["ns"]), int(c["n_epochs"])
    opt = make_opt(set_seed=seed, neval_episodes=ns, memory_replay=1, hip_dtype=dtype, max_novel_epochs=n_epochs,
                   dataset="tieredImageNet", avg_weights_follow_n_base=True)
    sd = syn.make_state_dict(int(c["sd_seed"]), n_cls=NB)
    sessions = syn.make_sessions(seed, ns, hw, class_signal=signal, first_novel=NB)
    bx, by = syn.make_base_batch(seed, int(c["n_base_batch"]), hw, n_base=NB, class_signal=signal)
    sx, sy = syn.make_base_support(seed, hw, n_base=NB, class_signal=signal)
    inits = syn.make_novel_inits(seed, ns)
    picks = [np.array([1]), np.array([3])][:ns]
    net = create_model("resnet18", NB, opt, dataset="tieredImageNet")
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net = net.cuda()
    net.mask_source = MaskSource(int(c["mask_seed"]))
    names_base = ["b%d" % i for i in range(NB)]
    base_loader = _Loader([(torch.from_numpy(bx), torch.from_numpy(by), torch.arange(len(by)))], names_base)
    meta = _Loader([(torch.from_numpy(s["support_xs"])[None], torch.from_numpy(s["support_ys"])[None],
                     torch.from_numpy(s["query_xs"])[None], torch.from_numpy(s["query_ys"])[None]) for s in sessions],
                   ["n%d" % i for i in range(NB + 100)])
    bsl = _Loader([(torch.from_numpy(sx)[None], torch.from_numpy(sy)[None], torch.zeros(1, 1, 3, hw, hw),
                    torch.zeros(1, 1, dtype=torch.long))], names_base)
    few_shot_finetune_incremental_test(net, {}, None, meta, base_loader, opt, base_support_loader=bsl, novel_inits=inits,
                                       memory_picks=picks, verbose=False)
    run = net.last_run
    f32 = dtype == "f32"
    assert run["classifier_weight"].shape == (NB + 5 * ns, 640)
    for s in range(ns):
        assert run["epochs"][s] == int(g["epochs"][s]) == n_epochs
        _cmp("loss s%d" % s, run["loss"][s], g["loss.%d" % s], 2e-4 if f32 else 5e-2, 2e-4 if f32 else 2e-2)
        _cmp("val acc s%d" % s, run["test_acc"][s], g["test_acc.%d" % s], 1e-6 if f32 else 200.0 / 125 + 1e-6, 0)
        _cmp("val top-5 s%d" % s, run["test_acc_top5"][s], g["test_acc_top5.%d" % s], 1e-4 if f32 else 500.0 / 125 + 1e-4, 0)
    _cmp("final classifier", run["classifier_weight"], g["classifier_weight"], 1e-4 if f32 else 5e-3, 1e-4 if f32 else 5e-3)
    if f32:
        _cmp("weighted avg", run["weighted_avg"], g["weighted_avg"], 1e-6, 0)
        _cmp("acc base", run["acc_base"], g["acc_base"], 1e-6, 0)
