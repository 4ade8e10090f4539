"""NumPy restatement of the incremental-session loop (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/eval/language_eval.py::few_shot_finetune_incremental_test (:71-454)
with validate (:18-43), eval_base (:46-69) and the helpers of eval/util.py
(accuracy :26-40, freeze_backbone_weights :62-69, get_optim :92-102 -> torch SGD).
The plain-linear classifier is restated with and without bias (eval_incremental.py:96-103 takes the bias from the
checkpoint; scripts/continual/slurm_run_backbone.sh:39 trains without).  With a bias: logits + b, regloss adds
lmbd * ||b[:n_base] - b_base||**2 (resnet_language.py:231-232, squared unlike the weight term), the optimiser updates it like
any parameter, and reglossnovel indexes the 1-D bias with two indices (:238) - an IndexError from session 2 on whenever
--lmbd_reg_novel is given, restated as such.

Semantics that parity depends on, all restated here:
  * net.train() once per session (:211); validate() switches to eval (:19) and
    nobody switches back => epoch 1 runs the backbone in TRAIN mode (BN batch
    stats + running-stat update + dropout/DropBlock), epochs >= 2 in EVAL mode.
  * loss = CE(support) [+ CE(memory)] + regloss + [reglossnovel] + loss1 (:252-290)
  * fresh SGD(momentum) per session (:231); only classifier.weight has a gradient.
  * stop rule (:298-318), per-epoch validation on every past query set (:321-326),
    memory pick (:353-359), base eval (:363-367), 2-dp rounded bookkeeping (:370-393).
`reuse_features=True` computes each eval-mode feature matrix once per session (the
backbone is frozen and in eval mode from epoch 2 on, so the values are identical)
while still advancing the per-block forward counters that feed DropBlock's gamma.
Parity pinned by tests/golden/loop_*.npz (tools/make_golden.py).
"""
import numpy as np

from . import subspace_ref as sr
from .resnet_ref import linear


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss (mean).  Returns (loss, dlogits)."""
    z = logits.astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    logp = z - lse
    n = logits.shape[0]
    loss = -logp[np.arange(n), labels].mean()
    d = np.exp(logp)
    d[np.arange(n), labels] -= 1.0
    return float(np.float32(loss)), d / n


def accuracy_top1(logits, labels):
    """eval/util.py:26-40 with topk=(1,): percentage (float32 like torch)."""
    pred = np.argmax(logits, axis=1)
    return float(np.float32((pred == labels).sum() * (100.0 / labels.shape[0]))), pred


def accuracy_topk(logits, labels, k=5):
    """eval/util.py:26-40 with topk=(..., k): the label is among the k largest logits (torch.topk breaks ties towards the lower
    index; a stable descending argsort does the same)."""
    order = np.argsort(-logits, axis=1, kind="stable")[:, :k]
    hit = (order == np.asarray(labels)[:, None]).any(axis=1)
    return float(np.float32(hit.sum() * (100.0 / len(labels))))


def memory_indices(pick, n_shots=5):
    """language_eval.py:354-358 for `inds = np.random.choice(n_shots, memory_replay)` == pick."""
    inds = np.asarray(pick)
    margin = 5 * np.arange(5)
    offset = np.arange(0, 125, 25)
    inds = np.tile(margin + inds, (5, 1)) + (np.tile(offset, (5, 1))).T
    return inds.flatten()


class _Bump:
    """Advance BasicBlock.num_batches_tracked as one forward would (feature reuse)."""

    def __init__(self, net):
        self.net = net

    def __call__(self, times=1):
        for k in self.net.nbt:
            self.net.nbt[k] += times


def run_incremental(net, sessions, base_batch, opt, novel_inits, base_support=None,
                    masks=None, memory_picks=None, reuse_features=True, n_base=60, embeds=None, names=None, mapping=None,
                    novel_bias_inits=None):
    """Run `len(sessions)` incremental sessions.  Returns a dict of everything the goldens pin."""
    f32 = np.float32
    bump = _Bump(net)
    W = net.sd["classifier.weight"].astype(f32).copy()          # live classifier.weight
    base_weight = W.copy()                                      # basenet._get_base_weights(), :106-107
    bvec = net.sd.get("classifier.bias")                        # live classifier.bias or None (--no_linear_bias)
    bvec = None if bvec is None else np.asarray(bvec, f32).copy()
    base_bias = None if bvec is None else bvec.copy()
    base_x, base_y = base_batch
    out = dict(loss=[], test_acc=[], test_acc_top5=[], acc_base=[], weighted_avg=[], epochs=[], novel_acc=[],
               memory_inds=[], train_acc=[], novel_vals=[], base_vals=[])

    def eval_feats(x):
        net.eval()
        return net.features(x)

    # initial base evaluation, :128
    net.sd["classifier.weight"] = W
    base_feat = eval_feats(base_x)
    acc_b, _ = accuracy_top1(linear(base_feat, W, bvec), base_y)
    out["weighted_avg"].append(acc_b)

    query_x, query_ids = [], []
    mem_x, mem_y = None, None
    reserve = None
    for idx, sess in enumerate(sessions):
        sx, sy, qx, qy = sess["support_xs"], sess["support_ys"], sess["query_xs"], sess["query_ys"]
        if base_support is not None:
            sx = np.concatenate([sx, base_support[0]], 0)                      # :149-150
        n_old = n_base + idx * opt.n_ways                                      # len(vocab_base) in the loop
        if idx == 1:                                                           # :172-185
            reserve = W[-opt.n_ways:].copy()
        elif idx > 1:
            reserve = np.concatenate([reserve, W[-opt.n_ways:].copy()], 0)
        novel_ids = np.sort(np.unique(qy))
        orig2id = {int(c): n_base + r + idx * opt.n_ways for r, c in enumerate(novel_ids)}   # :193-194
        qid = np.array([orig2id[int(y)] for y in qy], np.int64)
        sid = np.array([orig2id[int(y)] for y in sy], np.int64)
        query_x.append(qx)
        query_ids.append(qid)
        if base_support is not None:
            sid = np.concatenate([sid, np.asarray(base_support[1], np.int64)])  # :207-209
        # constant pullers of the semantic / linear-mapping variants (:216-227): base embeddings of the 60 base names,
        # novel embeddings of THIS session's class names; `names` = (label2human of the base loader, of the meta loader)
        pull_target = None
        if opt.label_pull is not None and getattr(opt, "attraction_override", None) != "distance2subspace":
            vocab_base = [n for n in names[0] if n != ""]
            e_novel = sr.get_embeds(embeds, [names[1][int(c)] for c in novel_ids], opt.word_embed_size)
            if getattr(opt, "attraction_override", None) == "mapping_linear_label2image":
                pull_target = sr.linear_map_target(e_novel, mapping[0], mapping[1]).astype(f32)
            else:
                e_base = sr.get_embeds(embeds, vocab_base, opt.word_embed_size)
                pull_target = sr.semantic_target(e_novel, e_base, base_weight[:n_base], opt.temperature)[0].astype(f32)
        net.train()                                                             # :211
        W = np.concatenate([W, np.asarray(novel_inits[idx], f32)], 0)           # augment_base_classifier_, :214
        net.sd["classifier.weight"] = W
        if bvec is not None:
            bvec = np.concatenate([bvec, np.asarray(novel_bias_inits[idx], f32)])
            net.sd["classifier.bias"] = bvec
        buf = bbuf = None                                                       # fresh SGD, :231
        train_loss, epoch, stable, go = 15, 1, 0, True
        losses, feat_cache = [], {}
        while go:
            # ---- support (+ memory) forward in the CURRENT mode
            if net.training or not reuse_features:
                feat_s = net.features(sx, masks)
                feat_m = net.features(mem_x, masks) if mem_x is not None else None
            else:
                if "s" not in feat_cache:
                    feat_cache["s"] = net.features(sx)
                    feat_cache["m"] = net.features(mem_x) if mem_x is not None else None
                else:
                    bump(1 if mem_x is None else 2)
                feat_s, feat_m = feat_cache["s"], feat_cache["m"]
            logits = linear(feat_s, W, bvec)
            loss, dlog = cross_entropy(logits, sid)                              # :252-253
            grad = dlog.T @ feat_s.astype(np.float64)
            gb = dlog.sum(axis=0)                                                # d loss / d bias (used when there is one)
            loss = f32(loss)
            if feat_m is not None:                                               # :256-258
                l2, d2 = cross_entropy(linear(feat_m, W, bvec), mem_y)
                loss = f32(loss + f32(l2))
                grad += d2.T @ feat_m.astype(np.float64)
                gb = gb + d2.sum(axis=0)
            if opt.lmbd_reg_transform_w is not None:                             # :261-265
                l, g = sr.frob_reg_and_grad(opt.lmbd_reg_transform_w, W[:n_base], base_weight)
                l = f32(l)
                if bvec is not None:                                             # resnet_language.py:231-232: + lmbd * norm(db)**2
                    db = bvec[:n_base].astype(np.float64) - base_bias.astype(np.float64)
                    nb_ = f32(np.sqrt((db * db).sum()))
                    l = f32(l + f32(f32(opt.lmbd_reg_transform_w) * f32(nb_ * nb_)))
                    gb[:n_base] += 2.0 * opt.lmbd_reg_transform_w * db
                loss = f32(loss + l)
                grad[:n_base] += g
            if opt.lmbd_reg_novel is not None and idx > 0:                       # :268-274
                if bvec is not None:                                             # resnet_language.py:238 `bias[rng1:rng2, :]`
                    raise IndexError("too many indices for tensor of dimension 1")
                k = reserve.shape[0]
                l, g = sr.frob_reg_and_grad(opt.lmbd_reg_novel, W[n_base:n_base + k], reserve)
                loss = f32(loss + f32(l))
                grad[n_base:n_base + k] += g
            if opt.label_pull is not None:                                       # :277-290
                if pull_target is not None:
                    l, g = sr.loss1_to_target_and_grad(opt.label_pull, pull_target, W[n_old:])
                else:
                    l, g = sr.loss1_and_grad(opt.label_pull, base_weight, W[n_old:])
                loss = f32(loss + f32(l))
                grad[n_old:] += g
            if getattr(opt, "adam", False):
                # ---- torch.optim.Adam(lr, weight_decay=0.0005) (eval/util.py:93-96; defaults betas (0.9, 0.999), eps 1e-8):
                #      L2 term added to the gradient, bias-corrected moments, scalar factors in double like Python
                g32 = grad.astype(f32) + f32(0.0005) * W
                if buf is None:
                    buf, buf2, t_adam = np.zeros_like(W), np.zeros_like(W), 0
                t_adam += 1
                buf = (buf + (g32 - buf) * f32(1.0 - 0.9)).astype(f32)
                buf2 = (buf2 * f32(0.999) + f32(1.0 - 0.999) * g32 * g32).astype(f32)
                bc1, bc2 = 1.0 - 0.9 ** t_adam, 1.0 - 0.999 ** t_adam
                denom = (np.sqrt(buf2) / f32(np.sqrt(bc2)) + f32(1e-8)).astype(f32)
                W = (W - f32(opt.learning_rate / bc1) * (buf / denom)).astype(f32)
                if bvec is not None:
                    gb32 = gb.astype(f32) + f32(0.0005) * bvec
                    if bbuf is None:
                        bbuf, bbuf2 = np.zeros_like(bvec), np.zeros_like(bvec)
                    bbuf = (bbuf + (gb32 - bbuf) * f32(1.0 - 0.9)).astype(f32)
                    bbuf2 = (bbuf2 * f32(0.999) + f32(1.0 - 0.999) * gb32 * gb32).astype(f32)
                    bden = (np.sqrt(bbuf2) / f32(np.sqrt(bc2)) + f32(1e-8)).astype(f32)
                    bvec = (bvec - f32(opt.learning_rate / bc1) * (bbuf / bden)).astype(f32)
            else:
                # ---- SGD step (torch.optim.SGD: wd added to grad, momentum buffer), :293-295
                g32 = grad.astype(f32) + f32(opt.weight_decay) * W
                buf = g32.copy() if buf is None else f32(opt.momentum) * buf + g32
                W = (W - f32(opt.learning_rate) * buf).astype(f32)
                if bvec is not None:                                             # the bias is a parameter like any other
                    gb32 = gb.astype(f32) + f32(opt.weight_decay) * bvec
                    bbuf = gb32.copy() if bbuf is None else f32(opt.momentum) * bbuf + gb32
                    bvec = (bvec - f32(opt.learning_rate) * bbuf).astype(f32)
            net.sd["classifier.weight"] = W
            if bvec is not None:
                net.sd["classifier.bias"] = bvec
            # ---- stop rule, :298-318
            lv = float(loss)
            if opt.target_train_loss == 0:
                stable = stable + 1 if abs(lv - train_loss) < opt.convergence_epsilon else 0
                if stable == opt.stable_epochs:
                    go = False
            tr_acc, _ = accuracy_top1(logits, sid)
            train_loss = lv
            losses.append(lv)
            if epoch >= opt.max_novel_epochs or (train_loss <= opt.target_train_loss
                                                 and epoch >= opt.min_novel_epochs + 1):
                go = False
            # ---- validation on every query set so far (eval mode from here on), :321-326
            net.eval()
            test_acc, test_acc5 = [], []
            for j, (xq, yq) in enumerate(zip(query_x, query_ids)):
                key = ("q", j)
                if reuse_features and key in feat_cache:
                    bump()
                else:
                    feat_cache[key] = net.features(xq)
                a, _ = accuracy_top1(linear(feat_cache[key], W, bvec), yq)
                test_acc.append(a)
                test_acc5.append(accuracy_topk(linear(feat_cache[key], W, bvec), yq, 5))   # validate's acc5, language_eval.py:40
            epoch += 1
        # ---- memory pick, :353-359
        if opt.memory_replay:
            inds = memory_indices(memory_picks[idx], opt.n_shots)
            out["memory_inds"].append(inds)
            mem_x = sx[inds] if mem_x is None else np.concatenate([mem_x, sx[inds]], 0)
            mem_y = sid[inds] if mem_y is None else np.concatenate([mem_y, sid[inds]], 0)
        # ---- base eval with the updated net, :363-367 (eval mode; BN stats may have moved)
        base_feat = net.features(base_x)
        acc_b, _ = accuracy_top1(linear(base_feat, W, bvec), base_y)
        test_acc = [round(a, 2) for a in test_acc]                               # :372
        ta = float(np.array(test_acc).mean())
        w1 = 60 if getattr(opt, "dataset", "miniImageNet") == "miniImageNet" else 200      # :383 (hard-coded class counts)
        w2 = n_old + opt.n_ways - 60                                             # :386
        if getattr(opt, "avg_weights_follow_n_base", False):                     # the build's explicit alternative (real counts)
            w1, w2 = n_base, n_old + opt.n_ways - n_base
        out["loss"].append(losses)
        out["train_acc"].append(tr_acc)
        out["test_acc"].append(test_acc)
        out["test_acc_top5"].append(test_acc5)
        out["novel_vals"].append(ta)          # AverageMeter contents, :379-380 (un-rounded)
        out["base_vals"].append(acc_b)
        out["novel_acc"].append(round(ta, 2))
        out["acc_base"].append(round(acc_b, 2))
        out["weighted_avg"].append(round((w1 * acc_b + w2 * ta) / (w1 + w2), 2))
        out["epochs"].append(epoch - 1)
    out["classifier_weight"] = W
    out["classifier_bias"] = bvec
    return out
