"""MI355X-native counterpart of eval/language_eval.py::few_shot_finetune_incremental_test (:71-454).

Same signature, same loaders, same printed result lines, same session semantics (label remap,
reserved rows, train-mode epoch 1 / eval-mode epochs >= 2, memory replay, stop rule, 2-dp
bookkeeping) - but one epoch is: ONE batched backbone launch sequence over support+memory+all
query sets (eval-mode images are independent, so batching them is results-identical), the
three-launch fused fine-tune step and one validation launch per query set, with the stop rule
evaluated on the device so that `epochs_per_sync` epochs are queued without a host round trip.
Every epoch still recomputes the frozen backbone like the reference does (no feature caching)
unless `reuse_features=True` is passed explicitly.
"""
import ctypes as C
import itertools

import numpy as np
import torch

from . import _lib
from . import functional as HF


def _drop_a_dim(data):
    """eval/util.py:131-138."""
    support_xs, support_ys, query_xs, query_ys = data
    _, _, height, width, channel = support_xs.size()
    support_xs = support_xs.view(-1, height, width, channel)
    query_xs = query_xs.view(-1, height, width, channel)
    return support_xs, support_ys.view(-1).detach().cpu().numpy(), query_xs, query_ys.view(-1).detach().cpu().numpy()


def _vocabs(base_loader, novel_loader, query_ys):
    """eval/util.py:112-129."""
    vocab_base = [n for n in base_loader.dataset.label2human if n != ""]
    novel_ids = np.sort(np.unique(query_ys))
    l2h = novel_loader.dataset.label2human
    vocab_novel = [l2h[i] for i in novel_ids]
    orig2id = dict(zip(novel_ids, len(vocab_base) + np.arange(len(novel_ids))))
    return vocab_base, vocab_novel, orig2id


def _acc(correct, n):
    """eval/util.py:26-40: float32 count * float32(100/n)."""
    return float(np.float32(correct) * np.float32(100.0 / n))


class _SessionBuffers:
    """Device buffers of one session's fused loop."""

    def __init__(self, dev, n_rows, n_cls, dim, max_epochs, n_sets):
        f32, i32 = torch.float32, torch.int32
        self.state = torch.zeros(5, dtype=i32, device=dev)            # subreg_loop_state (4 ints + 1 float)
        self.dlogits = torch.empty(n_rows * n_cls, dtype=f32, device=dev)
        self.rowloss = torch.empty(n_rows, dtype=f32, device=dev)
        self.rowcorrect = torch.empty(n_rows, dtype=i32, device=dev)
        self.norms = torch.zeros(3, dtype=f32, device=dev)
        self.rowl1 = torch.zeros(n_cls, dtype=f32, device=dev)
        self.losses = torch.zeros(max_epochs, dtype=f32, device=dev)
        self.train_acc = torch.zeros(max_epochs, dtype=f32, device=dev)
        self.correct = torch.zeros((max_epochs + 1) * n_sets, dtype=i32, device=dev)
        self.correct5 = torch.zeros((max_epochs + 1) * n_sets, dtype=i32, device=dev)   # top-5 hits (validate :40, never used by the loop)
        self.mom = torch.zeros(n_cls * dim, dtype=f32, device=dev)      # SGD momentum buffer / Adam exp_avg
        self.mom2 = torch.zeros(n_cls * dim, dtype=f32, device=dev)     # Adam exp_avg_sq (--adam, eval/util.py:93-96)
        self.bmom = torch.zeros(n_cls, dtype=f32, device=dev)           # the same two for classifier.bias (when there is one)
        self.bmom2 = torch.zeros(n_cls, dtype=f32, device=dev)


class IncrementalRunner:
    """State carried across the sessions of one run (classifier rows, BN stats, memory, reserved rows, query sets).

    start() = everything before the session loop (:100-142); run_session(idx) = the loop body (:145-395), i.e. one
    incremental EPISODE, the unit of BASELINE.json's metric; finish() = the final report (:451-454)."""

    def __init__(self, net, meta_valloader, base_val_loader, opt, base_support_loader=None, novel_inits=None,
                 memory_picks=None, epochs_per_sync=8, reuse_features=False, verbose=True, profile=False, use_graph=True,
                 ckpt=None, row_shard=None, novel_bias_inits=None):
        if getattr(opt, "track_weights", False) or getattr(opt, "save_preds_0", False):
            raise NotImplementedError("CSV tracking is outside the hot path (SURVEY.md section 8)")
        ao = getattr(opt, "attraction_override", None)
        if getattr(opt, "label_pull", None) is not None and ao not in (None, "distance2subspace", "mapping_linear_label2image"):
            raise NotImplementedError("attraction_override %r: the reference's loop knows None (semantic subspace reg), "
                                      "'distance2subspace' and 'mapping_linear_label2image'" % (ao,))
        if ao == "mapping_linear_label2image" and (ckpt is None or ao not in ckpt):
            raise ValueError("mapping_linear_label2image needs ckpt['mapping_linear_label2image'] (language_eval.py:225-226)")
        self.ckpt = ckpt
        # intra-seed data parallelism (sweep.RowShard): eval-mode forwards are split over the ranks of a group that hold the
        # same backbone, features exchanged with one all-gather per forward; None / size 1 = this rank does everything
        self.dp = row_shard if (row_shard is not None and row_shard.size > 1) else None
        # classifier with bias (eval_incremental.py:96-103: whatever the checkpoint holds): handled by the fused step; the one
        # combination the reference itself cannot run is refused where the reference fails (run_session)
        self.novel_bias_inits = novel_bias_inits
        # freeze_backbone_at = K (language_eval.py:243, eval/util.py:62-69): the backbone's parameters stop requiring grad at the
        # start of epoch K of whichever session gets there first.  Every script passes 1 (frozen before the first forward: the
        # fused loop below).  K > 1: the epochs before K fine-tune the WHOLE network (run_session::pre-freeze epochs); at K the
        # backbone freezes for good and the fused loop takes over.
        self.freeze_at = int(getattr(opt, "freeze_backbone_at", 1))
        if self.freeze_at < 1:
            raise ValueError("freeze_backbone_at must be >= 1 (the reference's loop counts epochs from 1)")
        self.backbone_frozen = self.freeze_at == 1
        self.lib = _lib.load()
        self.net, self.opt = net, opt
        self.meta_valloader, self.base_val_loader, self.base_support_loader = meta_valloader, base_val_loader, base_support_loader
        self.novel_inits, self.memory_picks = novel_inits, memory_picks
        self.epochs_per_sync, self.reuse_features = int(epochs_per_sync), reuse_features
        self.p = (lambda *a, **k: print(*a, **k)) if verbose else (lambda *a, **k: None)
        self.profile = profile
        self.use_graph = use_graph    # replay the per-epoch backbone forward as a hipGraph (one launch instead of ~30)
        self.fwd_events = []          # (start, end, n_images) of every backbone forward when profile=True
        self.images_forwarded = 0

    # ------------------------------------------------------------------ helpers
    def _forward(self, x, train=False, out=None, check_params=True):
        net = self.net
        if self.profile:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        r = self.hb.forward(x, train=train, masks=net.mask_source if train else None, out=out, check_params=check_params)
        if self.profile:
            e1.record()
            self.fwd_events.append((e0, e1, x.shape[0]))
        self.images_forwarded += x.shape[0]
        return r

    def _forward_eval(self, x, out=None, gather_buf=None, local_buf=None, forward=None):
        """Eval-mode forward of x [B,...] -> features [B, D]: locally, or this rank's row slice + all-gather.
        `forward(x_slice, out_rows)` overrides the launch (the hipGraph replay of an identical earlier call)."""
        if self.dp is None:
            if forward is not None:
                forward(x, out)
                return out
            return self._forward(x, out=out, check_params=False)
        B = x.shape[0]
        lo, hi, per = self.dp.rows(B)
        if local_buf is None:
            local_buf = torch.zeros(per, self.D, dtype=torch.float32, device=self.dev)
        if hi > lo:
            if forward is not None:
                forward(x[lo:hi], local_buf[:hi - lo])
            else:
                self._forward(x[lo:hi], out=local_buf[:hi - lo], check_params=False)
        full = self.dp.gather(local_buf, B, out=gather_buf)
        if out is not None and out.data_ptr() != full.data_ptr():
            out.copy_(full)
            return out
        return full

    def _eval_base(self):                                                        # :46-69
        self.net.eval()
        self.hb.refresh()
        feat = self._forward_eval(self.base_x)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.dev)
        W = self.net.classifier.weight.detach()
        bias = self.net.classifier.bias
        _lib.check(self.lib.subreg_validate(_lib.ptr(feat), _lib.ptr(self.base_y), _lib.ptr(W),
                                            _lib.ptr(bias.detach()) if bias is not None else None, feat.shape[0], W.shape[0],
                                            self.D, None, _lib.ptr(cnt), 0, 1, 0, _lib.stream_ptr()), "validate(base)")
        return _acc(int(cnt.item()), feat.shape[0])

    # ------------------------------------------------------------------ :100-142
    def start(self):
        net, opt = self.net, self.opt
        self.dev = net.classifier.weight.device
        torch.manual_seed(opt.set_seed)                                          # :101-102
        np.random.seed(opt.set_seed)
        self.hb = net.hip_backbone()
        self.D = net.classifier.weight.shape[1]
        self.base_weight = net.classifier.weight.detach().clone()                # :106-107
        self.n_base = self.base_weight.shape[0]
        self.base_bias = net.classifier.bias.detach().clone() if net.classifier.bias is not None else None
        # regularizer of the novel rows (:216-227, :277-290): projection onto span(W_base) ('distance2subspace', basis of the
        # constant W_base computed once) or a per-session constant target from the word embeddings (LangPuller.forward)
        self.pull_mode = None
        if opt.label_pull is not None and getattr(opt, "pulling", "regularize") == "regularize":
            self.pull_mode = "subspace" if getattr(opt, "attraction_override", None) == "distance2subspace" else "target"
        self.basis, self.basis_info = (HF.subspace_basis(self.base_weight) if self.pull_mode == "subspace" else (None, None))
        self.lang_puller, self.pullers = None, None
        self.meta_it = itertools.cycle(iter(self.meta_valloader))
        self.base_support_xs = self.base_support_ys = None
        if self.base_support_loader is not None:                                 # :112-116
            self.base_support_xs, self.base_support_ys, *_ = _drop_a_dim(next(itertools.cycle(iter(self.base_support_loader))))
        base_batch = next(itertools.cycle(iter(self.base_val_loader)))           # :121
        self.base_x = base_batch[0].squeeze(0).to(self.dev, torch.float32)
        self.base_y = base_batch[1].squeeze(0).to(self.dev, torch.int64)
        self.acc_novel_list, self.acc_base_list, self.weighted_avg_l = [], [], []
        self.novel_vals, self.base_vals = [], []                                 # AverageMeter contents (:379-380)
        self.weighted_avg_l.append(self._eval_base())                            # :128-129
        self.iter_num = 8 if getattr(opt, "continual", False) else opt.neval_episodes   # :132-136
        self.query_x, self.query_id = [], []
        self.mem_x = self.mem_y = None
        self.reserve = None
        self.run = dict(loss=[], test_acc=[], test_acc_top5=[], epochs=[], train_acc=[], memory_inds=[], graph_replays=[])
        self.vocab_base = self.vocab_novel = None
        return self

    # ------------------------------------------------------------------ :145-395, one episode
    def run_session(self, idx):
        net, opt, lib, dev, D, p = self.net, self.opt, self.lib, self.dev, self.D, self.p
        hb, s = self.hb, _lib.stream_ptr
        p("\n**** Iteration {}/{} ****\n".format(idx + 1, opt.neval_episodes))
        support_xs, support_ys, qx, qy = _drop_a_dim(next(self.meta_it))
        if self.base_support_xs is not None:
            support_xs = torch.cat([support_xs.to(self.base_support_xs.device), self.base_support_xs], 0)   # :149-150
        prev_vb, prev_vn = self.vocab_base, self.vocab_novel
        vocab_base, vocab_novel, orig2id = _vocabs(self.base_val_loader, self.meta_valloader, qy)
        if idx > 0:
            vocab_base = prev_vb + prev_vn                                     # :166-167
        self.vocab_base, self.vocab_novel = vocab_base, vocab_novel
        n_old = len(vocab_base)
        if self.pull_mode == "target":                                         # :218-227
            from .resnet_language import LangPuller
            if idx == 0:
                self.lang_puller = LangPuller(opt, vocab_base, vocab_novel)
            else:
                self.lang_puller.update_novel_embeds(vocab_novel)
            if getattr(opt, "attraction_override", None) == "mapping_linear_label2image":
                self.lang_puller.create_pulling_mapping(self.ckpt[opt.attraction_override])
            with torch.no_grad():
                self.pullers = self.lang_puller(self.base_weight[:self.n_base]).contiguous()
        W = net.classifier.weight.detach()
        if idx == 1:                                                           # :172-185
            self.reserve = W[-opt.n_ways:].clone()
        elif idx > 1:
            self.reserve = torch.cat((self.reserve, W[-opt.n_ways:].clone()), 0)
        reserve = self.reserve
        novel_labels = np.sort(np.unique(qy))
        orig2id = {k: v + idx * opt.n_ways for k, v in orig2id.items()}        # :193-194
        qid = torch.tensor([orig2id[y] for y in qy], dtype=torch.int64)
        sid = torch.tensor([orig2id[y] for y in support_ys], dtype=torch.int64)
        if self.base_support_ys is not None:
            sid = torch.cat([sid, torch.from_numpy(np.asarray(self.base_support_ys)).long()])   # :207-209
        self.query_x.append(qx.to(dev, torch.float32))
        self.query_id.append(qid.to(dev))
        query_x, query_id = self.query_x, self.query_id
        net.train()                                                            # :211
        net.augment_base_classifier_(len(novel_labels), novel_weight=None if self.novel_inits is None
                                     else torch.as_tensor(self.novel_inits[idx]),
                                     novel_bias=None if (self.novel_bias_inits is None or self.base_bias is None)
                                     else torch.as_tensor(self.novel_bias_inits[idx]))
        W = net.classifier.weight.data                                         # live [N, D], updated in place by the step
        bias = net.classifier.bias.data if net.classifier.bias is not None else None   # live [N]
        N = W.shape[0]
        sx = support_xs.to(dev, torch.float32)
        sid = sid.to(dev)
        mem_x, mem_y = self.mem_x, self.mem_y
        Bs, Bm = sx.shape[0], (0 if mem_x is None else mem_x.shape[0])
        n_sets = len(query_x)
        max_e = int(opt.max_novel_epochs)
        ses = _SessionBuffers(dev, Bs + Bm, N, D, max_e, n_sets)
        _lib.check(lib.subreg_loop_state_init(_lib.ptr(ses.state), s()), "loop_state_init")
        labels = sid if mem_x is None else torch.cat([sid, mem_y])
        all_x = torch.cat([sx] + ([mem_x] if mem_x is not None else []) + query_x, 0)
        if self.dp is None:
            feats = torch.empty(all_x.shape[0], D, dtype=torch.float32, device=dev)
            gather_buf = local_buf = None
        else:                                  # feats = the first rows of the all-gather's output buffer (no copy per epoch)
            lo_all, hi_all, per_all = self.dp.rows(all_x.shape[0])
            gather_buf = torch.zeros(self.dp.size * per_all, D, dtype=torch.float32, device=dev)
            local_buf = torch.zeros(per_all, D, dtype=torch.float32, device=dev)
            feats = gather_buf[:all_x.shape[0]]
        q_off = [Bs + Bm + sum(q.shape[0] for q in query_x[:j]) for j in range(n_sets)]
        query_labels = torch.cat(query_id, 0).contiguous()
        set_rows = (C.c_int * n_sets)(*[int(q.shape[0]) for q in query_x])
        d = _lib.StepDesc()
        d.feat, d.labels = feats.data_ptr(), labels.data_ptr()
        d.n_support, d.n_memory, d.n_classes, d.dim = Bs, Bm, N, D
        d.weight, d.momentum_buf = W.data_ptr(), ses.mom.data_ptr()
        d.w_base = self.base_weight.data_ptr()
        use_prev = opt.lmbd_reg_novel is not None and idx > 0
        if use_prev and bias is not None:
            # resnet_language.py:238 `self.classifier.bias[rng1:rng2, :]` on the 1-D bias: the reference's first epoch of
            # session 2 dies here whenever the classifier has a bias and --lmbd_reg_novel is given
            raise IndexError("too many indices for tensor of dimension 1")
        d.w_prev = reserve.data_ptr() if use_prev else None
        d.basis = self.basis.data_ptr() if self.basis is not None else None
        d.pull_target = self.pullers.data_ptr() if self.pull_mode == "target" else None
        d.n_base, d.n_prev, d.n_old = self.n_base, (reserve.shape[0] if use_prev else 0), n_old
        d.lr, d.momentum, d.weight_decay = opt.learning_rate, opt.momentum, opt.weight_decay
        if getattr(opt, "adam", False):                 # get_optim (eval/util.py:92-97): Adam(lr, weight_decay=0.0005), torch defaults
            d.adam, d.beta1, d.beta2, d.adam_eps, d.weight_decay = 1, 0.9, 0.999, 1e-8, 0.0005
            d.exp_avg_sq = ses.mom2.data_ptr()
        else:
            d.adam, d.exp_avg_sq = 0, None
        if bias is not None:
            d.bias, d.bias_momentum_buf, d.bias_base = bias.data_ptr(), ses.bmom.data_ptr(), self.base_bias.data_ptr()
            d.bias_exp_avg_sq = ses.bmom2.data_ptr() if d.adam else None
        bias_p = _lib.ptr(bias) if bias is not None else None
        d.lmbd_base = opt.lmbd_reg_transform_w or 0.0
        d.lmbd_prev = opt.lmbd_reg_novel or 0.0
        d.pull = opt.label_pull or 0.0
        d.use_base_reg = int(opt.lmbd_reg_transform_w is not None)
        d.use_prev_reg, d.use_pull = int(use_prev), int(self.pull_mode is not None)
        d.dlogits, d.rowloss, d.rowcorrect = ses.dlogits.data_ptr(), ses.rowloss.data_ptr(), ses.rowcorrect.data_ptr()
        d.norms, d.rowl1, d.state = ses.norms.data_ptr(), ses.rowl1.data_ptr(), ses.state.data_ptr()
        d.losses, d.train_acc = ses.losses.data_ptr(), ses.train_acc.data_ptr()
        opt.stable = True if opt.target_train_loss == 0 else False            # :238
        d.max_epochs, d.min_epochs, d.stable_epochs = max_e, int(opt.min_novel_epochs), int(opt.stable_epochs)
        d.stable_mode, d.target_loss, d.convergence_eps = int(opt.stable), opt.target_train_loss, opt.convergence_epsilon
        fwd_per_epoch = 1 + (1 if Bm else 0) + n_sets     # forwards the reference makes per epoch (feeds DropBlock's gamma)

        def validate_only():
            if n_sets <= _lib.MAX_QUERY_SETS:          # all query sets in one launch (rows and labels are consecutive)
                _lib.check(lib.subreg_validate_sets(_lib.ptr(feats[q_off[0]:]), _lib.ptr(query_labels), _lib.ptr(W), bias_p, set_rows,
                                                    n_sets, N, D, _lib.ptr(ses.state), _lib.ptr(ses.correct), _lib.ptr(ses.correct5),
                                                    n_sets, 1, s()), "validate_sets")
                return
            for j in range(n_sets):                    # more sessions than one launch takes: one launch per set
                _lib.check(lib.subreg_validate(_lib.ptr(feats[q_off[j]:]), _lib.ptr(query_id[j]), _lib.ptr(W), bias_p,
                                               query_x[j].shape[0], N, D, _lib.ptr(ses.state), _lib.ptr(ses.correct), j,
                                               n_sets, int(j == n_sets - 1), s()), "validate")

        def step_and_validate():
            _lib.check(lib.subreg_finetune_step(C.byref(d), s()), "finetune_step")
            validate_only()

        trainable = [] if self.backbone_frozen else [p_ for n_, p_ in net.named_parameters()
                                                       if not n_.startswith("classifier") and p_.requires_grad]
        if trainable and self.freeze_at > 1:
            # ---- pre-freeze epochs 1 .. K-1 (language_eval.py:242-295 with a trainable backbone): the support forward keeps a stash
            #      (train mode in epoch 1, eval mode afterwards: validate() leaves the model there, :19); the fused step computes the
            #      loss, d(logits) and the classifier's own SGD update as in every other epoch; d(features) = d(logits) @ W (the W
            #      the forward used) goes back through the backbone's HIP backward, and every backbone parameter takes the same
            #      SGD(lr, momentum, weight_decay) step (get_optim builds ONE optimiser over net.parameters() per session, :231);
            #      the query sets are then forwarded through the UPDATED backbone.  The regularizers do not reach the backbone.
            #      Semantics AFTER the freeze are those of torch >= 2.0 (the torch the goldens were generated with, 2.10):
            #      optimizer.zero_grad() sets gradients to None, so a frozen parameter (requires_grad False, grad None) is skipped by
            #      SGD - no weight decay, no coasting momentum.  Under the reference's documented environment (setup.sh: torch 1.7,
            #      zero_grad(set_to_none=False)) the frozen parameters keep ZERO gradient tensors and the session's single SGD
            #      optimiser goes on applying weight decay and momentum to the backbone after epoch K; that legacy behaviour is
            #      NOT emulated here (it is an artefact of the old default, not something language_eval.py asks for), and the
            #      golden loop_hw32_freeze3.npz does not pin it.
            #      --adam: get_optim's Adam(lr, weight_decay=0.0005) (eval/util.py:92-97) - element-wise, so the backbone's own
            #      Adam beside the fused step's Adam on the classifier is the reference's single optimiser.
            #      Replay memory (sessions after one that ended before the freeze): the reference forwards the memory batch a second
            #      time through the network (:252-258) - two gradient-carrying forwards (train mode in epoch 1: two batch-statistics
            #      passes, two running-stat updates), each with its own stash; their backward passes accumulate.
            from .train import SGD as _BackboneSGD, Adam as _BackboneAdam
            if getattr(opt, "adam", False):
                opt_bb = _BackboneAdam(trainable, lr=opt.learning_rate, weight_decay=0.0005)
            else:
                opt_bb = _BackboneSGD(trainable, lr=opt.learning_rate, momentum=opt.momentum, weight_decay=opt.weight_decay)
            dfeat = torch.empty(Bs + Bm, D, dtype=torch.float32, device=dev)
            while True:
                st = ses.state.cpu()
                epoch_next = int(st[0]) + 1
                if bool(st[1]) or epoch_next >= self.freeze_at:
                    break
                with torch.enable_grad():
                    feat = net.features(sx)                                    # BackboneTrainFn: train mode in epoch 1, eval after
                    feat_m = net.features(mem_x) if Bm else None
                feats[:Bs].copy_(feat.detach())
                if Bm:
                    feats[Bs:Bs + Bm].copy_(feat_m.detach())
                self.images_forwarded += Bs + Bm
                w_used = W.clone()
                _lib.check(lib.subreg_finetune_step(C.byref(d), s()), "finetune_step")
                _lib.check(lib.subreg_linear_bwd(_lib.ptr(ses.dlogits), _lib.ptr(feats), _lib.ptr(w_used), None, None, _lib.ptr(dfeat),
                                                 Bs + Bm, N, D, s()), "linear_bwd")
                if Bm:
                    import warnings
                    with warnings.catch_warnings():     # (the second backward accumulates into the first one's gradients: by design here)
                        warnings.filterwarnings("ignore", message="subreg_hip: backward.. found existing .grad", category=RuntimeWarning)
                        torch.autograd.backward([feat, feat_m], [dfeat[:Bs], dfeat[Bs:]])
                else:
                    feat.backward(dfeat)
                opt_bb.step()
                opt_bb.zero_grad()
                net.eval()                                                     # validate() flips the mode for good, :19
                if self.dp is not None:
                    # a row-sharded seed: every rank of the group made this whole-network step itself, from the same inputs (the
                    # support set is not sharded).  The weight-gradient kernels sum with float atomics, so two ranks may differ in
                    # the last bit - and must not, they take the stop decisions together: the leader's network is THE network
                    # (one flat-buffer broadcast per pre-freeze epoch; the query forwards below are sharded again)
                    from . import sweep as _sweep
                    _sweep.broadcast_module(net, self.dp.leader, group=self.dp.group)
                hb.refresh(force=True)                                         # weights (and in epoch 1 the running statistics) moved
                self._forward_eval(all_x[Bs + Bm:], out=feats[Bs + Bm:])
                for i in range(len(hb.nbt)):
                    hb.nbt[i] += n_sets - 1
                validate_only()
            st = ses.state.cpu()
            if not bool(st[1]):                                                # the loop reached epoch K: freeze (eval/util.py:62-69)
                for n_, p_ in net.named_parameters():
                    p_.requires_grad = n_.startswith("classifier")
                self.backbone_frozen = True
                self.p("Freezing the backbone.")
                hb.refresh(force=True)
        else:
            # ---- epoch 1: TRAIN-mode support (+memory) forward (BN batch stats, running-stat update, masks), :252-258
            self._forward(sx, train=True, out=feats[:Bs])
            if Bm:
                self._forward(mem_x, train=True, out=feats[Bs:Bs + Bm])
            torch._foreach_add_([m.num_batches_tracked for m in net._bns], 1 + (1 if Bm else 0))     # one launch, not one per BatchNorm
            net.eval()                                                             # validate() flips the mode for good, :19
            hb.refresh()                                                           # BN running statistics moved: fold once
            self._forward_eval(all_x[Bs + Bm:], out=feats[Bs + Bm:])
            for i in range(len(hb.nbt)):
                hb.nbt[i] += n_sets - 1
            step_and_validate()
        # ---- epochs >= 2: eval mode, one batched forward per epoch.  The forward is identical every epoch (frozen
        #      backbone, constant inputs): after one eager pass its launch sequence is captured into a hipGraph and
        #      replayed - every epoch still executes all 22 convolutions, only the host-side launches are saved.
        graph, eager_done, replays = None, 0, 0
        need_forward = True
        # what THIS rank forwards per epoch: everything, or its row slice of all_x (features all-gathered afterwards)
        if self.dp is None:
            x_loc, out_loc = all_x, feats
        else:
            x_loc, out_loc = all_x[lo_all:hi_all], local_buf[:hi_all - lo_all]
        while True:
            st = ses.state.cpu()
            done, stop = int(st[0]), bool(st[1])
            if stop:
                break
            k = min(self.epochs_per_sync, max_e - done)
            executed = 0
            for e in range(k):
                # reuse_features: ONE eval-mode forward of every row through the backbone as it is from now on (frozen) - the first
                # pass through here, whether epoch 1 or the pre-freeze epochs 1 .. K-1 came before - then the features are reused
                if not self.reuse_features or need_forward:
                    need_forward = False
                    if x_loc.shape[0] == 0:
                        pass                                                    # more ranks than rows: nothing to forward here
                    elif graph is not None:
                        if self.profile:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                        graph.replay()
                        replays += 1
                        if self.profile:
                            e1.record()
                            self.fwd_events.append((e0, e1, x_loc.shape[0]))
                        self.images_forwarded += x_loc.shape[0]
                        for i in range(len(hb.nbt)):
                            hb.nbt[i] += 1
                    else:
                        self._forward(x_loc, out=out_loc, check_params=False)   # frozen backbone, nothing changed
                        eager_done += 1
                        if self.use_graph and not self.reuse_features and eager_done == 1 and max_e - done > 4:
                            nbt_keep = list(hb.nbt)
                            torch.cuda.synchronize()
                            graph = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(graph):
                                hb.forward(x_loc, out=out_loc, check_params=False)
                            hb.nbt = nbt_keep                                   # the capture pass launches nothing
                    if self.dp is not None:
                        self.dp.gather(local_buf, all_x.shape[0], out=gather_buf)   # feats = gather_buf[:rows]
                    executed += 1
                step_and_validate()
            ran = int(ses.state.cpu()[0]) - done   # epochs that really advanced the loop
            for i in range(len(hb.nbt)):            # count forwards like the reference (queued-but-stopped ones do not exist there)
                hb.nbt[i] += ran * fwd_per_epoch - executed
        epochs = done
        if graph is not None:
            # the captured forward points at this session's buffers and at the lanes' side streams: destroy it explicitly, on an
            # idle device, BEFORE those buffers can be released (not whenever the garbage collector gets to it)
            torch.cuda.synchronize()
            graph.reset()
            del graph
        losses = ses.losses[:epochs].cpu().numpy().astype(np.float64)
        correct = ses.correct.view(-1, n_sets)[epochs].cpu().numpy()
        test_acc = [round(_acc(int(c), query_x[j].shape[0]), 2) for j, c in enumerate(correct)]   # :372
        top5 = None
        if n_sets <= _lib.MAX_QUERY_SETS:           # validate's acc5 (:40); the reference computes it and uses it nowhere
            top5 = [_acc(int(c), query_x[j].shape[0]) for j, c in enumerate(ses.correct5.view(-1, n_sets)[epochs].cpu().numpy())]
        if opt.memory_replay:                                                  # :353-359
            pick = (self.memory_picks[idx] if self.memory_picks is not None
                    else np.random.choice(opt.n_shots, opt.memory_replay))
            inds = np.tile(5 * np.arange(5) + np.asarray(pick), (5, 1)) + (np.tile(np.arange(0, 125, 25), (5, 1))).T
            inds = torch.from_numpy(inds.flatten()).to(dev)
            self.run["memory_inds"].append(inds.cpu().numpy())
            self.mem_x = sx[inds] if mem_x is None else torch.cat((mem_x, sx[inds]), 0)
            self.mem_y = sid[inds] if mem_y is None else torch.cat((mem_y, sid[inds]), 0)
        acc_base_ = self._eval_base()                                          # :363-367
        p("Novel session accuracies: ", test_acc)
        ta = float(np.array(test_acc).mean())
        # :383-386.  The reference hard-codes the class counts (w1 = 60 for miniImageNet, else 200 "tiered"; w2 subtracts
        # 60) although its own tieredImageNet pretraining uses 351 base classes (train_supervised.py:94).  Default = the
        # reference's numbers; opt.avg_weights_follow_n_base=True (this build's flag, absent from configs.py) weights by the
        # real counts: w1 = n_base, w2 = novel classes seen so far.
        if getattr(opt, "avg_weights_follow_n_base", False):
            w1 = self.n_base
            w2 = len(vocab_base) + len(vocab_novel) - self.n_base
        else:
            w1 = 60 if opt.dataset == "miniImageNet" else 200
            w2 = len(vocab_base) + len(vocab_novel) - 60
        weighted_avg = (w1 * acc_base_ + w2 * ta) / (w1 + w2)
        self.weighted_avg_l.append(round(weighted_avg, 2))
        self.acc_novel_list.append(round(ta, 2))
        self.acc_base_list.append(round(acc_base_, 2))
        self.novel_vals.append(ta)
        self.base_vals.append(acc_base_)
        p("***Running weighted avg: {}".format(weighted_avg))
        run = self.run
        run["loss"].append(losses)
        run["test_acc"].append(test_acc)
        run["epochs"].append(epochs)
        run["graph_replays"].append(replays)
        run["test_acc_top5"].append(top5)
        run["train_acc"].append(ses.train_acc[:epochs].cpu().numpy())
        return epochs

    # ------------------------------------------------------------------ :451-454
    def finish(self):
        run, p = self.run, self.p
        run.update(weighted_avg=self.weighted_avg_l, novel_acc=self.acc_novel_list, acc_base=self.acc_base_list,
                   classifier_weight=self.net.classifier.weight.detach().cpu().numpy(), basis_info=self.basis_info,
                   classifier_bias=(self.net.classifier.bias.detach().cpu().numpy() if self.net.classifier.bias is not None else None))
        self.net.last_run = run
        p("Overall continual accuracies: ", self.weighted_avg_l)
        p("Novel only incremental: ", self.acc_novel_list)
        p("Base only incremental: ", self.acc_base_list)
        return float(np.mean(self.novel_vals)), float(np.mean(self.base_vals))


def few_shot_finetune_incremental_test(net, ckpt, criterion, meta_valloader, base_val_loader, opt, vis=False,
                                       base_support_loader=None, *, novel_inits=None, memory_picks=None,
                                       epochs_per_sync=8, reuse_features=False, verbose=True, novel_bias_inits=None):
    """Drop-in for the reference function.  Extra keyword-only arguments (all optional):
      novel_inits     list of [n_ways, 640] init rows passed to augment_base_classifier_(novel_weight=...)
      novel_bias_inits  list of [n_ways] init values passed as novel_bias=... (classifier with bias only)
      memory_picks    list of np.random.choice(n_shots, memory_replay) results (else drawn from np.random)
      epochs_per_sync epochs queued per host synchronisation
      reuse_features  opt-in: compute the (constant) eval-mode features once per session
    `criterion` is accepted for signature compatibility (CrossEntropyLoss is fused into the step); `ckpt` is read only
    for ckpt['mapping_linear_label2image'] (the LinearMap state_dict, language_eval.py:225-226).  Returns (acc_novel.avg, acc_base.avg); details in net.last_run."""
    if vis:
        raise NotImplementedError("visualisation is outside the hot path (SURVEY.md section 8)")
    r = IncrementalRunner(net, meta_valloader, base_val_loader, opt, base_support_loader, novel_inits, memory_picks,
                          epochs_per_sync, reuse_features, verbose, ckpt=ckpt, novel_bias_inits=novel_bias_inits).start()
    for idx in range(r.iter_num):
        r.run_session(idx)
    return r.finish()
