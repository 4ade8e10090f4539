"""Drop-in module surface of /root/reference/models/resnet_language.py for the incremental path.

Same class / method names, constructor arguments, state_dict keys (133 for resnet18 without
classifier bias) and train()/eval() semantics as the reference, but `forward` runs the
hand-written gfx950 kernels (one C call for the whole backbone) instead of cuDNN/ATen.
The nn.Conv2d / nn.BatchNorm2d children are PARAMETER CONTAINERS only (names, init,
.cuda(), state_dict); they are never called.

Not mirrored (out of the hot path): SELayer, the imagenet-style resnet50+,
the lang-linear classifiers.
"""
import os
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import functional as HF
from .backbone import HipBackbone


class LinearMap(nn.Module):
    """resnet_language.py:12-18 (alternative regularizer target; parameter container + torch linear)."""

    def __init__(self, indim, outdim):
        super().__init__()
        self.map = nn.Linear(indim, outdim)

    def forward(self, x):
        return HF.LinearFn.apply(x, self.map.weight, self.map.bias)


def _get_embeds(embed_pth, vocab, dim=500):
    """models/util.py:50-66: mean of per-word vectors; unknown word -> zeros."""
    with open(embed_pth, "rb") as f:
        table = pickle.load(f)
    embeds = [0] * len(vocab)
    for i, token in enumerate(vocab):
        words = token.split(" ")
        for w in words:
            try:
                embeds[i] += table[w]
            except KeyError:
                embeds[i] = np.zeros(dim)
        embeds[i] /= len(words)
    return torch.stack([torch.from_numpy(np.asarray(e)) for e in embeds], 0)


class LangPuller(nn.Module):
    """resnet_language.py:20-97.  `distance2subspace` (get_projected_weight + loss1) is the hot-path
    variant and runs on the fused HIP kernels with the basis of the CONSTANT base matrix cached; the
    semantic variant (forward) needs the word-embedding pickles and is loaded lazily."""

    def __init__(self, opt, vocab_base, vocab_novel):
        super().__init__()
        if getattr(opt, "use_synonyms", False):
            # resnet_language.py:35-45: the synonym branch reads <dataset>_dim<dim>_base_synonyms.pickle (not shipped with the
            # reference) into a Python LIST of per-label entries and then calls `.float()` on that list (:45) - it cannot get
            # past __init__ in the reference.  Same failures here, in the same order: the missing file, a missing label, then
            # the list that has no .float().
            pth = os.path.join(opt.word_embed_path, "{0}_dim{1}_base_synonyms.pickle".format(opt.dataset, opt.word_embed_size))
            with open(pth, "rb") as f:
                label_syn_embeds = pickle.load(f)
            base_embeds = [label_syn_embeds[base_label] for base_label in vocab_base]
            raise AttributeError("'%s' object has no attribute 'float'" % type(base_embeds).__name__)
        self.mapping_model = None
        self.opt = opt
        self.vocab_base = vocab_base
        self.vocab_novel = vocab_novel
        self.temp = getattr(opt, "temperature", 1)
        self._embeds_loaded = False
        self._basis_key, self._basis = None, None

    # -- semantic variant: embeddings are only touched when it is actually used
    def _embed_path(self):
        opt = self.opt
        return os.path.join(opt.word_embed_path, "{0}_dim{1}.pickle".format(opt.dataset, opt.word_embed_size))

    def _load_embeds(self):
        if self._embeds_loaded:
            return
        opt = self.opt
        self.novel_embeds = _get_embeds(self._embed_path(), self.vocab_novel).float().cuda()
        self.base_embeds = _get_embeds(self._embed_path(), self.vocab_base).float().cuda()
        if getattr(opt, "glove", False):
            self.base_embeds, self.novel_embeds = self.base_embeds[:, :300], self.novel_embeds[:, :300]
        self._embeds_loaded = True

    def update_novel_embeds(self, vocab_novel):
        self.vocab_novel = vocab_novel
        if self._embeds_loaded:
            self.novel_embeds = _get_embeds(self._embed_path(), vocab_novel).float().cuda()
            if getattr(self.opt, "glove", False):
                self.novel_embeds = self.novel_embeds[:, :300]

    def create_pulling_mapping(self, state_dict, base_weight_size=640):
        self._load_embeds()
        self.mapping_model = LinearMap(self.novel_embeds.size(1), base_weight_size)
        self.mapping_model.load_state_dict(state_dict)
        self.mapping_model = self.mapping_model.cuda()

    def forward(self, base_weight, mask=False):
        """Semantic subspace regularizer target (:75-87): one fused kernel (scores, softmax, @ W_base), or the linear
        mapping of the novel embeddings (HIP linear kernel) when create_pulling_mapping was called."""
        self._load_embeds()
        if self.mapping_model is None:
            return HF.SemanticTargetFn.apply(base_weight, self.novel_embeds, self.base_embeds, self.temp, mask)
        with torch.no_grad():
            return self.mapping_model(self.novel_embeds)

    # -- hot-path variant
    def basis(self, base_weight):
        key = (base_weight.data_ptr(), base_weight._version, tuple(base_weight.shape))
        if key != self._basis_key:
            self._basis, self._basis_info = HF.subspace_basis(base_weight)
            self._basis_key = key
        return self._basis

    def get_projected_weight(self, base_weight, weights):
        return HF.SubspaceProjectFn.apply(weights, self.basis(base_weight))

    def loss1(self, pull, inspired, weights):
        return HF.SqDiffFn.apply(inspired, weights, pull)


class _Block(nn.Module):
    """Parameter container with BasicBlock's child names (resnet_language.py:243-266)."""

    def __init__(self, inplanes, planes, stride, downsample, drop_block, block_size):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride, self.drop_block, self.block_size = stride, drop_block, block_size

    def forward(self, x):
        raise RuntimeError("BasicBlock is a parameter container; the fused HIP backbone runs the whole stack "
                           "(call the ResNet module)")


class ResNet(nn.Module):
    """resnet_language.py:101-240 with the same constructor and method surface."""

    def __init__(self, block, n_blocks, keep_prob=1.0, avg_pool=False, drop_rate=0.0, dropblock_size=5,
                 num_classes=-1, use_se=False, vocab=None, opt=None):
        if vocab is not None:
            assert opt is not None
        super().__init__()
        if use_se:
            raise NotImplementedError("SE blocks are not part of the incremental hot path")
        if vocab is not None:
            raise NotImplementedError("lang-linear classifiers are not part of the incremental hot path")
        if not avg_pool or drop_rate != 0.1:
            raise NotImplementedError("the HIP backbone implements create_model's configuration "
                                      "(avg_pool=True, drop_rate=0.1; models/util.py:15-18)")
        self.n_blocks = tuple(n_blocks)
        self.block_size = 1 if getattr(opt, "no_dropblock", False) else dropblock_size     # :116-118
        widths, inplanes = (64, 160, 320, 640), 3
        for si, (nb, planes) in enumerate(zip(self.n_blocks, widths)):
            layers = []
            for bi in range(nb):
                first = bi == 0
                down = None
                if first:
                    down = nn.Sequential(nn.Conv2d(inplanes, planes, 1, 1, bias=False), nn.BatchNorm2d(planes))
                db = si >= 2 and (nb == 1 or (bi == nb - 1 and not first))
                layers.append(_Block(inplanes if first else planes, planes, 2 if first else 1, down, db, self.block_size))
            inplanes = planes
            setattr(self, "layer%d" % (si + 1), nn.Sequential(*layers))
        self.keep_prob, self.keep_avg_pool, self.drop_rate, self.vocab = keep_prob, avg_pool, drop_rate, vocab
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="leaky_relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.num_classes = num_classes
        if self.num_classes > 0:
            self.classifier = nn.Linear(640, self.num_classes, bias=getattr(opt, "linear_bias", True))
        self.hip_dtype = getattr(opt, "hip_dtype", None) or os.environ.get("SUBREG_DTYPE", "bf16")
        self.mask_source = None      # optional injected masks for train-mode forwards (parity tests)
        self._hip = None

    # ------------------------------------------------------------------ HIP backbone plumbing
    def _apply(self, fn, *a, **k):
        self._hip = None             # .cuda()/.to() replace the tensors the backbone points at
        return super()._apply(fn, *a, **k)

    def hip_backbone(self):
        if self._hip is None or self._hip.dtype_name != self.hip_dtype:
            params = {}
            for k, v in list(self.named_parameters()) + list(self.named_buffers()):
                if not k.startswith("classifier"):
                    params[k] = v
            self._hip = HipBackbone(params, self.n_blocks, self.hip_dtype, self.block_size)
            self._hip.dtype_name = self.hip_dtype
            self._bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        return self._hip

    def features(self, x, return_stages=False):
        if not x.is_cuda:
            raise RuntimeError("subreg_hip.ResNet runs on the MI355X only (no CPU fallback); move the input to cuda")
        hb = self.hip_backbone()
        if torch.is_grad_enabled() and any(p.requires_grad for n, p in self.named_parameters()
                                           if not n.startswith("classifier")):
            # pretraining path (train_supervised.py:229-244) and whole-network fine-tuning before freeze_backbone_at
            # (eval/language_eval.py:242-295): forward with a stash, backward on the HIP kernels.  In eval mode (the fine-tuning
            # loop after its first validate()) BatchNorm uses and keeps its running statistics and there is no dropout
            from .train import BackboneTrainFn
            names, params = zip(*[(n, p) for n, p in self.named_parameters() if not n.startswith("classifier")])
            feat = BackboneTrainFn.apply(x, hb, self.mask_source if self.training else "eval", names, *params)
            if self.training:
                torch._foreach_add_([m.num_batches_tracked for m in self._bns], 1)      # one launch, not one per BatchNorm
            if return_stages:
                # is_feat=True (models/resnet_language.py:170-192): the block outputs are in the stash (each is the next block's
                # input for the backward).  They are handed out as COPIES without a grad_fn: the backward of this path starts at
                # `feat`; a loss on the stage features themselves (the reference's distill/ code, out of scope) gets no gradient
                if not getattr(self, "_warned_stage_grad", False):
                    import warnings
                    self._warned_stage_grad = True
                    warnings.warn("subreg_hip: is_feat=True with a grad-requiring backbone returns the stage features detached "
                                  "(gradients flow through the pooled feature / the logits only)", RuntimeWarning, stacklevel=3)
                B, _, h, w = x.shape
                stages, stash = [], hb._train_stash
                for bi, (_n, _ci, cout, stride, _ds, _db) in enumerate(hb.blocks):
                    h, w = h // stride, w // stride
                    stages.append(stash.named[(bi, "out")][:B * h * w * cout].view(B, h, w, cout).permute(0, 3, 1, 2).float().contiguous())
                return feat, stages
            return feat
        hb = self.hip_backbone()
        if not self.training and not return_stages:
            # eval mode over a backbone without gradients: the unchanged reference loop calls this with the same shapes every epoch
            # (support set, every query set) - cached hipGraph per shape, see HipBackbone.forward_graphed
            return hb.forward_graphed(x.float())
        out = hb.forward(x.float(), train=self.training, masks=self.mask_source, return_stages=return_stages)
        if self.training:
            torch._foreach_add_([m.num_batches_tracked for m in self._bns], 1)
        return out

    def forward(self, x, is_feat=False, get_alphas=False):
        if is_feat:
            feat, stages = self.features(x, return_stages=True)
            ends, acc = [], -1
            for nb in self.n_blocks:
                acc += nb
                ends.append(stages[acc])
        else:
            feat = self.features(x)
        out = feat
        if self.num_classes > 0:
            out = HF.LinearFn.apply(feat, self.classifier.weight, self.classifier.bias)
        if is_feat:
            return ends + [feat], out
        return out

    # ------------------------------------------------------------------ classifier surgery + L2-to-old-weights regs
    def _get_base_weights(self):
        base_weight = self.classifier.weight.detach().clone().requires_grad_(False)
        if self.classifier.bias is not None:
            return base_weight, self.classifier.bias.detach().clone().requires_grad_(False)
        return base_weight, None

    def augment_base_classifier_(self, n, novel_weight=None, novel_bias=None):
        base_device = self.classifier.weight.device
        base_weight = self.classifier.weight.detach()
        base_bias = self.classifier.bias.detach() if self.classifier.bias is not None else None
        if novel_weight is None:
            novel_classifier = nn.Linear(base_weight.size(1), n, bias=(base_bias is not None))
            novel_weight = novel_classifier.weight.detach()
            if base_bias is not None and novel_bias is None:
                novel_bias = novel_classifier.bias.detach()
        augmented_weight = torch.cat([base_weight, novel_weight.to(base_device)], 0)
        self.classifier.weight = nn.Parameter(augmented_weight, requires_grad=True)
        if base_bias is not None:
            self.classifier.bias = nn.Parameter(torch.cat([base_bias, novel_bias.to(base_device)]), requires_grad=True)

    def regloss(self, lmbd, base_weight, base_bias=None):
        reg = HF.FrobFn.apply(self.classifier.weight[:base_weight.size(0), :], base_weight, lmbd)
        if base_bias is not None:
            reg = reg + HF.SqDiffFn.apply(self.classifier.bias[:base_weight.size(0)], base_bias, lmbd)
        return reg

    def reglossnovel(self, lmbd, novel_weight, novel_bias=None):
        rng1, rng2 = self.num_classes, self.num_classes + novel_weight.size(0)
        reg = HF.FrobFn.apply(self.classifier.weight[rng1:rng2, :], novel_weight, lmbd)
        if novel_bias is not None:
            reg = reg + HF.SqDiffFn.apply(self.classifier.bias[rng1:rng2], novel_bias, lmbd)
        return reg


def resnet12(keep_prob=1.0, avg_pool=False, **kwargs):
    return ResNet(None, [1, 1, 1, 1], keep_prob=keep_prob, avg_pool=avg_pool, **kwargs)


def resnet18(keep_prob=1.0, avg_pool=False, **kwargs):
    return ResNet(None, [1, 1, 2, 2], keep_prob=keep_prob, avg_pool=avg_pool, **kwargs)


model_pool = ["resnet12", "resnet18"]
model_dict = {"resnet12": resnet12, "resnet18": resnet18}


def create_model(name, n_cls, opt, vocab=None, dataset="miniImageNet"):
    """models/util.py:6-35 for the two pooled models."""
    if dataset not in ("miniImageNet", "tieredImageNet"):
        raise NotImplementedError("dataset not supported: {}".format(dataset))
    if name not in model_dict:
        raise NotImplementedError("model {} not supported in dataset {}:".format(name, dataset))
    return model_dict[name](avg_pool=True, drop_rate=0.1, dropblock_size=5, num_classes=n_cls, vocab=vocab, opt=opt)
