"""torch-CPU restatement of one fine-tune epoch of the incremental loop (TEST INFRASTRUCTURE, see oracle/__init__.py).

Same algorithm as oracle/resnet_ref.py + oracle/loop_ref.py, but on the third-party library the reference itself
computes with (torch.nn.functional on CPU tensors: MKL-DNN convolutions, native batch_norm, autograd, torch.qr's
successor torch.linalg.qr), so that `bench.py`'s `cpu_baseline` times what the reference's CPU path would cost on the
same host instead of NumPy's nine-GEMM convolution.  Follows
  models/resnet_language.py  ResNet.forward :170-192, BasicBlock.forward :268-301 (eval mode), LangPuller
                             get_projected_weight :92-97 / loss1 :89-90, ResNet.regloss :229-233, reglossnovel :235-240
  eval/language_eval.py      one epoch of the fine-tune loop :252-295 (support forward, CE + three regularizers,
                             SGD step on classifier.weight) and the per-epoch validation :18-43
  eval/util.py               get_optim :92-102 (SGD lr/momentum/wd), accuracy :26-40
Pinned by tests/test_oracle_golden.py against tests/golden/backbone.npz (features) and, for the classifier step, against the NumPy restatement (loop_ref +
subspace_ref, themselves pinned by the loop goldens).  Nothing under subspace-reg_amd/ imports this file.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .resnet_ref import BN_EPS, LEAK, block_specs


class TorchCpuRef:
    def __init__(self, sd, n_blocks=(1, 1, 2, 2)):
        self.specs = block_specs(n_blocks)
        self.sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items() if np.asarray(v).dtype != np.int64}

    def _bn(self, x, p):
        sd = self.sd
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            False, 0.1, BN_EPS)

    @torch.no_grad()
    def features(self, x):
        """Eval-mode backbone: [B,3,H,W] float32 tensor -> [B,640] (resnet_language.py:170-182, 268-301)."""
        sd = self.sd
        for s in self.specs:
            n = s["name"]
            out = F.leaky_relu(self._bn(F.conv2d(x, sd[n + ".conv1.weight"], padding=1), n + ".bn1"), LEAK)
            out = F.leaky_relu(self._bn(F.conv2d(out, sd[n + ".conv2.weight"], padding=1), n + ".bn2"), LEAK)
            out = self._bn(F.conv2d(out, sd[n + ".conv3.weight"], padding=1), n + ".bn3")
            res = self._bn(F.conv2d(x, sd[n + ".downsample.0.weight"]), n + ".downsample.1") if s["downsample"] else x
            out = F.leaky_relu(out + res, LEAK)
            x = F.max_pool2d(out, s["stride"]) if s["stride"] > 1 else out
        return F.adaptive_avg_pool2d(x, 1).flatten(1)


def projected_weight(base_weight, w):
    """LangPuller.get_projected_weight (:92-97): thin QR of W_base^T per call, as the reference does every epoch."""
    q, _ = torch.linalg.qr(base_weight.t(), mode="reduced")
    mut = w @ q
    return (mut / torch.norm(q.t(), dim=1).unsqueeze(0)) @ q.t()


def finetune_epoch(net, W, mom, base_weight, prev_rows, support_x, support_y, query_sets, hp):
    """One epoch >= 2 of language_eval.py:252-326 with the backbone frozen: support forward, loss, backward, SGD step on
    W (in place, momentum buffer `mom`), then validation of every query set.  Returns (loss, [accuracy per set])."""
    n_base = base_weight.shape[0]
    feat = net.features(support_x)
    Wp = W.detach().clone().requires_grad_(True)
    loss = F.cross_entropy(feat @ Wp.t(), support_y)
    loss = loss + hp["lmbd_base"] * torch.norm(Wp[:n_base] - base_weight)                      # regloss :229-233
    if prev_rows is not None and prev_rows.shape[0]:
        k = prev_rows.shape[0]
        loss = loss + hp["lmbd_prev"] * torch.norm(Wp[n_base:n_base + k] - prev_rows)          # reglossnovel :235-240
    novel = Wp[n_base + (0 if prev_rows is None else prev_rows.shape[0]):]
    loss = loss + hp["pull"] * torch.norm(projected_weight(base_weight, novel) - novel) ** 2   # loss1 :89-90
    loss.backward()
    with torch.no_grad():                                                                      # torch.optim.SGD, eval/util.py:98-101
        d = Wp.grad + hp["wd"] * W
        if mom is None:
            mom = d.clone()
        else:
            mom.mul_(hp["momentum"]).add_(d)
        W.sub_(hp["lr"] * mom)
        accs = []
        for qx, qy in query_sets:                                                              # validate :18-43
            pred = (net.features(qx) @ W.t()).argmax(1)
            accs.append(float((pred == qy).float().sum() * (100.0 / len(qy))))
    return float(loss.detach()), accs, mom


# ---------------------------------------------------------------------------------------------------------------------
# Pretraining step (train_supervised.py:229-244: output = model(input); loss = criterion(output, target); loss.backward())
# with the STORAGE ROUNDING of the bf16 HIP path emulated: every tensor that path keeps in HBM as bf16 - the packed input, the
# raw convolution outputs, the activations, the block outputs, and in the backward the gradients with respect to exactly those
# tensors plus the pre-activation sum of a block - is rounded to bf16 here too (round-to-nearest-even, like the kernels);
# convolutions / BatchNorm / reductions accumulate in fp32 on both sides.  With the same rounding points a LeakyReLU / MaxPool
# decision flips only where the two accumulation ORDERS differ by an ulp, so the HIP gradients can be gated much tighter against
# this than against the reference's fp32 autograd (tests/test_hip_train.py).
def _round_bf16_t(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _StoreBf16(torch.autograd.Function):
    """A tensor stored as bf16 whose gradient is stored as bf16 too."""

    @staticmethod
    def forward(ctx, x):
        return _round_bf16_t(x)

    @staticmethod
    def backward(ctx, g):
        return _round_bf16_t(g)


class _GradBf16(torch.autograd.Function):
    """A value that lives in registers in the forward (not rounded) but whose gradient is stored as bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _round_bf16_t(g)


def train_step_grads(sd, x, labels, masks, bf16=True, n_blocks=(1, 1, 2, 2), no_dropblock=True, eval_mode=False):
    """One train-mode forward + backward of the backbone + linear classifier on CPU.  sd: state_dict (numpy), x [B,3,H,W],
    labels [B], masks: a MaskSource (consumed in the reference's order).  Returns (loss, {parameter name: gradient}).
    eval_mode: the model in eval mode with every parameter still requiring grad - BatchNorm normalises with its running
    statistics, no dropout / DropBlock (eval/language_eval.py:242-295 before freeze_backbone_at, after the first validate())."""
    from .resnet_ref import DROP_RATE, dropblock_gamma
    assert no_dropblock, "block_size 1 only (every script of the reference passes --no_dropblock)"
    store = _StoreBf16.apply if bf16 else (lambda t: t)
    gstore = _GradBf16.apply if bf16 else (lambda t: t)
    P = {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(True) for k, v in sd.items()
         if np.asarray(v).dtype != np.int64 and "running_" not in k}

    def wq(name):                                   # the packed bf16 copy the kernels multiply with; gradient goes to the fp32 master
        w = P[name]
        return w + (_round_bf16_t(w) - w).detach() if bf16 else w

    def bn(t, p):
        if eval_mode:
            return F.batch_norm(t, torch.from_numpy(np.asarray(sd[p + ".running_mean"])), torch.from_numpy(np.asarray(sd[p + ".running_var"])),
                                P[p + ".weight"], P[p + ".bias"], False, 0.1, BN_EPS)
        return F.batch_norm(t, None, None, P[p + ".weight"], P[p + ".bias"], True, 0.1, BN_EPS)

    a = store(torch.as_tensor(x, dtype=torch.float32))
    for s in block_specs(n_blocks):
        n = s["name"]
        out = store(F.leaky_relu(bn(store(F.conv2d(a, wq(n + ".conv1.weight"), padding=1)), n + ".bn1"), LEAK))
        out = store(F.leaky_relu(bn(store(F.conv2d(out, wq(n + ".conv2.weight"), padding=1)), n + ".bn2"), LEAK))
        out = bn(store(F.conv2d(out, wq(n + ".conv3.weight"), padding=1)), n + ".bn3")
        res = bn(store(F.conv2d(a, wq(n + ".downsample.0.weight"))), n + ".downsample.1") if s["downsample"] else a
        out = F.leaky_relu(gstore(out + res), LEAK)
        if s["stride"] > 1:
            out = F.max_pool2d(out, s["stride"])
        B, C, H, W = out.shape
        if eval_mode:
            a = store(out)
            continue
        if s["drop_block"]:                         # DropBlock with block_size 1 (:311-325): drop with probability gamma
            gamma = dropblock_gamma(1, H, 1)
            keep = 1.0 - masks.bernoulli((B, C, H, W), gamma)
            scale = keep.size / keep.sum()
        else:
            keep = masks.dropout_keep((B, C, H, W), DROP_RATE)
            scale = np.float32(1.0) / np.float32(1.0 - DROP_RATE)
        a = store(out * torch.from_numpy(keep.astype(np.float32)) * float(scale))
    feat = a.mean(dim=(2, 3))
    logits = feat @ P["classifier.weight"].t()
    loss = F.cross_entropy(logits, torch.as_tensor(labels, dtype=torch.long))
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in P.items() if v.grad is not None}
