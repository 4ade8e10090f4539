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
