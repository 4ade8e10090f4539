"""NumPy restatement of the regularizers on the path (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/models/resnet_language.py:
  LangPuller.get_projected_weight :92-97   thin QR of W_base^T, P = ((wQ)/||Q^T rows||) Q^T
  LangPuller.loss1                :89-90   pull * ||P - w||_F^2      (squared)
  LangPuller.forward              :75-87   semantic target softmax(E_n E_b^T / temp) W_base, or LinearMap(E_n)
  models/util.py get_embeds       :50-66   class-name embedding = mean of word vectors (with its unknown-word quirk)
  ResNet.regloss                  :229-233 lmbd * ||W[:nb] - W_base||_F   (NOT squared)
  ResNet.reglossnovel             :235-240 lmbd * ||W[nc:nc+k] - W_prev||_F (NOT squared)
Gradients are what torch autograd produces for those expressions (the gradient of
loss1 flows through BOTH `inspired` and `weights`, eval/language_eval.py:282-287;
torch.norm has sub-gradient 0 at 0).
Parity pinned by tests/golden/reg_*.npz (tools/make_golden.py).
"""
import numpy as np


def orthonormal_basis(base_weight):
    """Q [dim, n_base] of the thin QR of base_weight^T (torch.qr(tr, some=True), :93-94)."""
    q, _ = np.linalg.qr(np.asarray(base_weight, dtype=np.float64).T, mode="reduced")
    return q


def get_projected_weight(base_weight, weights, dtype=np.float32):
    q = orthonormal_basis(base_weight)
    w = np.asarray(weights, dtype=np.float64)
    mut = w @ q
    mutnorm = mut / np.linalg.norm(q.T, axis=1)[None, :]
    return (mutnorm @ q.T).astype(dtype)


def loss1_and_grad(pull, base_weight, weights):
    """pull*||P(w)-w||^2 and d/dw with P differentiable in w (both paths)."""
    q = orthonormal_basis(base_weight)
    w = np.asarray(weights, dtype=np.float64)
    d = 1.0 / np.linalg.norm(q.T, axis=1)            # column norms of Q (== 1 up to rounding)
    a = (q * d[None, :]) @ q.T                         # P = w @ a
    r = w @ a - w                                      # residual [k, dim]
    loss = pull * float((r * r).sum())
    grad = 2.0 * pull * (r @ a.T - r)                  # d/dw ||w(a - I)||^2
    return loss, grad


def loss1_to_target_and_grad(pull, target, weights):
    """pull*||target - w||^2 for a CONSTANT target (semantic / linear-mapping variants, :75-90)."""
    r = np.asarray(weights, np.float64) - np.asarray(target, np.float64)
    return pull * float((r * r).sum()), 2.0 * pull * r


def get_embeds(table, vocab, dim=500):
    """models/util.py:50-66: per class name the mean of its words' vectors.  Restated with the reference's quirk: an
    unknown word RESETS the running sum to zeros (words after it are still added) and the divisor stays len(words)."""
    out = []
    for token in vocab:
        words = token.split(" ")
        acc = 0
        for w in words:
            if w in table:
                acc = acc + np.asarray(table[w])
            else:
                acc = np.zeros(dim)
        out.append(np.asarray(acc, np.float64) / len(words))
    return np.stack(out, 0).astype(np.float32)             # .float() at resnet_language.py:32,47


def semantic_target(novel_embeds, base_embeds, base_weight, temp=1.0, mask=False):
    """LangPuller.forward, resnet_language.py:75-83: softmax(E_n E_b^T / temp [diag := -9999]) @ W_base.  Returns
    (target, probs) in float64."""
    scores = np.asarray(novel_embeds, np.float64) @ np.asarray(base_embeds, np.float64).T
    if mask:
        np.fill_diagonal(scores, -9999.0)
    z = scores / temp
    z = z - z.max(axis=1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(axis=1, keepdims=True)
    return p @ np.asarray(base_weight, np.float64), p


def linear_map_target(novel_embeds, map_weight, map_bias):
    """LangPuller.forward with a mapping model (:84-87): LinearMap(novel_embeds) = E_n W^T + b."""
    return np.asarray(novel_embeds, np.float64) @ np.asarray(map_weight, np.float64).T + np.asarray(map_bias, np.float64)


def frob_reg_and_grad(lmbd, w_rows, anchor):
    """lmbd*||w_rows - anchor||_F (not squared) and its gradient (0 at 0 like torch.norm)."""
    d = np.asarray(w_rows, np.float64) - np.asarray(anchor, np.float64)
    nrm = float(np.sqrt((d * d).sum()))
    if nrm == 0.0:
        return 0.0, np.zeros_like(d)
    return lmbd * nrm, lmbd * d / nrm
