"""NumPy restatement of the regularizers on the path (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/models/resnet_language.py:
  LangPuller.get_projected_weight :92-97   thin QR of W_base^T, P = ((wQ)/||Q^T rows||) Q^T
  LangPuller.loss1                :89-90   pull * ||P - w||_F^2      (squared)
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


def frob_reg_and_grad(lmbd, w_rows, anchor):
    """lmbd*||w_rows - anchor||_F (not squared) and its gradient (0 at 0 like torch.norm)."""
    d = np.asarray(w_rows, np.float64) - np.asarray(anchor, np.float64)
    nrm = float(np.sqrt((d * d).sum()))
    if nrm == 0.0:
        return 0.0, np.zeros_like(d)
    return lmbd * nrm, lmbd * d / nrm
