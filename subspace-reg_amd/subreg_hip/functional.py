"""torch.autograd.Function stubs over the C ABI (the reference's "nn.Module + autograd" seam).

Every op launches a hand-written gfx950 kernel on torch's current stream; torch only owns the
memory and records the graph.  Reference call sites are named on each class.
"""
import torch

from . import _lib


def _f32c(t):
    assert t.is_cuda, "subreg_hip ops need CUDA (HIP) tensors - there is no CPU fallback"
    assert t.dtype == torch.float32
    return t.contiguous()


class LinearFn(torch.autograd.Function):
    """nn.Linear(640, n_cls, bias) - models/resnet_language.py:138-140,187."""

    @staticmethod
    def forward(ctx, feat, weight, bias):
        lib = _lib.load()
        feat, weight = _f32c(feat), _f32c(weight)
        bias_c = _f32c(bias) if bias is not None else None
        B, D = feat.shape
        N = weight.shape[0]
        logits = torch.empty(B, N, dtype=torch.float32, device=feat.device)
        _lib.check(lib.subreg_linear_fwd(_lib.ptr(feat), _lib.ptr(weight), _lib.ptr(bias_c), _lib.ptr(logits), B, N, D,
                                         _lib.stream_ptr()), "linear_fwd")
        ctx.save_for_backward(feat, weight)
        ctx.has_bias = bias is not None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        feat, weight = ctx.saved_tensors
        dlogits = _f32c(dlogits)
        B, D = feat.shape
        N = weight.shape[0]
        need_f, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dfeat = torch.empty_like(feat) if need_f else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(N, dtype=torch.float32, device=feat.device) if need_b else None
        _lib.check(lib.subreg_linear_bwd(_lib.ptr(dlogits), _lib.ptr(feat), _lib.ptr(weight), _lib.ptr(dw), _lib.ptr(db),
                                         _lib.ptr(dfeat), B, N, D, _lib.stream_ptr()), "linear_bwd")
        return dfeat, dw, db


class SubspaceProjectFn(torch.autograd.Function):
    """P = w Q^T Q, Q rows = orthonormal basis of span(W_base) - LangPuller.get_projected_weight, :92-97.
    The projector is symmetric, so the backward is the same projection applied to grad_output."""

    @staticmethod
    def forward(ctx, w, q):
        lib = _lib.load()
        w, q = _f32c(w), _f32c(q)
        k, D = w.shape
        p = torch.empty_like(w)
        _lib.check(lib.subreg_subspace_project(_lib.ptr(w), _lib.ptr(q), _lib.ptr(p), k, q.shape[0], D,
                                               _lib.stream_ptr()), "subspace_project")
        ctx.save_for_backward(q)
        return p

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (q,) = ctx.saved_tensors
        g = _f32c(g)
        out = torch.empty_like(g)
        _lib.check(lib.subreg_subspace_project(_lib.ptr(g), _lib.ptr(q), _lib.ptr(out), g.shape[0], q.shape[0], g.shape[1],
                                               _lib.stream_ptr()), "subspace_project(bwd)")
        return out, None


def subspace_basis(base_weight):
    """Orthonormal basis rows [n_base, D] of span(base_weight rows) (replaces torch.qr, :93-94)."""
    lib = _lib.load()
    wb = _f32c(base_weight.detach())
    nb, D = wb.shape
    q = torch.empty_like(wb)
    scratch = torch.empty(nb * D, dtype=torch.float64, device=wb.device)
    info = torch.zeros(1, dtype=torch.int32, device=wb.device)
    _lib.check(lib.subreg_subspace_basis(_lib.ptr(wb), _lib.ptr(q), _lib.ptr(scratch), nb, D, _lib.ptr(info),
                                         _lib.stream_ptr()), "subspace_basis")
    return q, info


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() with mean reduction (train_supervised.py:138, language_eval.py:252); `counters` (optional
    int32[2] tensor) receives the batch's top-1 / top-5 hit counts in the same launch."""

    @staticmethod
    def forward(ctx, logits, target, counters=None, topk=5):
        lib = _lib.load()
        z = _f32c(logits)
        t = target.to(torch.int64).contiguous()
        B, N = z.shape
        rowloss = torch.empty(B, dtype=torch.float32, device=z.device)
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        _lib.check(lib.subreg_softmax_ce(_lib.ptr(z), _lib.ptr(t), B, N, int(topk), _lib.ptr(rowloss), _lib.ptr(loss), _lib.ptr(dz),
                                         _lib.ptr(counters), _lib.stream_ptr()), "softmax_ce")
        ctx.save_for_backward(dz)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None, None, None


def cross_entropy(logits, target, counters=None, topk=5):
    return CrossEntropyFn.apply(logits, target, counters, topk)


class SemanticTargetFn(torch.autograd.Function):
    """softmax(E_novel E_base^T / temp [diag := -9999]) @ W_base - LangPuller.forward, :75-83 (gradient to W_base)."""

    @staticmethod
    def forward(ctx, base_weight, novel_embeds, base_embeds, temp, mask):
        lib = _lib.load()
        wb, en, eb = _f32c(base_weight), _f32c(novel_embeds), _f32c(base_embeds)
        assert en.shape[1] == eb.shape[1] and eb.shape[0] == wb.shape[0]
        probs = torch.empty(en.shape[0], eb.shape[0], dtype=torch.float32, device=wb.device)
        out = torch.empty(en.shape[0], wb.shape[1], dtype=torch.float32, device=wb.device)
        _lib.check(lib.subreg_semantic_target(_lib.ptr(en), _lib.ptr(eb), _lib.ptr(wb), en.shape[0], eb.shape[0], en.shape[1],
                                              wb.shape[1], float(temp), int(bool(mask)), _lib.ptr(probs), _lib.ptr(out),
                                              _lib.stream_ptr()), "semantic_target")
        ctx.save_for_backward(probs)
        ctx.shape = tuple(wb.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (probs,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        lib = _lib.load()
        g = _f32c(g)
        gw = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        _lib.check(lib.subreg_semantic_target_bwd(_lib.ptr(probs), _lib.ptr(g), probs.shape[0], probs.shape[1], ctx.shape[1],
                                                  _lib.ptr(gw), _lib.stream_ptr()), "semantic_target(bwd)")
        return gw, None, None, None, None


class SqDiffFn(torch.autograd.Function):
    """pull * ||inspired - weights||_F^2 - LangPuller.loss1, :89-90 (gradient to both arguments)."""

    @staticmethod
    def forward(ctx, inspired, weights, pull):
        lib = _lib.load()
        a, b = _f32c(inspired), _f32c(weights)
        assert a.shape == b.shape
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        _lib.check(lib.subreg_sqdiff(_lib.ptr(a), _lib.ptr(b), a.numel(), float(pull), _lib.ptr(loss), None, 0.0, None,
                                     None, _lib.stream_ptr()), "sqdiff")
        ctx.save_for_backward(a, b)
        ctx.pull = float(pull)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = _f32c(g.reshape(1))
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        _lib.check(lib.subreg_sqdiff(_lib.ptr(a), _lib.ptr(b), a.numel(), 0.0, None, _lib.ptr(g), 2.0 * ctx.pull,
                                     _lib.ptr(ga), _lib.ptr(gb), _lib.stream_ptr()), "sqdiff(bwd)")
        return ga, gb, None


class FrobFn(torch.autograd.Function):
    """lmbd * ||rows - anchor||_F, NOT squared, zero sub-gradient at 0 - ResNet.regloss/reglossnovel, :229-240."""

    @staticmethod
    def forward(ctx, rows, anchor, lmbd):
        lib = _lib.load()
        a, b = _f32c(rows), _f32c(anchor)
        assert a.shape == b.shape
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        _lib.check(lib.subreg_frob(_lib.ptr(a), _lib.ptr(b), a.numel(), float(lmbd), _lib.ptr(loss), None, None,
                                   _lib.stream_ptr()), "frob")
        ctx.save_for_backward(a, b)
        ctx.lmbd = float(lmbd)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = _f32c(g.reshape(1))
        ga = torch.empty_like(a)
        _lib.check(lib.subreg_frob(_lib.ptr(a), _lib.ptr(b), a.numel(), ctx.lmbd, None, _lib.ptr(g), _lib.ptr(ga),
                                   _lib.stream_ptr()), "frob(bwd)")
        return ga, None, None
