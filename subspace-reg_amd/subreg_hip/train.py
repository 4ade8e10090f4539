"""Pretraining step on the HIP backbone: train-mode forward with a stash + backward, exposed to torch autograd.

Counterpart of /root/reference/train_supervised.py:229-244 (`output = model(input)`; `loss.backward()`): when the
backbone's parameters require grad, `ResNet.features` routes through `BackboneTrainFn`, whose backward fills the
gradients of all conv weights and BN affine parameters with the gfx950 kernels of csrc/backward.hip.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib


class TrainStash:
    """Caller-owned buffers of one (B, H, W) training step (include/subreg_hip.h: subreg_train_desc)."""

    def __init__(self, hb, B, H, W):
        self.hb, self.shape = hb, (B, H, W)
        lib, dev, td = hb.lib, hb.device, hb.tdtype
        f32 = torch.float32
        self.keep = []                       # every tensor the descriptors point at
        self.named = {}                      # (block, slot, field) -> tensor, for tests / diagnostics
        self.blk = (_lib.BlockTrain * len(hb.blocks))()
        self.desc = _lib.TrainDesc()
        self.desc.blocks = C.cast(self.blk, C.POINTER(_lib.BlockTrain))

        def buf(n, dtype=td):
            t = torch.empty(int(n), dtype=dtype, device=dev)
            self.keep.append(t)
            return t

        # the packed input images (layer1.0's conv1 / shortcut read them in the forward AND for their weight gradients): per stash,
        # not in the backbone's workspace - a second forward before this one's backward (replay memory) must not overwrite them
        self.col = buf(B * H * W * 32)
        h, w, cmax, amax, pmax = H, W, 0, B * H * W * 32, 0
        gw_need, gw_down_need, gw_users, grad_slots = 0, 0, [], []
        self.grads = {}                      # state_dict key -> fp32 gradient tensor
        self.dgrad = []                      # (conv name, cout, cin, k, packed tensor)
        for bi, (name, cin, cout, stride, ds, _db) in enumerate(hb.blocks):
            npix = B * h * w
            amax, cmax = max(amax, npix * cout), max(cmax, cout)
            pmax = max(pmax, B * (h + 2) * (w + 2) * cout)
            convs = [("conv1", name + ".conv1", name + ".bn1", cin, 3, True), ("conv2", name + ".conv2", name + ".bn2", cout, 3, True),
                     ("conv3", name + ".conv3", name + ".bn3", cout, 3, False)]
            if ds:
                convs.append(("down", name + ".downsample.0", name + ".downsample.1", cin, 1, False))
            for slot, cname, bname, ci, k, has_act in convs:
                tc = getattr(self.blk[bi], slot)
                self.named[(bi, slot, "raw")] = buf(npix * cout)
                tc.raw = self.named[(bi, slot, "raw")].data_ptr()
                tc.act = None
                if has_act:
                    self.named[(bi, slot, "act")] = buf(npix * cout)
                    tc.act = self.named[(bi, slot, "act")].data_ptr()
                for f in ("mean", "invstd", "bscale", "bshift"):
                    self.named[(bi, slot, f)] = buf(cout, f32)
                    setattr(tc, f, self.named[(bi, slot, f)].data_ptr())
                kin = 32 if ci == 3 else ci
                kk = 1 if ci == 3 else k
                nsplit = lib.subreg_conv_wgrad_splits(B, h, w, kin, cout, kk, hb.dtype)
                if slot == "down":
                    gw_down_need = max(gw_down_need, nsplit * cout * kk * kk * kin)
                else:
                    gw_need = max(gw_need, nsplit * cout * kk * kk * kin)
                gw_users.append((tc, slot == "down"))
                grad_slots.append((tc, cname + ".weight", (cout, ci, k, k), bname + ".weight", bname + ".bias", cout))
                tc.w_dgrad = None
                if bi > 0 or slot in ("conv2", "conv3"):      # no gradient w.r.t. the images: layer1.0 conv1/shortcut need none
                    wd = buf(cout * k * k * ci)
                    self.named[(bi, slot, "w_dgrad")] = wd
                    tc.w_dgrad = wd.data_ptr()
                    self.dgrad.append((cname, cout, ci, k, wd))
            h, w = h // stride, w // stride
            self.named[(bi, "out")] = buf(B * h * w * cout)
            self.blk[bi].out = self.named[(bi, "out")].data_ptr()
        for i in range(2):
            self.desc.g[i] = buf(amax).data_ptr()
        for f in ("dv", "dr", "dt", "dr2"):
            setattr(self.desc, f, buf(amax).data_ptr())
        # every parameter gradient lives in ONE flat buffer; .grad of every backbone parameter is a view of it
        total = sum(int(np.prod(shape)) + 2 * c for _tc, _n, shape, _g, _b, c in grad_slots)
        self.flat_grads = torch.zeros(total, dtype=f32, device=dev)
        self.grad_views = []                                   # (name, offset, shape)
        off = 0
        for tc, wname, shape, gname, bname_, c in grad_slots:
            for name, shp, field in ((wname, shape, "grad_w"), (gname, (c,), "grad_gamma"), (bname_, (c,), "grad_beta")):
                n = int(np.prod(shp))
                self.grad_views.append((name, off, shp))
                self.grads[name] = self.flat_grads[off:off + n].view(shp)
                setattr(tc, field, self.grads[name].data_ptr())
                off += n
        # contiguous range of every block's parameter gradients in the flat buffer (conv weights + BN affine, block order):
        # a data-parallel caller all-reduces a stage's range while the backward of the earlier blocks is still running
        self.block_ranges, off = [], 0
        per_block = {}
        for name, o, shp in self.grad_views:
            bi = [i for i, (bn, *_r) in enumerate(hb.blocks) if name.startswith(bn + ".")][0]
            lo, hi = per_block.get(bi, (o, o))
            per_block[bi] = (min(lo, o), max(hi, o + int(np.prod(shp))))
        self.block_ranges = [per_block[i] for i in range(len(hb.blocks))]
        self.flat_grads._subreg_stash = self     # lets the optimiser recognise gradients that are views of this buffer
        self.conv_weight_offsets = {off: name for name, off, shp in self.grad_views if len(shp) == 4}
        self.opt_packed = None                   # conv-weight versions for which the optimiser already wrote the packed copies
        # scratch for the per-split partial dW: one buffer for the 3x3 convs (their dW chains run one at a time, on one stream) and
        # one for the 1x1 shortcut convs, whose branch may run on the other stream at the same time (backbone_train.hip)
        gw_shared, gw_down = buf(gw_need, f32), buf(max(gw_down_need, 1), f32)
        for tc, is_down in gw_users:
            tc.gw_packed = (gw_down if is_down else gw_shared).data_ptr()
        self.desc.pad_x = buf(pmax).data_ptr() if hb.dtype == _lib.BF16 else None
        self.desc.pad_dy = buf(pmax).data_ptr() if hb.dtype == _lib.BF16 else None
        self.desc.bn_partial = buf(lib.subreg_bn_bwd_slices(B * H * W) * cmax * 2, torch.float64).data_ptr()
        z = torch.zeros(cmax, dtype=f32, device=dev)
        self.keep.append(z)
        self.desc.zero_shift = z.data_ptr()
        # K-split workspace of the small-M 3x3 convolutions (subreg_conv_fwd_ws): forward shapes and dX shapes (channels swapped)
        need, h, w = 0, H, W
        for bi, (name, cin, cout, stride, ds, _db) in enumerate(hb.blocks):
            for ci, co in ((cin, cout), (cout, cout), (cout, cin)):
                if ci % 32 == 0:
                    need = max(need, lib.subreg_conv_splitk_floats(B, h, w, ci, co, 3, hb.dtype))
            h, w = h // stride, w // stride
        if need > 0:
            self.desc.splitk_ws, self.desc.splitk_ws_floats = buf(need, f32).data_ptr(), need
        # two-stream schedule (subreg_train_desc.side_stream; SUBREG_TRAIN_ONE_STREAM=1 keeps the step on one stream for A/B runs):
        # dW chains and the shortcut branch beside the BatchNorm-backward -> dX chain.  The stream, the events and the three extra
        # buffers live as long as this stash.
        self.side_stream, self._events = None, []
        if os.environ.get("SUBREG_TRAIN_ONE_STREAM", "0") != "1":
            # (a stream whose kernels really run beside the caller's: HIP deals streams onto a few hardware queues in creation order, and two
            # streams on one queue execute one after the other - HipBackbone._parallel_streams picks by measurement, once per backbone)
            if getattr(hb, "_train_side_stream", None) is None:
                hb._train_side_stream = hb._parallel_streams(1)[0]
            self.side_stream = hb._train_side_stream
            self.desc.side_stream = self.side_stream.cuda_stream
            for i in range(len(self.desc.events)):
                e = C.c_void_p()
                _lib.check(lib.subreg_event_create(C.byref(e)), "event_create")
                self._events.append(e)
                self.desc.events[i] = e.value
            self.desc.dr_alt = buf(amax).data_ptr()
            self.desc.bn_partial_side = buf(lib.subreg_bn_bwd_slices(B * H * W) * cmax * 2, torch.float64).data_ptr()
            self.desc.stats_side = buf(lib.subreg_backbone_stats_floats(C.byref(hb._desc), B, H, W), f32).data_ptr()

    def __del__(self):
        # the events are the only thing here that torch's allocator does not own
        try:
            if self._events:
                torch.cuda.synchronize()
                for e in self._events:
                    self.hb.lib.subreg_event_destroy(e)
                self._events = []
        except Exception:
            pass

    def repack_dgrad(self):
        s = _lib.stream_ptr()
        p = self.hb.params
        for cname, cout, ci, k, wd in self.dgrad:
            _lib.check(self.hb.lib.subreg_pack_conv_weight_dgrad(_lib.ptr(p[cname + ".weight"]), _lib.ptr(wd), cout, ci, k,
                                                                 self.hb.dtype, s), "pack_conv_weight_dgrad")


def backward_stages(n_blocks):
    """(first_block, last_block) ranges of the staged backward, last stage first.  resnet18's six blocks -> [5], [4], [2, 3],
    [0, 1]: 44 MB / 37 MB / 24 MB / 2.6 MB of fp32 gradients - collectives large enough for the xGMI rings, and the largest
    ones start while four more blocks of backward work remain to hide them."""
    if n_blocks <= 2:
        return [(0, n_blocks - 1)]
    stages = [(n_blocks - 1, n_blocks - 1), (n_blocks - 2, n_blocks - 2)]
    rest = n_blocks - 2
    if rest > 2:
        stages.append((rest // 2, rest - 1))
        stages.append((0, rest // 2 - 1))
    else:
        stages.append((0, rest - 1))
    return stages


def conv_weight_versions(hb):
    """(data_ptr, _version) of every conv weight: changes whenever torch code rebinds or writes one of them (the library's own
    in-place optimiser step goes through raw pointers and does not count)."""
    return tuple((hb.params[c + ".weight"].data_ptr(), hb.params[c + ".weight"]._version) for _bi, _slot, c, *_r in hb._convs())


def _take_stash(hb, B, H, W):
    """The stash of one gradient-carrying forward.  A stash is busy from its forward until its backward; a step with two forwards
    before the backward (support, then replay memory: eval/language_eval.py:252-258) therefore gets two.  At most two per shape and
    four in all are kept: a forward whose backward never comes (a loss that is only looked at) must not pile stashes up - the least
    recently used one is then taken over, which is what every forward did before."""
    pool = hb.__dict__.setdefault("_train_stashes", [])
    same = [st for st in pool if st.shape == (B, H, W)]
    stash = next((st for st in same if not st.in_flight), None)
    if stash is None and len(same) >= 2:
        stash = min(same, key=lambda st: st.tick)
    if stash is None:
        idle = sorted((st for st in pool if not st.in_flight), key=lambda st: st.tick)
        while len(pool) >= 4 and idle:
            pool.remove(idle.pop(0))
        stash = TrainStash(hb, B, H, W)
        pool.append(stash)
    hb._stash_tick = getattr(hb, "_stash_tick", 0) + 1
    stash.tick, stash.in_flight = hb._stash_tick, True
    hb._train_stash = stash                  # the most recent one (is_feat, tests)
    return stash


class BackboneTrainFn(torch.autograd.Function):
    """feat = backbone(x) with a stash; backward -> gradients of every backbone parameter (`names` order).  masks is a MaskSource /
    None for a TRAIN-mode forward (batch statistics, running-stat update, dropout / DropBlock), or the string "eval" for an
    eval-mode forward (running statistics, no masks: whole-network fine-tuning before freeze_backbone_at with the model in eval
    mode, eval/language_eval.py:242-295 after the first validate())."""

    @staticmethod
    def forward(ctx, x, hb, masks, names, *params):
        B, _, H, W = x.shape
        x = x.contiguous().float()
        eval_mode = isinstance(masks, str) and masks == "eval"
        for i in range(len(hb.nbt)):             # BasicBlock's own forward counter (resnet_language.py:269: every call, either mode)
            hb.nbt[i] += 1
        stash = _take_stash(hb, B, H, W)
        hb._ensure_workspace(B, H, W)
        stash.desc.eval_mode = 1 if eval_mode else 0
        if eval_mode:
            hb._clear_masks()
        else:
            hb._prepare_masks(B, H, W, masks)
        # the keep masks belong to THIS forward: another forward before this one's backward (replay memory) prepares its own
        stash.mask_state = (list(hb._keep), getattr(hb, "_mask_scale_dev", None), getattr(hb, "_mask_counts", None),
                            [(hb._blk[bi].keep_mask, hb._blk[bi].mask_scale, hb._blk[bi].mask_scale_dev) for bi in range(len(hb.blocks))])
        # weights moved since the last step: every conv's raw forward copy and dX copy in ONE launch (the eval-mode folded
        # copies are not needed here and are rebuilt by the next eval-mode forward) - unless the optimiser's fused step
        # (SGD.step -> subreg_sgd_pack_train) already wrote them for exactly these weights
        hb._bind_pointers()
        if stash.opt_packed is None or stash.opt_packed != conv_weight_versions(hb):
            _lib.check(hb.lib.subreg_backbone_pack_train(C.byref(hb._desc), C.byref(stash.desc), _lib.stream_ptr()), "backbone_pack_train")
        stash.opt_packed = None
        feat = torch.empty(B, hb.out_dim, dtype=torch.float32, device=x.device)
        col_keep, hb._desc.col = hb._desc.col, stash.col.data_ptr()
        try:
            _lib.check(hb.lib.subreg_backbone_forward_stash(C.byref(hb._desc), C.byref(stash.desc), _lib.ptr(x), B, H, W,
                                                            _lib.ptr(feat), _lib.stream_ptr()), "backbone_forward_stash")
        finally:
            hb._desc.col = col_keep
        hb._fold_versions = None             # running statistics moved
        ctx.hb, ctx.stash, ctx.names, ctx.bhw = hb, stash, names, (B, H, W)
        ctx.stash_tick = stash.tick          # backward refuses a stash that a later forward has taken over (see _take_stash)
        ctx.param_objs = params              # the Parameter objects themselves: backward assigns their .grad (see there)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        hb, stash = ctx.hb, ctx.stash
        B, H, W = ctx.bhw
        if stash.tick != ctx.stash_tick:
            raise RuntimeError("subreg_hip: this forward's stash was taken over by a later forward of the same shape (more than two "
                               "gradient-carrying forwards of one shape were pending): call backward() before forwarding again")
        dfeat = dfeat.contiguous().float()
        hook = getattr(hb, "grad_stage_hook", None)
        # Gradients that are already there (a second backward without zero_grad(), or zero_grad(set_to_none=False)) must be
        # ACCUMULATED into, like autograd does.  The kernels below overwrite the stash's flat buffer, and the usual `.grad` is a
        # view of exactly that buffer: take the old values out of it BEFORE the launches.
        flat = stash.flat_grads
        lo_ptr, hi_ptr = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        old = {}
        for n, p in zip(ctx.names, ctx.param_objs):
            if p.requires_grad and p.grad is not None:
                g = p.grad
                old[n] = g.clone() if lo_ptr <= g.data_ptr() < hi_ptr else g
        if old and not getattr(hb, "_warned_accumulate", False):
            # correct, but off the fast path: the sums below are fresh tensors, not views of the flat buffer, so SGD.step() takes
            # its per-tensor path (no fused update + re-pack launch) for this step - say so once
            import warnings
            hb._warned_accumulate = True
            warnings.warn("subreg_hip: backward() found existing .grad tensors and accumulates into copies (gradient accumulation / "
                          "zero_grad(set_to_none=False)); the fused SGD step needs zero_grad() with set_to_none=True before every "
                          "backward and is bypassed for this step", RuntimeWarning, stacklevel=2)
        if old and hook is not None:
            raise RuntimeError("subreg_hip: gradient accumulation over several backward passes is not supported together with the "
                               "data-parallel stage hook (the flat gradient buffer is being all-reduced in place): call "
                               "zero_grad() (set_to_none=True) before every backward")
        for bi, (km, ms, msd) in enumerate(stash.mask_state[3]):
            hb._blk[bi].keep_mask, hb._blk[bi].mask_scale, hb._blk[bi].mask_scale_dev = km, ms, msd
        col_keep, hb._desc.col = hb._desc.col, stash.col.data_ptr()
        try:
            if hook is None:
                _lib.check(hb.lib.subreg_backbone_backward(C.byref(hb._desc), C.byref(stash.desc), _lib.ptr(dfeat), B, H, W,
                                                           _lib.stream_ptr()), "backbone_backward")
            else:
                # data parallel: the backward is issued stage by stage (last blocks first); as soon as a stage's launches are queued
                # its gradient range goes to the hook (pretrain.GradientSync: asynchronous all-reduce on RCCL's stream, which waits
                # for exactly those launches) while this stream continues with the earlier blocks
                nb = len(hb.blocks)
                for first, last in backward_stages(nb):
                    _lib.check(hb.lib.subreg_backbone_backward_blocks(C.byref(hb._desc), C.byref(stash.desc), _lib.ptr(dfeat), B, H, W,
                                                                      first, last, _lib.stream_ptr()), "backbone_backward_blocks")
                    lo, hi = stash.block_ranges[first][0], stash.block_ranges[last][1]
                    hook(stash.flat_grads[lo:hi])
        finally:
            hb._desc.col = col_keep
        # Every parameter gradient is a view of the stash's ONE flat buffer (valid until the next backward of this model
        # overwrites it, i.e. for the optimiser step that follows).  The views are assigned to `.grad` here, directly:
        # handing them to autograd as return values makes AccumulateGrad copy each of the 66 tensors (a returned view is never
        # "stolen"), and the optimiser would then see 66 unrelated tensors instead of one buffer it can update with one launch.
        # CONTRACT: the parameter gradients do not travel through autograd's return values (all None below) - they are only
        # visible as `p.grad` after `loss.backward()`.  torch.autograd.grad(), backward(inputs=[...]), per-tensor hooks and
        # DDP-style reducers therefore see no backbone gradients; pretrain.GradientSync is the supported data-parallel path.
        views = {name: flat[off:off + int(np.prod(shp))].view(shp) for name, off, shp in stash.grad_views}
        for n, p in zip(ctx.names, ctx.param_objs):
            if not p.requires_grad:
                continue
            p.grad = views[n] if n not in old else old[n] + views[n]   # accumulate like autograd does (out of place)
        stash.in_flight = False
        return (None, None, None, None) + (None,) * len(ctx.names)


class SGD:
    """torch.optim.SGD(momentum, weight_decay) semantics (train_supervised.py:133-136) on the fused HIP kernel."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        self.params = [p for p in params]
        self.momentum, self.weight_decay = momentum, weight_decay
        # one param group, torch-style: adjust_learning_rate (util.py:45-51) writes param_groups[i]['lr']
        self.param_groups = [{"params": self.params, "lr": lr}]
        self.bufs = [None] * len(self.params)
        self._multi = {}                     # device pointer tables of the fused update, per parameter group
        self._stash_mom = {}                 # flat momentum buffer per backbone (laid out like a train stash's flat gradient buffer)

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, v):
        self.param_groups[0]["lr"] = v

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def _fused_conv_step(self, lib, stash, items, first):
        """Conv weights whose gradients are views of a train stash's flat buffer: update + both re-packings in ONE launch
        (subreg_sgd_pack_train).  Needs every conv weight of the backbone in this group, bound to the stash's backbone.
        Returns the items it did not handle."""
        base, hb = items[0][2], stash.hb
        off_of = {i: (p.grad.data_ptr() - base.data_ptr()) // 4 for i, p, _b in items}
        conv = [(i, p) for i, p, _b in items if p.dim() == 4 and off_of[i] in stash.conv_weight_offsets]
        if len(conv) != len(stash.conv_weight_offsets):
            return items
        for i, p in conv:
            if hb.params[stash.conv_weight_offsets[off_of[i]]].data_ptr() != p.data.data_ptr():
                return items                                   # these parameters are not the stash's backbone
        mom = self._stash_mom.get(id(hb))                      # keyed by the backbone: a new stash (another batch size) keeps its momentum
        if mom is None or mom.shape != base.shape:
            mom = self._stash_mom[id(hb)] = torch.zeros_like(base)
            if not first:                                      # earlier steps went through the per-tensor path: keep their momentum
                for i, p, _b in items:
                    mom[off_of[i]:off_of[i] + p.numel()].view_as(p).copy_(self.bufs[i])
        for i, p, _b in items:                                 # momentum buffers = views of one flat buffer laid out like the gradients
            self.bufs[i] = mom[off_of[i]:off_of[i] + p.numel()].view_as(p)
        hb._bind_pointers()
        _lib.check(lib.subreg_sgd_pack_train(C.byref(hb._desc), C.byref(stash.desc), _lib.ptr(base), _lib.ptr(stash.flat_grads), _lib.ptr(mom),
                                             self.lr, self.momentum, self.weight_decay, int(first), _lib.stream_ptr()), "sgd_pack_train")
        # the raw forward copies are shared by every stash of this backbone, the dX copies are per stash: only THIS stash's are
        # current now (tensor._version does not see the library's raw-pointer update, so the others must be told)
        for other in hb.__dict__.get("_train_stashes", []):
            other.opt_packed = None
        stash.opt_packed = conv_weight_versions(hb)
        done = {i for i, _p in conv}
        return [it for it in items if it[0] not in done]

    def step(self):
        """One fused launch for all parameters whose gradients are views of ONE flat buffer (what BackboneTrainFn.backward
        returns for the backbone: 66 tensors), the per-tensor kernel for the rest (the classifier)."""
        lib = _lib.load()
        live = [(i, p) for i, p in enumerate(self.params) if p.grad is not None]
        first_any = [self.bufs[i] is None for i, _p in live]
        for i, p in live:
            if self.bufs[i] is None:
                self.bufs[i] = torch.empty_like(p.data)
        groups = {}
        for (i, p), first in zip(live, first_any):
            g = p.grad
            base = g._base if (g._base is not None and g.is_contiguous() and g.dtype == torch.float32) else None
            groups.setdefault((id(base) if base is not None else None, first), []).append((i, p, base))
        for (bid, first), items in groups.items():
            stash = getattr(items[0][2], "_subreg_stash", None) if bid is not None else None
            if stash is not None:
                items = self._fused_conv_step(lib, stash, items, first)      # conv weights done; BN affine parameters remain
                if not items:
                    continue
            if bid is None or len(items) < 4:
                for i, p, _b in items:
                    g = p.grad.contiguous()
                    _lib.check(lib.subreg_sgd_momentum(_lib.ptr(p.data), _lib.ptr(g), _lib.ptr(self.bufs[i]), p.numel(), self.lr,
                                                       self.momentum, self.weight_decay, int(first), _lib.stream_ptr()), "sgd_momentum")
                continue
            base = items[0][2]
            key = tuple(i for i, _p, _b in items)
            cache = self._multi.get(key)
            if cache is None or cache["ptrs"] != [p.data.data_ptr() for _i, p, _b in items]:
                dev = base.device
                sizes = [p.numel() for _i, p, _b in items]
                cache = {"ptrs": [p.data.data_ptr() for _i, p, _b in items],
                         "params": torch.tensor([p.data.data_ptr() for _i, p, _b in items], dtype=torch.int64, device=dev),
                         "bufs": torch.tensor([self.bufs[i].data_ptr() for i, _p, _b in items], dtype=torch.int64, device=dev),
                         "ends": torch.tensor(np.cumsum(sizes), dtype=torch.int64, device=dev), "total": int(sum(sizes))}
                self._multi[key] = cache
            offs = [(p.grad.data_ptr() - base.data_ptr()) // 4 for _i, p, _b in items]
            if cache.get("offs_host") != offs:
                cache["offs_host"] = offs
                cache["offs"] = torch.tensor(offs, dtype=torch.int64, device=base.device)
            _lib.check(lib.subreg_sgd_momentum_multi(_lib.ptr(cache["params"]), _lib.ptr(cache["bufs"]), _lib.ptr(base),
                                                     _lib.ptr(cache["offs"]), _lib.ptr(cache["ends"]), len(items), cache["total"],
                                                     self.lr, self.momentum, self.weight_decay, int(first), _lib.stream_ptr()),
                       "sgd_momentum_multi")


class Adam:
    """torch.optim.Adam(lr, weight_decay) semantics (train_supervised.py:128-131, `--adam`) on the HIP kernel: one launch per
    parameter tensor (the option is outside every script's path; the conv weights' packed copies are rebuilt by the next
    train-mode forward, as after any update the fused SGD step did not make)."""

    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.weight_decay, self.betas, self.eps = weight_decay, betas, eps
        self.param_groups = [{"params": self.params, "lr": lr}]      # adjust_learning_rate (util.py:45-51) writes param_groups[i]['lr']
        self.state = [None] * len(self.params)                       # (exp_avg, exp_avg_sq, step)

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        lib = _lib.load()
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            if self.state[i] is None:
                self.state[i] = [torch.zeros_like(p.data), torch.zeros_like(p.data), 0]
            st = self.state[i]
            st[2] += 1
            g = p.grad.contiguous().float()
            _lib.check(lib.subreg_adam(_lib.ptr(p.data), _lib.ptr(g), _lib.ptr(st[0]), _lib.ptr(st[1]), p.numel(), self.lr,
                                       self.betas[0], self.betas[1], self.eps, self.weight_decay, st[2], _lib.stream_ptr()), "adam")


class GraphedStep:
    """One pretraining step - forward, loss, zero_grad, backward, optimiser step (train_supervised.py:229-244) - captured as ONE
    hipGraph per (input shapes, learning rate) and replayed: ~220 kernel launches on two streams become one graph launch (the eager
    step's host enqueue is 2.7 ms of a 3.9 ms step at batch 64, profiles/r05_train_step.txt).

        step = GraphedStep(model, optimizer, lambda x, y: criterion(model(x), y))
        loss = step(x, y)            # a 0-dim tensor owned by the graph: read it (.item()) before the next call

    `loss_fn(*inputs)` runs the model and returns the loss (or a tuple whose first element is the loss: the rest - logits,
    accuracy counters - is returned with it; all are graph-owned).  What a replay does beside the graph launch: copy the inputs into
    the graph's buffers, advance the host-side forward counters, and hand the masks' seeds and DropBlock's step-dependent gamma
    (resnet_language.py:294-296) to the device (HipBackbone.set_mask_params: the captured mask kernels read them when they run).

    The first `warmup` calls of a key run eagerly (they size the stash, create the optimiser's state and caches, and put the
    fused optimiser step on its steady-state path); the next one captures.  Anything that fails to capture falls back to the
    eager step for that key, with a RuntimeWarning - the eager step runs the same kernels.
    A data-parallel run (HipBackbone.grad_stage_hook set) and injected mask sources stay eager.
    """

    def __init__(self, model, optimizer, loss_fn, warmup=2, max_graphs=4):
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.warmup, self.max_graphs = max(int(warmup), 2), max_graphs
        self.entries = {}                      # key -> dict(calls, graph, inputs, outputs, failed)
        self.replays = 0

    def _key(self, inputs):
        groups = tuple((g["lr"],) for g in self.opt.param_groups)
        hyper = (getattr(self.opt, "momentum", None), getattr(self.opt, "weight_decay", None), self.model.training)
        return tuple((tuple(t.shape), t.dtype) for t in inputs) + groups + hyper

    def _eager(self, inputs):
        """One step.  How the gradients are taken is dictated by what a capture survives (tools/graph_probe.py, PROBE_HOLD=1):
        a leaf's AccumulateGrad node is shared with every older autograd graph that is still alive (a loss the caller kept from an
        eager step) and belongs to the stream of the forward that created it - the default stream.  A gradient that reaches such a
        node inside a capture makes the engine synchronise the capturing stream with the default stream, which pulls the default
        stream into the capture, and hipStreamEndCapture takes the process down (the reason torch's own recipe warms up on a side
        stream - which does not help against graphs the CALLER holds).  So:
          * parameters whose gradients travel through autograd (the classifier; anything but the backbone) are replaced, for the
            duration of the step, by fresh leaf Parameters over the SAME storage: their gradient edges are new;
          * the gradients are taken with torch.autograd.grad (no AccumulateGrad nodes run) and assigned to the real parameters;
          * the backbone's gradients do not travel through autograd at all (BackboneTrainFn.backward assigns the views of its flat
            buffer to `.grad` and returns None: allow_unused);
          * the results are handed out detached, so this step's graph dies here."""
        opt_params = [p for g in self.opt.param_groups for p in g["params"] if p.requires_grad]
        wanted = {id(p) for p in opt_params}
        backbone = {id(p) for n, p in self.model.named_parameters() if not n.startswith("classifier")}
        swaps = []
        for mod in self.model.modules():
            for name, p in list(mod._parameters.items()):
                if p is not None and id(p) in wanted and id(p) not in backbone:
                    alias = torch.nn.Parameter(p.detach(), requires_grad=True)
                    mod._parameters[name] = alias
                    swaps.append((mod, name, p, alias))
        try:
            out = self.loss_fn(*inputs)
            loss = out[0] if isinstance(out, tuple) else out
            self.opt.zero_grad()
            alias_of = {id(p): a for _m, _n, p, a in swaps}
            leaves = [alias_of.get(id(p), p) for p in opt_params]
            grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        finally:
            for mod, name, p, _alias in swaps:
                mod._parameters[name] = p
        for p, g in zip(opt_params, grads):
            if g is not None:
                p.grad = g
        self.opt.step()
        if isinstance(out, tuple):
            return tuple(t.detach() if isinstance(t, torch.Tensor) else t for t in out)
        return out.detach()

    def __call__(self, *inputs):
        hb = self.model.hip_backbone()
        key = self._key(inputs)
        ent = self.entries.get(key)
        if ent is None:
            if len(self.entries) >= self.max_graphs:                     # a new learning rate / shape: the oldest entry goes
                oldest = next(iter(self.entries))
                if self.entries[oldest]["graph"] is not None:            # (its last replay may still be in flight; an entry that never
                    torch.cuda.synchronize()                             #  captured - a learning rate that changes every step - costs nothing)
                self.entries.pop(oldest)
            ent = self.entries[key] = dict(calls=0, graph=None, inputs=None, outputs=None, failed=False)
        ent["calls"] += 1
        eager_only = (ent["failed"] or getattr(hb, "grad_stage_hook", None) is not None or self.model.mask_source is not None
                      or not torch.is_grad_enabled())
        if eager_only or (ent["graph"] is None and ent["calls"] <= self.warmup):
            return self._eager(inputs)
        x = inputs[0]
        H, W = int(x.shape[2]), int(x.shape[3])
        if ent["graph"] is None:
            self._capture(ent, hb, inputs, H, W)
            if ent["failed"]:
                return self._eager(inputs)
        # ---- replay
        for dst, src in zip(ent["inputs"], inputs):
            dst.copy_(src, non_blocking=True)
        stash = ent["stash"]
        from . import _lib
        if stash.opt_packed is None or stash.opt_packed != conv_weight_versions(hb):
            # somebody else moved the weights since this graph's last step (an eager step of another batch shape, a checkpoint
            # load): the captured forward reads the packed copies as they are, so rebuild them first
            hb._bind_pointers()
            _lib.check(hb.lib.subreg_backbone_pack_train(C.byref(hb._desc), C.byref(stash.desc), _lib.stream_ptr()), "backbone_pack_train")
        if self.model.training:
            for i in range(len(hb.nbt)):
                hb.nbt[i] += 1
            hb.set_mask_params(H, W)
        ent["graph"].replay()
        self.replays += 1
        hb._fold_versions = None                                           # running statistics and weights moved
        for other in hb.__dict__.get("_train_stashes", []):
            other.opt_packed = None
        stash.opt_packed = conv_weight_versions(hb)                        # (what the captured optimiser step establishes)
        return ent["outputs"]

    def _capture(self, ent, hb, inputs, H, W):
        import warnings
        ent["inputs"] = [torch.empty_like(t, memory_format=torch.contiguous_format) for t in inputs]
        for dst, src in zip(ent["inputs"], inputs):
            dst.copy_(src)
        nbt_keep = list(hb.nbt)
        training = self.model.training
        hb.use_mask_params = True
        try:
            if training:
                # (creates the device records; the replay that follows the capture draws this step's seeds itself, from the same
                # host generator state an eager step would have seen)
                rng = torch.get_rng_state()
                hb.set_mask_params(H, W)
                torch.set_rng_state(rng)
            torch.cuda.synchronize()
            self.opt.zero_grad()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self._eager(ent["inputs"])
            ent["graph"], ent["outputs"], ent["stash"] = g, out, hb._train_stash
        except Exception as e:                                            # noqa: BLE001 - any capture failure: run eagerly instead
            ent["failed"], ent["graph"] = True, None
            torch.cuda.synchronize()
            warnings.warn("subreg_hip: the training step could not be captured as a hipGraph (%s: %s); running it eagerly"
                          % (type(e).__name__, e), RuntimeWarning, stacklevel=3)
        finally:
            hb.use_mask_params = False
            hb.nbt = nbt_keep                                             # (the capture pass launched nothing)
