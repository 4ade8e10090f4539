"""Host-side owner of one backbone on one GPU: packed weights, folded BN, workspaces.

Drives `subreg_backbone_forward` (csrc/backbone.hip), i.e. ResNet.forward up to `feat`
(/root/reference/models/resnet_language.py:170-182) over BasicBlock.forward (:268-301).
All tensors are torch CUDA tensors used as plain device memory; torch does no math here.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .synthetic import backbone_blocks

DROP_RATE = 0.1     # models/util.py:15-18
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def dropblock_gamma(num_batches_tracked, feat_size, block_size, drop_rate=DROP_RATE):
    """BasicBlock.forward :294-296."""
    keep_rate = max(1.0 - drop_rate / (20 * 2000) * num_batches_tracked, 1.0 - drop_rate)
    return (1 - keep_rate) / block_size ** 2 * feat_size ** 2 / (feat_size - block_size + 1) ** 2


class HipBackbone:
    """params: dict with the reference's state_dict key names -> LIVE fp32 CUDA tensors (conv weights, BN
    weight/bias/running_mean/running_var).  BN running stats are updated in place by train-mode forwards."""

    MAX_EVAL_CHUNK = 1536     # images per launch sequence: one launch for a whole epoch's batch (<= 1125 images at 8 sessions) - larger grids
                              # waste less on partial last rounds (+8 % episodes/s over 512); 6.3 GB of workspaces in bf16
    EVAL_LANES = int(os.environ.get("SUBREG_EVAL_LANES", "2"))   # (the environment override is for profiling runs: with one lane the
                              # per-kernel durations of a rocprofv3 trace are not inflated by the lanes' overlap)
                              # eval-mode forwards: the batch is cut into this many sub-batches (images are independent in eval mode),
                              # each running its own 22-conv launch sequence on its own HIP stream and workspaces, forked from and
                              # joined to the caller's stream once per forward.  While one lane's kernel drains its partial last round
                              # of workgroups, the other lanes' kernels fill the idle CUs (tools/bench_forward.py, graph replay,
                              # 2 lanes vs 1: +5.8 % at 250 images, +5.3 % at 500, +4.7 % at 750, +2.6 % at 1125; 3-4 lanes: less).

    EVAL_LANE_MIN = int(os.environ.get("SUBREG_EVAL_LANE_MIN", "64"))   # images per lane below which a forward is not split

    def __init__(self, params, n_blocks=(1, 1, 2, 2), dtype="bf16", block_size=1):
        self.lib = _lib.load()
        self.params = params
        self.dtype = _lib.dtype_code(dtype)
        self.tdtype = torch.bfloat16 if self.dtype == _lib.BF16 else torch.float32
        self.block_size = block_size
        self.blocks = backbone_blocks(n_blocks)
        self.device = params[self.blocks[0][0] + ".conv1.weight"].device
        assert self.device.type == "cuda", "HipBackbone needs CUDA (HIP) tensors"
        self.nbt = [0] * len(self.blocks)          # BasicBlock.num_batches_tracked (:260,269): counts EVERY forward
        self.out_dim = self.blocks[-1][2]
        self._packed, self._scale, self._shift = {}, {}, {}
        self._versions = None
        self._fold_versions = None
        self._fold_token = 0           # bumped whenever the packed / folded weights are rebuilt (invalidates cached graphs)
        self._ws_token = 0             # ... whenever a workspace is re-allocated
        self._graphs = {}              # ((B, 3, H, W), buffer set) -> dict(calls, graph, x, feat, tokens): forward_graphed()
        self._spec_pending, self._spec_next, self._spec_last, self._spec_streams = [], {}, None, []   # input-sequence prefetch
        self.prefetch_hits = 0
        self._cap = (0, 0, 0)          # allocated (workspace bytes, stats floats, im2col elements)
        self._ws_ok = set()            # (B, H, W) known to fit the allocation
        self._lanes = []               # extra eval lanes: dict(desc, ws, col, cap, stream, ok)
        self._keep = []
        self._blk = (_lib.BlockDesc * len(self.blocks))()
        self._desc = _lib.BackboneDesc()
        self._desc.n_blocks = len(self.blocks)
        self._desc.blocks = C.cast(self._blk, C.POINTER(_lib.BlockDesc))
        self._desc.dtype = self.dtype
        self._desc.bn_eps, self._desc.bn_momentum = BN_EPS, BN_MOMENTUM
        self._alloc_static()
        self.refresh(force=True)

    # ------------------------------------------------------------------ descriptors
    def _convs(self):
        for bi, (name, cin, cout, _stride, ds, _db) in enumerate(self.blocks):
            first = bi == 0 and cin == 3
            yield bi, "conv1", name + ".conv1", name + ".bn1", cin, cout, 3, first
            yield bi, "conv2", name + ".conv2", name + ".bn2", cout, cout, 3, False
            yield bi, "conv3", name + ".conv3", name + ".bn3", cout, cout, 3, False
            if ds:
                yield bi, "down", name + ".downsample.0", name + ".downsample.1", cin, cout, 1, first

    def _alloc_static(self):
        dev = self.device
        self._folded, self._ident, self._shift3 = {}, {}, []
        for bi, slot, cname, bname, cin, cout, k, first in self._convs():
            kin, kk = (32, 1) if first else (cin, k)
            self._packed[cname] = torch.empty(cout * kk * kk * kin, dtype=self.tdtype, device=dev)   # raw (train mode)
            self._folded[cname] = torch.empty(cout * kk * kk * kin, dtype=self.tdtype, device=dev)   # * BN scale (eval)
            self._scale[cname] = torch.empty(cout, dtype=torch.float32, device=dev)
            self._shift[cname] = torch.empty(cout, dtype=torch.float32, device=dev)
            cd = getattr(self._blk[bi], slot)
            cd.cin, cd.cout, cd.ksize, cd.cin_raw, cd.ksize_raw = kin, cout, kk, cin, k
        s = _lib.stream_ptr()
        for bi, (name, _cin, cout, stride, ds, _db) in enumerate(self.blocks):
            self._blk[bi].stride = stride
            self._blk[bi].keep_mask = None
            self._blk[bi].mask_scale = 1.0
            self._shift3.append(torch.empty(cout, dtype=torch.float32, device=dev))
            self._blk[bi].shift3 = self._shift3[-1].data_ptr()
            self._blk[bi].w_identity = None
            if not ds:                         # identity shortcut: accumulated as a GEMM with I (resnet_language.py:271,288)
                if cout not in self._ident:
                    self._ident[cout] = torch.empty(cout * cout, dtype=self.tdtype, device=dev)
                    _lib.check(self.lib.subreg_pack_identity(_lib.ptr(self._ident[cout]), cout, self.dtype, s), "pack_identity")
                self._blk[bi].w_identity = self._ident[cout].data_ptr()

    def _bind_pointers(self):
        p = self.params
        for bi, slot, cname, bname, cin, cout, k, first in self._convs():
            cd = getattr(self._blk[bi], slot)
            w = p[cname + ".weight"]
            assert w.dtype == torch.float32 and w.is_contiguous()
            cd.w, cd.w_folded, cd.w_oihw = self._packed[cname].data_ptr(), self._folded[cname].data_ptr(), w.data_ptr()
            cd.bn_weight, cd.bn_bias = p[bname + ".weight"].data_ptr(), p[bname + ".bias"].data_ptr()
            cd.running_mean, cd.running_var = p[bname + ".running_mean"].data_ptr(), p[bname + ".running_var"].data_ptr()
            cd.scale, cd.shift = self._scale[cname].data_ptr(), self._shift[cname].data_ptr()

    def refresh(self, force=False):
        """Re-pack the conv weights (raw + BN-scale-folded) and re-fold BN if the module's tensors changed
        (tensor._version / data_ptr)."""
        p = self.params
        ver = tuple((p[c + ".weight"].data_ptr(), p[c + ".weight"]._version) for _, _, c, *_ in self._convs()) + \
            tuple((p[b + s].data_ptr(), p[b + s]._version) for _, _, _, b, *_ in self._convs()
                  for s in (".weight", ".bias", ".running_mean", ".running_var"))
        if force or ver != self._fold_versions:
            self._bind_pointers()
            _lib.check(self.lib.subreg_backbone_fold(C.byref(self._desc), _lib.stream_ptr()), "backbone_fold")
            self._fold_versions = ver
            self._fold_token += 1

    def refresh_raw(self):
        """After an optimiser step in training: re-pack only the raw conv weights (all a train-mode forward reads); the
        BN-folded copies are left stale and rebuilt by the next refresh()."""
        self._bind_pointers()
        _lib.check(self.lib.subreg_backbone_pack_raw(C.byref(self._desc), _lib.stream_ptr()), "backbone_pack_raw")
        self._fold_versions = None
        self._fold_token += 1

    def _col_elems(self, B, H, W, need_col):
        """Elements of the first layer's im2col buffer this forward needs: none when layer 1 reads the fp32 image itself
        (eval mode, bf16, 84x84-class images: csrc/conv_first.hip + the image-fed shortcut of csrc/conv64_resident.hip)."""
        if not need_col and not self.lib.subreg_backbone_needs_col(C.byref(self._desc), B, H, W, 0):
            return 0            # (the library's own predicate: shape support AND a first block of the layer1.0 form)
        return B * H * W * 32

    def _ensure_workspace(self, B, H, W, need_col=True):
        """Workspaces for a forward of B images.  Neither size is monotone in B (the BN-partial row count follows the
        tile height the conv picks for the batch: a smaller batch can need MORE partial rows), so capacity is tracked in
        bytes / floats as the library reports them for exactly this (B, H, W), never inferred from a larger batch.
        need_col=False: an eval-mode forward (the im2col buffer is allocated only where the direct first layer does not apply)."""
        if (B, H, W, need_col) in self._ws_ok:
            return
        dev = self.device
        nbytes = self.lib.subreg_backbone_ws_bytes(C.byref(self._desc), B, H, W)
        nstats = self.lib.subreg_backbone_stats_floats(C.byref(self._desc), B, H, W)
        ncol = self._col_elems(B, H, W, need_col)
        assert nbytes > 0 and nstats > 0
        cap_b, cap_s, cap_c = self._cap
        if nbytes > cap_b or nstats > cap_s:
            self._ws_token += 1
            cap_b, cap_s = max(cap_b, nbytes), max(cap_s, nstats)
            self._ws = [torch.empty(cap_b, dtype=torch.uint8, device=dev) for _ in range(4)]
            self._stats = torch.empty(cap_s, dtype=torch.float32, device=dev)
            for i in range(4):
                self._desc.ws[i] = self._ws[i].data_ptr()
            self._desc.stats = self._stats.data_ptr()
        if ncol > cap_c:
            self._ws_token += 1
            cap_c = ncol
            self._col = torch.empty(cap_c, dtype=self.tdtype, device=dev)
            self._desc.col = self._col.data_ptr()
        self._cap = (cap_b, cap_s, cap_c)
        self._ws_ok.add((B, H, W, need_col))

    def _lane(self, i, B, H, W):
        """Descriptor of eval lane i >= 1 (lane 0 is self._desc): same packed weights, own workspaces and stream."""
        while len(self._lanes) < i:
            d = _lib.BackboneDesc()
            C.memmove(C.byref(d), C.byref(self._desc), C.sizeof(d))
            self._lanes.append(dict(desc=d, cap=(0, 0), ok=set(), stream=torch.cuda.Stream(device=self.device)))
        ln = self._lanes[i - 1]
        ln["desc"].blocks, ln["desc"].n_blocks = self._desc.blocks, self._desc.n_blocks
        if (B, H, W) not in ln["ok"]:
            nbytes = self.lib.subreg_backbone_ws_bytes(C.byref(self._desc), B, H, W)
            ncol = self._col_elems(B, H, W, False)                   # lanes run eval-mode forwards only
            cb, cc = ln["cap"]
            if nbytes > cb:
                if cb > 0:
                    self._ws_token += 1                # (a lane's FIRST workspaces move nothing a cached graph points at)
                cb = nbytes
                ln["ws"] = [torch.empty(cb, dtype=torch.uint8, device=self.device) for _ in range(4)]
                for k in range(4):
                    ln["desc"].ws[k] = ln["ws"][k].data_ptr()
            if ncol > cc:
                if cc > 0:
                    self._ws_token += 1
                cc = ncol
                ln["col"] = torch.empty(cc, dtype=self.tdtype, device=self.device)
                ln["desc"].col = ln["col"].data_ptr()
            ln["desc"].stats = None
            ln["cap"] = (cb, cc)
            ln["ok"].add((B, H, W))
        return ln

    def release(self):
        """Drop the workspaces, lane streams and train stash (tests / long-lived processes that are done with this backbone).
        The next forward re-allocates what it needs."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self._lanes, self._ws_ok, self._cap, self._keep = [], set(), (0, 0, 0), []
        self._graphs = {}
        self._spec_pending, self._spec_next, self._spec_last = [], {}, None
        self._ws_token += 1
        self._ws = self._col = self._stats = None
        self._train_stash = None
        self.__dict__.pop("_train_stashes", None)
        for i in range(4):
            self._desc.ws[i] = None
        self._desc.col = self._desc.stats = None

    # ------------------------------------------------------------------ train-mode masks
    def _clear_masks(self):
        """No dropout / DropBlock (an eval-mode forward that keeps a stash): every block's keep mask off."""
        self._keep = []
        for bi in range(len(self.blocks)):
            self._blk[bi].keep_mask = None
            self._blk[bi].mask_scale = 1.0
            self._blk[bi].mask_scale_dev = None

    def mask_params(self, H, W):
        """(seed, drop probability) of every block's free-running keep mask for the NEXT forward's counters (self.nbt as that forward
        sees it): dropout's rate (:299), DropBlock's gamma (:294-296) or, for block_size > 1, the complement of the seeds' rate."""
        out, h = [], H
        for bi, (_name, _cin, _cout, stride, _ds, db) in enumerate(self.blocks):
            h = h // stride
            seed = int(torch.randint(0, 2 ** 62, (1,)))
            if not db:
                out.append((seed, DROP_RATE))
            else:
                gamma = float(dropblock_gamma(self.nbt[bi], h, self.block_size))
                out.append((seed, gamma if self.block_size == 1 else 1.0 - gamma))
        return out

    def set_mask_params(self, H, W):
        """Device-resident mask parameters (graph replay, train.GraphedStep): refresh them for the forward that follows."""
        n = len(self.blocks)
        if getattr(self, "_mask_params_dev", None) is None:
            self._mask_params_dev = torch.zeros(2 * n, dtype=torch.int64, device=self.device)     # n records of 16 bytes
        vals = (_lib.MaskParam * n)()
        for i, (seed, p) in enumerate(self.mask_params(H, W)):
            vals[i].seed, vals[i].p_drop = seed, min(max(p, 0.0), 0.999999)
        _lib.check(self.lib.subreg_mask_params_set(_lib.ptr(self._mask_params_dev), n, C.cast(vals, C.c_void_p), _lib.stream_ptr()),
                   "mask_params_set")

    def _prepare_masks(self, B, H, W, masks):
        """Keep masks of every block output (dropout :299 / DropBlock :311-325), uploaded as NHWC u8."""
        self._keep = []
        s = _lib.stream_ptr()
        h, w = H, W
        # device-resident seeds / probabilities (self.use_mask_params, set by train.GraphedStep around its capture): the launches
        # below then read them when they RUN, so a replayed graph draws fresh masks
        pdev = getattr(self, "_mask_params_dev", None) if (getattr(self, "use_mask_params", False) and masks is None) else None

        def random_keep(dst, n, bi, p, count):
            if pdev is not None:
                _lib.check(self.lib.subreg_random_keep_mask_dev(_lib.ptr(dst), n, pdev.data_ptr() + 16 * bi, count, s), "random_keep_mask_dev")
            else:
                _lib.check(self.lib.subreg_random_keep_mask(_lib.ptr(dst), n, int(torch.randint(0, 2 ** 62, (1,))), p, count, s),
                           "random_keep_mask")
        # DropBlock's count_ones and the rescale factor numel / count_ones stay on the device (one counter and one float per
        # block), as in the reference (:318-323): reading them back cost two host synchronisations per training step
        counts = torch.zeros(len(self.blocks), dtype=torch.int32, device=self.device)
        self._mask_scale_dev = torch.ones(len(self.blocks), dtype=torch.float32, device=self.device)
        self._mask_counts = counts

        def device_scale(bi, n):
            if os.environ.get("SUBREG_MASK_HOST_COUNT") == "1":      # A/B switch: the earlier host read of the counter
                return n / max(int(counts[bi].item()), 1)
            _lib.check(self.lib.subreg_mask_scale(_lib.ptr(counts[bi:]), n, _lib.ptr(self._mask_scale_dev[bi:]), s), "mask_scale")
            self._blk[bi].mask_scale_dev = self._mask_scale_dev[bi:].data_ptr()
            return 0.0                             # the host-side field is not read when mask_scale_dev is set

        for bi, (name, _cin, cout, stride, _ds, db) in enumerate(self.blocks):
            h, w = h // stride, w // stride
            n = B * cout * h * w
            keep = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._blk[bi].mask_scale_dev = None
            if not db:
                scale = float(np.float32(1.0) / np.float32(1.0 - DROP_RATE))
                if masks is None:
                    random_keep(keep, n, bi, DROP_RATE, None)
                else:
                    m = torch.from_numpy(masks.dropout_keep((B, cout, h, w), DROP_RATE)).to(self.device)
                    _lib.check(self.lib.subreg_mask_nchw_to_nhwc(_lib.ptr(m), _lib.ptr(keep), B, cout, h, w, 0, s), "mask")
            else:
                bs = self.block_size
                gamma = dropblock_gamma(self.nbt[bi], h, bs)
                shape = (B, cout, h - (bs - 1), w - (bs - 1))
                if masks is None and bs == 1:
                    random_keep(keep, n, bi, float(gamma), _lib.ptr(counts[bi:]))
                    scale = device_scale(bi, n)
                elif bs == 1:                                   # injected element mask (block_size 1)
                    bm = 1.0 - masks.bernoulli(shape, gamma)
                    scale = bm.size / bm.sum()
                    m = torch.from_numpy(np.ascontiguousarray(bm, dtype=np.float32)).to(self.device)
                    _lib.check(self.lib.subreg_mask_nchw_to_nhwc(_lib.ptr(m), _lib.ptr(keep), B, cout, h, w, 0, s), "mask")
                else:
                    # block_size > 1 (no --no_dropblock): the Bernoulli seeds come from the injected source or from the device
                    # generator; DropBlock._compute_block_mask (:327-357) runs on the device
                    ns = int(np.prod(shape))
                    if masks is not None:
                        sample = torch.from_numpy(np.ascontiguousarray(masks.bernoulli(shape, gamma) != 0).astype(np.uint8)).to(self.device)
                    else:
                        sample = torch.empty(ns, dtype=torch.uint8, device=self.device)
                        random_keep(sample, ns, bi, float(1.0 - gamma), None)   # 1 with probability gamma
                    _lib.check(self.lib.subreg_dropblock_mask(_lib.ptr(sample), _lib.ptr(keep), B, cout, h, w, bs, _lib.ptr(counts[bi:]), s),
                               "dropblock_mask")
                    scale = device_scale(bi, n)
            self._keep.append(keep)
            self._blk[bi].keep_mask = keep.data_ptr()
            self._blk[bi].mask_scale = float(scale)

    def mask_scale(self, bi):
        """The rescale factor of block bi's keep mask as a host float (tests / diagnostics: synchronises when it lives on the device)."""
        if self._blk[bi].mask_scale_dev:
            return float(self._mask_scale_dev[bi].item())
        return float(self._blk[bi].mask_scale)

    # ------------------------------------------------------------------ forward
    GRAPH_EVAL = os.environ.get("SUBREG_GRAPH_EVAL", "1") != "0"      # forward_graphed(): replay cached hipGraphs (0: always eager)
    GRAPH_CACHE = 12                                                  # (shape, buffer set) entries kept (least recently used goes first)
    EVAL_PREFETCH = int(os.environ.get("SUBREG_EVAL_PREFETCH", "2"))  # forward_graphed(): forwards started ahead of their call (0: none)

    def _graph_entry(self, key):
        """Cache entry of (shape, buffer set); the least recently used entry goes when the cache is full."""
        ent = self._graphs.pop(key, None)
        if ent is None:
            ent = dict(calls=0, graph=None, x=None, feat=None, tokens=None, eager_only=False)
            while len(self._graphs) >= self.GRAPH_CACHE:                  # least recently used first (dict order = use order)
                old_key = next(iter(self._graphs))
                if self._graphs[old_key]["graph"] is not None:
                    torch.cuda.synchronize(self.device)                   # (a replay may still be in flight)
                del self._graphs[old_key]
        self._graphs[key] = ent                                           # (re-inserted: most recently used)
        return ent

    def _capture_entry(self, ent, key, x, lane_base):
        """Capture the eval forward of x's shape on workspace set lane_base into ent; False (and ent stays eager-only) on failure."""
        ent["x"] = torch.empty_like(x, memory_format=torch.contiguous_format)
        ent["feat"] = torch.empty(x.shape[0], self.out_dim, dtype=torch.float32, device=self.device)
        ent["x"].copy_(x)
        nbt_keep = list(self.nbt)
        torch.cuda.synchronize(self.device)
        try:
            g = torch.cuda.CUDAGraph()
            # thread_local: HIP calls of OTHER host threads (a DataLoader's pin_memory thread) do not invalidate this capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.forward(ent["x"], out=ent["feat"], check_params=False, lane_base=lane_base)
        except Exception as e:                                            # noqa: BLE001 - whatever broke the capture: this shape stays eager
            import warnings
            self.nbt = nbt_keep
            ent.update(graph=None, x=None, feat=None, eager_only=True)
            torch.cuda.synchronize(self.device)
            warnings.warn("subreg_hip: eval forward of shape %s could not be captured as a hipGraph (%s: %s); it runs eagerly"
                          % (key, type(e).__name__, e), RuntimeWarning, stacklevel=3)
            return False
        self.nbt = nbt_keep                                               # (the capture pass launches nothing)
        ent["graph"], ent["tokens"] = g, (self._fold_token, self._ws_token)
        return True

    def forward_graphed(self, x):
        """Eval-mode forward for callers that run the SAME shapes again and again over a backbone that does not change (the
        reference's own loop over the drop-in module, eval/language_eval.py:242-326: one support forward and one forward per
        query set and epoch, 125 images each).

        1. The launch sequence (~25 kernels on one or two lanes) is captured into a hipGraph the second time a shape is seen and
           replayed from then on - copy the input into the graph's buffer, replay, copy the features out.
        2. Input-sequence prefetch (EVAL_PREFETCH forwards deep).  That loop reads every result on the host before it asks for the
           next forward (`.item()`, `.cpu()`: language_eval.py:36-43), so its nine 125-image forwards per epoch run one after another
           with the GPU idle in between, each at the fill of a 125-image batch.  The calls repeat epoch after epoch with the SAME
           tensor objects in the same order: the backbone remembers which tensor followed which, and when a call arrives it starts
           the forwards of the next EVAL_PREFETCH tensors of that chain on side streams with workspace sets of their own.  A later
           call is served from such a forward only if it passes THE SAME tensor object at the SAME `_version` under the same
           weights (anything else - another tensor, an in-place edit, re-packed weights - drops the predictions and runs the
           forward now), so every result is the forward of exactly what was passed; nothing is cached across calls: each call is
           one execution of all 22 convolutions.  What changes is when they run: two forwards ahead overlap each other and the
           host's round trips, like the two lanes of a 250-image forward.
        The caches are dropped whenever the weights are re-packed (refresh() sees tensor._version / data_ptr changes) or a workspace
        moves."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3
        if not self.GRAPH_EVAL:
            return self.forward(x)
        self.refresh()
        shape = tuple(x.shape)
        tokens = (self._fold_token, self._ws_token)
        # every captured entry whose weights were re-packed or whose workspaces moved since its capture goes, whatever its shape (each
        # keeps a copy of its input and features alive: ~95 MB at 1125 images).  A replay may still be in flight on the caller's or
        # a lane's stream: synchronise before a graph is destroyed, as IncrementalRunner does before graph.reset().
        stale = [k for k, e in self._graphs.items() if e["graph"] is not None and e["tokens"] != tokens]
        if stale:
            torch.cuda.synchronize(self.device)
            for k in stale:
                del self._graphs[k]
            self._spec_pending = []
        cur = torch.cuda.current_stream()
        feat = None
        pend = self._spec_pending
        if pend:
            p = pend[0]
            if p["x"] is x and p["version"] == x._version and p["tokens"] == tokens:
                pend.pop(0)
                cur.wait_event(p["done"])                                 # that forward is complete before the copy below reads its output
                feat = p["ent"]["feat"].clone()
                self.prefetch_hits += 1
            else:                                                         # not what was predicted: every prediction goes
                self._spec_pending = pend = []
        if feat is None:
            ent = self._graph_entry((shape, 0))
            ent["calls"] += 1
            if ent["eager_only"] or (ent["graph"] is None and ent["calls"] < 2):
                feat = self.forward(x, check_params=False)                # first sight of a shape: eager (sizes the workspaces)
                self.nbt = [n - 1 for n in self.nbt]                      # (counted below, like every call)
            elif ent["graph"] is None and not self._capture_entry(ent, shape, x, 0):
                feat = self.forward(x, check_params=False)
                self.nbt = [n - 1 for n in self.nbt]
            else:
                if ent["x"].data_ptr() != x.data_ptr():
                    ent["x"].copy_(x)
                ent["graph"].replay()
                feat = ent["feat"].clone()
        for i in range(len(self.nbt)):
            self.nbt[i] += 1
        if self.EVAL_PREFETCH > 0:
            self._prefetch(x, tokens, cur)
        return feat

    def _parallel_streams(self, n):
        """n streams whose kernels really run beside each other's and beside the current stream's.  HIP deals streams onto a few hardware
        queues in creation order (and by priority), and two streams on one queue execute one after the other: in a process that had made
        other streams before, the two prefetch streams shared a queue and depth 2 ran like depth 1 (11.0 instead of 7.9 ms per nine
        forwards, profiles/r06_prefetch_queues.txt: which stream counts collide depends on the priority, none is safe).  So the
        streams are CHOSEN by measurement, once: a spin kernel on two candidates at a time - those that finish together in one
        spin's time run in parallel."""
        dev, cur = self.device, torch.cuda.current_stream()
        cands = [torch.cuda.Stream(device=dev, priority=-(i % 2)) for i in range(10)]
        if os.environ.get("SUBREG_PREFETCH_CALIBRATE", "1") == "0":
            return cands[:n]
        spin = 2_000_000                                               # ticks of torch.cuda._sleep (~1 ms); only ratios are used

        def timed(streams):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record(cur)
            for st in streams:
                if st is not cur:
                    st.wait_stream(cur)
                with torch.cuda.stream(st):
                    torch.cuda._sleep(spin)
            for st in streams:
                if st is not cur:
                    cur.wait_stream(st)
            e1.record(cur)
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1)
        try:
            timed([cur])
            one = min(timed([cur]), timed([cur]))
            free = [c for c in cands if min(timed([c, cur]), timed([c, cur])) < 1.3 * one]     # beside the caller's stream
            chosen = []
            for c in free:                                                                    # and beside each other
                if all(min(timed([c, o]), timed([c, o])) < 1.3 * one for o in chosen):
                    chosen.append(c)
                    if len(chosen) == n:
                        break
            self.prefetch_streams_calibrated = len(chosen)
            return chosen + [c for c in cands if c not in chosen][:n - len(chosen)]
        except Exception:                                              # noqa: BLE001 - no spin kernel on this build: creation order it is
            return cands[:n]

    def _prefetch(self, x, tokens, cur):
        """Remember that x followed the previous call's tensor; start the forwards of the tensors that followed x last time."""
        hist = self._spec_next
        if self._spec_last is not None and self._spec_last is not x:
            hist.pop(id(self._spec_last), None)
            hist[id(self._spec_last)] = (self._spec_last, x)              # (strong references: an id is only compared while its tensor lives)
            while len(hist) > 16:                                        # (each entry keeps two caller tensors alive: ~10 MB per 125 images)
                hist.pop(next(iter(hist)))
        self._spec_last = x
        pend = self._spec_pending
        t = pend[-1]["x"] if pend else x
        seen = {id(x)} | {id(p["x"]) for p in pend}
        while len(pend) < self.EVAL_PREFETCH:
            nxt = hist.get(id(t))
            if nxt is None or nxt[0] is not t or id(nxt[1]) in seen:
                return
            nx = nxt[1]
            base = self._graphs.get((tuple(nx.shape), 0))
            if base is None or base["graph"] is None or not nx.is_cuda or nx.dtype != torch.float32:
                return                                                    # (only shapes whose own graph exists: seen at least twice)
            used = {p["set"] for p in pend}
            sset = next(s for s in range(1, self.EVAL_PREFETCH + 2) if s not in used)
            if len(self._spec_streams) <= sset:
                self._spec_streams = [None] + self._parallel_streams(self.EVAL_PREFETCH + 1)
            st = self._spec_streams[sset]
            key = (tuple(nx.shape), sset)
            ent = self._graph_entry(key)
            if ent["eager_only"]:
                return
            if ent["graph"] is None and not self._capture_entry(ent, key, nx, sset * max(1, int(self.EVAL_LANES))):
                return
            st.wait_stream(cur)                                           # behind the packed weights and the last copy out of this set's buffer
            with torch.cuda.stream(st):
                ent["x"].copy_(nx)
                ent["graph"].replay()
                done = torch.cuda.Event()
                done.record(st)
            pend.append(dict(x=nx, version=nx._version, tokens=tokens, ent=ent, set=sset, done=done))
            seen.add(id(nx))
            t = nx

    def forward(self, x, train=False, masks=None, return_stages=False, out=None, check_params=True, lane_base=0):
        """x: [B,3,H,W] fp32 CUDA (NCHW like the reference) -> feat [B,out_dim] fp32.
        check_params=False skips the (host-side) scan for changed weights/BN tensors when the caller knows
        nothing changed since the last forward (the fused loop's eval epochs).
        lane_base > 0 (eval mode only): run on the workspace sets lane_base, lane_base + 1, ... instead of 0, 1, ... - a forward that
        may overlap another one of this backbone (forward_graphed's prefetch); its first lane runs on the CURRENT stream."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3
        x = x.contiguous()
        B, _, H, W = x.shape
        if check_params or self._fold_versions is None:
            self.refresh()
        for i in range(len(self.nbt)):
            self.nbt[i] += 1
        feat = out if out is not None else torch.empty(B, self.out_dim, dtype=torch.float32, device=self.device)
        stages, stage_ptrs = None, None
        if return_stages:
            stages, h, w = [], H, W
            for (_n, _ci, cout, stride, _ds, _db) in self.blocks:
                h, w = h // stride, w // stride
                stages.append(torch.empty(B, cout, h, w, dtype=torch.float32, device=self.device))
        if train or return_stages:
            chunk = B
        else:                                        # balanced chunks (1125 images -> 3 x 375, not 512 + 512 + 101)
            n_chunks = -(-B // self.MAX_EVAL_CHUNK)
            chunk = -(-B // n_chunks)
        lanes = 1 if (train or return_stages) else max(1, min(int(self.EVAL_LANES), chunk // self.EVAL_LANE_MIN))
        assert lane_base == 0 or not (train or return_stages), "workspace sets other than 0 serve eval-mode forwards only"
        if lanes > 1 or lane_base > 0:
            assert not train, "train-mode forwards write BN statistics: lanes carry no stats buffer (desc.stats = None)"
            # eval mode, several lanes: sub-batch i of every chunk goes to lane lane_base + i (its own workspaces; lane 0 of the set runs
            # on the current stream, the others on streams of their own)
            sub = -(-chunk // lanes)
            cur = torch.cuda.current_stream()
            if lane_base == 0:
                self._ensure_workspace(sub, H, W, need_col=False)
            sets = [None if lane_base + i == 0 else self._lane(lane_base + i, sub, H, W) for i in range(lanes)]
            for ln in sets[1:]:
                ln["stream"].wait_stream(cur)                       # fork: x (and anything queued before) is ready
            for b0 in range(0, B, chunk):
                nb = min(chunk, B - b0)
                for i in range(lanes):
                    lo, hi = b0 + min(i * sub, nb), b0 + min((i + 1) * sub, nb)
                    if hi <= lo:
                        continue
                    desc = self._desc if sets[i] is None else sets[i]["desc"]
                    with torch.cuda.stream(cur if i == 0 else sets[i]["stream"]):
                        _lib.check(self.lib.subreg_backbone_forward(C.byref(desc), _lib.ptr(x[lo:hi]), hi - lo, H, W,
                                                                    _lib.ptr(feat[lo:hi]), None, 0, _lib.stream_ptr()),
                                   "backbone_forward")
            for ln in sets[1:]:
                cur.wait_stream(ln["stream"])                       # join
            return feat
        self._ensure_workspace(chunk, H, W, need_col=train)
        s = _lib.stream_ptr()
        if train:
            self._prepare_masks(B, H, W, masks)
        for b0 in range(0, B, chunk):
            nb = min(chunk, B - b0)
            if stages is not None:
                stage_ptrs = (C.c_void_p * len(stages))(*[t.data_ptr() for t in stages])
            _lib.check(self.lib.subreg_backbone_forward(C.byref(self._desc), _lib.ptr(x[b0:b0 + nb]), nb, H, W,
                                                        _lib.ptr(feat[b0:b0 + nb]), stage_ptrs,
                                                        _lib.FWD_TRAIN if train else 0, s), "backbone_forward")
        if train:
            self._fold_versions = None      # running stats moved: re-fold before the next eval forward
            for bi in range(len(self.blocks)):
                self._blk[bi].keep_mask = None
        if return_stages:
            return feat, stages
        return feat
