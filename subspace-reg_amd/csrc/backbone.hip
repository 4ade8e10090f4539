// Whole-backbone forward as ONE C-ABI call: the launch sequence of ResNet.forward
// (models/resnet_language.py:170-182) over BasicBlock.forward (:268-301), eval and train
// mode, on caller-owned workspaces.  Keeping the per-layer orchestration native removes
// ~25 Python->ctypes round trips per forward and makes the whole forward capturable into
// one hipGraph by the host side.
#include <stdlib.h>
#include <string.h>

#include "subreg_common.h"

namespace {

// images per call from which layer 1's conv1 + conv2 run as the fused kernel (SUBREG_FUSED12_MIN overrides, for measurements)
int fused12_min_batch() {
    static const int v = [] { const char* e = getenv("SUBREG_FUSED12_MIN"); return e && *e ? atoi(e) : 1; }();
    return v;
}

struct Shape { int h, w; };

inline size_t elem_size(int dtype) { return dtype == SUBREG_BF16 ? 2 : 4; }

// first block consumes the K=32 im2col rows
inline bool packed_first(const subreg_backbone_desc* d) { return d->blocks[0].conv1.ksize == 1 && d->blocks[0].conv1.cin == 32; }

#define TRY(expr)                        \
    do {                                 \
        const int _rc = (expr);          \
        if (_rc != SUBREG_OK) return _rc; \
    } while (0)

// eval-mode scale/shift of one BN, then both packed copies of its conv: raw (train mode) and scale-folded (eval mode)
int fold_one(const subreg_conv_desc& c, float eps, int dtype, void* stream) {
    if (!c.w) return SUBREG_OK;
    TRY(subreg_bn_fold(c.bn_weight, c.bn_bias, c.running_mean, c.running_var, c.scale, c.shift, c.cout, eps, stream));
    const int mode = (c.cin_raw == 3) ? 1 : 0;
    TRY(subreg_pack_conv_weight(c.w_oihw, nullptr, const_cast<void*>(c.w), c.cout, c.cin_raw, c.ksize_raw, mode, dtype, stream));
    return subreg_pack_conv_weight(c.w_oihw, c.scale, const_cast<void*>(c.w_folded), c.cout, c.cin_raw, c.ksize_raw, mode, dtype,
                                   stream);
}

// train mode: raw conv + partial stats + finalize (scale/shift of THIS batch, running stats updated)
int conv_train(const subreg_backbone_desc* d, const subreg_conv_desc& c, const void* x, void* raw, int B, int H, int W,
               void* stream) {
    TRY(subreg_conv_fwd(x, c.w, raw, nullptr, nullptr, nullptr, d->stats, nullptr, nullptr, 0, B, H, W, c.cin, c.cout, c.ksize,
                        SUBREG_CONV_RAW_STATS, d->dtype, stream));
    const int rows = subreg_conv_stats_rows(d->dtype, B, H, W, c.cout);
    return subreg_bn_train_finalize(d->stats, rows, c.cout, (long long)B * H * W, c.bn_weight, c.bn_bias, c.running_mean,
                                    c.running_var, d->bn_momentum, d->bn_eps, c.scale, c.shift, nullptr, nullptr, stream);
}

}  // namespace

extern "C" int subreg_abi_version(void) { return SUBREG_ABI_VERSION; }

// Does a forward of this backbone read the first layer's im2col buffer (desc.col)?  ONE predicate for the library and its callers:
// 0 only for eval-mode forwards whose layer 1 reads the fp32 NCHW image itself (conv_first.hip + the image-fed kernels of
// conv64_resident.hip), which needs the (B, H, W, dtype) support of subreg_layer1_direct_supported AND a first block of the
// layer1.0 shape (stride 2, a 1x1 downsample branch, 64 output channels).
extern "C" int subreg_backbone_needs_col(const subreg_backbone_desc* d, int B, int H, int W, int train) {
    if (!d || d->n_blocks < 1 || !d->blocks) return 1;
    const bool direct = !train && d->blocks[0].stride == 2 && d->blocks[0].down.w && d->blocks[0].conv1.cout == 64 &&
                        subreg_layer1_direct_supported(B, H, W, d->dtype);
    return direct ? 0 : 1;
}

extern "C" const char* subreg_strerror(int code) {
    switch (code) {
        case SUBREG_OK: return "ok";
        case SUBREG_EINVAL: return "invalid argument";
        case SUBREG_EUNSUPPORTED: return "unsupported shape for the gfx950 kernels";
        case SUBREG_EHIP: return "HIP runtime call failed";
        default: return code <= -1000 ? "kernel launch failed (hipError = -(code+1000))" : "unknown error";
    }
}

extern "C" long long subreg_backbone_ws_bytes(const subreg_backbone_desc* d, int B, int H, int W) {
    if (!d || !d->blocks || d->n_blocks <= 0) return SUBREG_EINVAL;
    long long worst = (long long)B * H * W * 32;          // im2col rows share the size class of layer 1
    int h = H, w = W;
    for (int i = 0; i < d->n_blocks; ++i) {
        const long long e = (long long)B * h * w * d->blocks[i].conv1.cout;
        if (e > worst) worst = e;
        if (d->blocks[i].stride == 2) { h /= 2; w /= 2; }
    }
    return worst * (long long)elem_size(d->dtype);
}

extern "C" long long subreg_backbone_stats_floats(const subreg_backbone_desc* d, int B, int H, int W) {
    if (!d || !d->blocks || d->n_blocks <= 0) return SUBREG_EINVAL;
    long long worst = 0;
    int h = H, w = W;
    for (int i = 0; i < d->n_blocks; ++i) {
        const int c = d->blocks[i].conv1.cout;
        const long long e = (long long)subreg_conv_stats_rows(d->dtype, B, h, w, c) * c * 2;
        if (e > worst) worst = e;
        if (d->blocks[i].stride == 2) { h /= 2; w /= 2; }
    }
    return worst;
}

extern "C" int subreg_backbone_fold(const subreg_backbone_desc* d, void* stream) {
    SUBREG_CHECK_ARG(d && d->blocks && d->n_blocks > 0);
    for (int i = 0; i < d->n_blocks; ++i) {
        const subreg_block_desc& b = d->blocks[i];
        TRY(fold_one(b.conv1, d->bn_eps, d->dtype, stream));
        TRY(fold_one(b.conv2, d->bn_eps, d->dtype, stream));
        TRY(fold_one(b.conv3, d->bn_eps, d->dtype, stream));
        TRY(fold_one(b.down, d->bn_eps, d->dtype, stream));
        SUBREG_CHECK_ARG(b.shift3 != nullptr);
        if (b.down.w) TRY(subreg_vec_add(b.shift3, b.conv3.shift, b.down.shift, b.conv3.cout, stream));   // one epilogue shift
        else TRY(hipMemcpyAsync(b.shift3, b.conv3.shift, sizeof(float) * b.conv3.cout, hipMemcpyDeviceToDevice,
                                (hipStream_t)stream) == hipSuccess ? SUBREG_OK : SUBREG_EHIP);
    }
    return SUBREG_OK;
}

extern "C" int subreg_backbone_pack_raw(const subreg_backbone_desc* d, void* stream) {
    SUBREG_CHECK_ARG(d && d->blocks && d->n_blocks > 0);
    for (int i = 0; i < d->n_blocks; ++i) {
        const subreg_block_desc& b = d->blocks[i];
        const subreg_conv_desc* cs[4] = {&b.conv1, &b.conv2, &b.conv3, &b.down};
        for (const subreg_conv_desc* c : cs)
            if (c->w)
                TRY(subreg_pack_conv_weight(c->w_oihw, nullptr, const_cast<void*>(c->w), c->cout, c->cin_raw, c->ksize_raw,
                                            c->cin_raw == 3 ? 1 : 0, d->dtype, stream));
    }
    return SUBREG_OK;
}

extern "C" int subreg_backbone_forward(const subreg_backbone_desc* d, const float* x_nchw, int B, int H, int W, float* feat,
                                       float* const* stage_out, int flags, void* stream) {
    SUBREG_CHECK_ARG(d && d->blocks && d->n_blocks > 0 && x_nchw && feat && B > 0 && H > 0 && W > 0);
    SUBREG_CHECK_ARG(d->ws[0] && d->ws[1] && d->ws[2] && d->ws[3]);
    SUBREG_CHECK_ARG(packed_first(d));
    const bool train = flags & SUBREG_FWD_TRAIN;
    SUBREG_CHECK_ARG(!train || d->stats);
    const int dt = d->dtype;
    // eval mode, bf16, 84x84-class images: layer 1 reads the fp32 image itself (conv_first.hip; conv3's shortcut inside
    // conv64_resident.hip) - no im2col buffer is written or read.  Otherwise conv1 and the shortcut are K = 32 GEMMs over it.
    const bool direct = !subreg_backbone_needs_col(d, B, H, W, train ? 1 : 0);
    SUBREG_CHECK_ARG(direct || d->col);

    if (!direct) TRY(subreg_pack_input(x_nchw, d->col, B, H, W, dt, stream));
    const void* cur = direct ? nullptr : d->col;
    // K split over workgroups for the small maps of an eval-mode forward (the third free workspace slot holds the partial sums):
    // SUBREG_EVAL_SPLITK=1.  Measured (tools/bench_splitk.py, profiles/r05_eval_splitk.txt): layer 4.1's convolutions 1.3x at 125 images,
    // 1.9x at 63, nothing above ~160 - and another summation order, which moves a near-chance bf16 golden (loop_hw32_bias) from two
    // flipped query images to three; off by default for that reason.
    static const bool eval_splitk = [] { const char* e = getenv("SUBREG_EVAL_SPLITK"); return e && e[0] == '1'; }();
    const long long slot_floats = eval_splitk ? subreg_backbone_ws_bytes(d, B, H, W) / 4 : 0;   // what the caller sized every workspace slot for
    int cur_slot = -1;                 // workspace slot holding `cur` (-1: the im2col buffer)
    int h = H, w = W;
    for (int i = 0; i < d->n_blocks; ++i) {
        const subreg_block_desc& b = d->blocks[i];
        const bool pool = b.stride == 2;
        SUBREG_CHECK_ARG(b.stride == 1 || b.stride == 2);
        SUBREG_CHECK_ARG(b.down.w || b.conv1.cin == b.conv3.cout);
        // three free slots besides the one holding `cur`
        int fs[3], nf = 0;
        for (int s = 0; s < 4 && nf < 3; ++s) if (s != cur_slot) fs[nf++] = s;
        void* A = d->ws[fs[0]];
        void* Bf = d->ws[fs[1]];
        void* C = d->ws[fs[2]];
        const int pflag = pool ? SUBREG_CONV_POOL2 : 0;
        int out_slot;
        if (!train) {
            // BN scale is folded into the packed weights; conv3 accumulates the shortcut branch (1x1 conv+BN, or the
            // identity) as a second GEMM over the block input, so no separate shortcut tensor is written or re-read
            const bool img_in = direct && i == 0;
            // layer 1 from the image: conv1 + conv2 in one launch where the fused kernel takes the shape, else conv1 by itself
            int fused = SUBREG_EUNSUPPORTED;
            // (round 3's fused kernel lost to the two launches below ~190 images per call and was gated on that; since its round-4
            // rework it wins at every batch: +1-2 % on the whole forward at 8-32 images, +5-12 % at 64-100, +3-4.5 % on two-lane
            // forwards of 130-380 images; profiles/r04_forward_fused12_threshold.txt)
            if (img_in && b.conv2.cin == 64 && b.conv2.cout == 64 && B >= fused12_min_batch())
                fused = subreg_conv12_first_fused(x_nchw, b.conv1.w_folded, b.conv1.shift, b.conv2.w_folded, b.conv2.shift, Bf, B, h, w,
                                                  SUBREG_CONV_LRELU, dt, stream);
            if (fused != SUBREG_OK) {
                if (fused != SUBREG_EUNSUPPORTED) return fused;
                if (img_in) TRY(subreg_conv_first_fwd(x_nchw, b.conv1.w_folded, A, b.conv1.shift, B, h, w, b.conv1.cout, SUBREG_CONV_LRELU, dt, stream));
                // (the third free slot is the K-split workspace of the small maps: 5x5 below ~160 images, see splitk_plan in conv_fwd.hip)
                else TRY(subreg_conv_fwd_ws(cur, b.conv1.w_folded, A, nullptr, b.conv1.shift, nullptr, nullptr, nullptr, nullptr, 0, B, h, w,
                                            b.conv1.cin, b.conv1.cout, b.conv1.ksize, SUBREG_CONV_LRELU, dt, eval_splitk ? (float*)C : nullptr,
                                            slot_floats, stream));
                TRY(subreg_conv_fwd_ws(A, b.conv2.w_folded, Bf, nullptr, b.conv2.shift, nullptr, nullptr, nullptr, nullptr, 0, B, h, w,
                                       b.conv2.cin, b.conv2.cout, b.conv2.ksize, SUBREG_CONV_LRELU, dt, eval_splitk ? (float*)C : nullptr,
                                       slot_floats, stream));
            }
            const void* w2 = b.down.w ? b.down.w_folded : b.w_identity;
            const int cin2 = b.down.w ? b.down.cin : b.conv3.cout;
            SUBREG_CHECK_ARG(w2 != nullptr);
            if (img_in) TRY(subreg_conv_fwd_image_shortcut(Bf, b.conv3.w_folded, A, b.shift3, x_nchw, w2, B, h, w, b.conv3.cin,
                                                           b.conv3.cout, SUBREG_CONV_LRELU | pflag, dt, stream));
            else if (eval_splitk && !b.down.w && !pool && subreg_conv_splitk_floats(B, h, w, b.conv3.cin, b.conv3.cout, b.conv3.ksize, dt) > 0)
                // identity shortcut on a map small enough for the K split: the block input is added in the reduce pass instead of
                // riding along as a second GEMM
                TRY(subreg_conv_fwd_ws(Bf, b.conv3.w_folded, A, nullptr, b.shift3, cur, nullptr, nullptr, nullptr, 0, B, h, w, b.conv3.cin,
                                       b.conv3.cout, b.conv3.ksize, SUBREG_CONV_LRELU, dt, (float*)C, slot_floats, stream));
            else TRY(subreg_conv_fwd(Bf, b.conv3.w_folded, A, nullptr, b.shift3, nullptr, nullptr, cur, w2, cin2, B, h, w,
                                     b.conv3.cin, b.conv3.cout, b.conv3.ksize, SUBREG_CONV_LRELU | pflag, dt, stream));
            out_slot = fs[0];
        } else {
            TRY(conv_train(d, b.conv1, cur, A, B, h, w, stream));
            TRY(subreg_bn_apply(A, b.conv1.scale, b.conv1.shift, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, A, B, h, w,
                                b.conv1.cout, SUBREG_CONV_LRELU, dt, stream));
            TRY(conv_train(d, b.conv2, A, Bf, B, h, w, stream));
            TRY(subreg_bn_apply(Bf, b.conv2.scale, b.conv2.shift, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, Bf, B, h, w,
                                b.conv2.cout, SUBREG_CONV_LRELU, dt, stream));
            TRY(conv_train(d, b.conv3, Bf, A, B, h, w, stream));
            const void* res = cur;
            const float *rsc = nullptr, *rsh = nullptr;
            if (b.down.w) {
                TRY(conv_train(d, b.down, cur, C, B, h, w, stream));
                res = C; rsc = b.down.scale; rsh = b.down.shift;
            }
            TRY(subreg_bn_apply(A, b.conv3.scale, b.conv3.shift, res, rsc, rsh, b.keep_mask, b.mask_scale, b.mask_scale_dev, Bf, B, h, w,
                                b.conv3.cout, SUBREG_CONV_LRELU | pflag, dt, stream));
            out_slot = fs[1];
        }
        cur = d->ws[out_slot];
        cur_slot = out_slot;
        if (pool) { h /= 2; w /= 2; }
        if (stage_out && stage_out[i]) TRY(subreg_nhwc_to_nchw(cur, stage_out[i], B, b.conv3.cout, h, w, dt, stream));
    }
    return subreg_avgpool(cur, feat, B, h, w, d->blocks[d->n_blocks - 1].conv3.cout, dt, stream);
}
