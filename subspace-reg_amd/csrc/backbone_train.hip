// Pretraining step over the whole backbone: train-mode forward that stashes what autograd would save, and the
// backward launch sequence (train_supervised.py:229-244 `output = model(input)` ... `loss.backward()` over
// models/resnet_language.py BasicBlock.forward :268-301).  Host-side orchestration only; kernels live in
// conv_fwd.hip (forward and dX), backward.hip, elementwise.hip.
#include "subreg_common.h"

namespace {

#define TRY(expr)                        \
    do {                                 \
        const int _rc = (expr);          \
        if (_rc != SUBREG_OK) return _rc; \
    } while (0)

// raw conv + batch statistics; scale/shift of this batch, mean/invstd saved for the backward
int conv_stash(const subreg_backbone_desc* d, const subreg_conv_desc& c, const subreg_conv_train& tc, const void* x, int B, int H,
               int W, void* stream) {
    TRY(subreg_conv_fwd(x, c.w, tc.raw, nullptr, nullptr, nullptr, d->stats, nullptr, nullptr, 0, B, H, W, c.cin, c.cout, c.ksize,
                        SUBREG_CONV_RAW_STATS, d->dtype, stream));
    const int rows = subreg_conv_stats_rows(d->dtype, B, H, W, c.cout);
    return subreg_bn_train_finalize(d->stats, rows, c.cout, (long long)B * H * W, c.bn_weight, c.bn_bias, c.running_mean,
                                    c.running_var, d->bn_momentum, d->bn_eps, tc.bscale, tc.bshift, tc.mean, tc.invstd, stream);
}

// BN backward (+ fused LeakyReLU' of `act`) then the weight gradient of the conv that produced `raw`
int bn_and_wgrad(const subreg_backbone_desc* d, const subreg_train_desc* t, const subreg_conv_desc& c, const subreg_conv_train& tc,
                 const void* dy, const void* act, const void* conv_input, void* draw, int B, int H, int W, void* stream) {
    TRY(subreg_bn_bwd(dy, act, tc.raw, tc.mean, tc.invstd, c.bn_weight, t->bn_partial, tc.grad_gamma, tc.grad_beta, draw,
                      (long long)B * H * W, c.cout, d->dtype, stream));
    TRY(subreg_conv_wgrad(conv_input, draw, tc.gw_packed, t->pad_x, t->pad_dy, B, H, W, c.cin, c.cout, c.ksize, d->dtype, stream));
    return subreg_unpack_wgrad(tc.gw_packed, tc.grad_w, c.cout, c.cin_raw, c.ksize_raw, c.cin_raw == 3 ? 1 : 0,
                               subreg_conv_wgrad_splits(B, H, W, c.cin, c.cout, c.ksize, d->dtype), stream);
}

}  // namespace

extern "C" int subreg_backbone_forward_stash(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* x_nchw, int B,
                                             int H, int W, float* feat, void* stream) {
    SUBREG_CHECK_ARG(d && t && d->blocks && t->blocks && d->n_blocks > 0 && x_nchw && feat && B > 0 && H > 0 && W > 0);
    SUBREG_CHECK_ARG(d->col && d->stats);
    const int dt = d->dtype;
    TRY(subreg_pack_input(x_nchw, d->col, B, H, W, dt, stream));
    const void* cur = d->col;
    int h = H, w = W;
    for (int i = 0; i < d->n_blocks; ++i) {
        const subreg_block_desc& b = d->blocks[i];
        const subreg_block_train& tb = t->blocks[i];
        const int pflag = b.stride == 2 ? SUBREG_CONV_POOL2 : 0;
        SUBREG_CHECK_ARG(tb.conv1.raw && tb.conv1.act && tb.conv2.raw && tb.conv2.act && tb.conv3.raw && tb.out);
        TRY(conv_stash(d, b.conv1, tb.conv1, cur, B, h, w, stream));
        TRY(subreg_bn_apply(tb.conv1.raw, tb.conv1.bscale, tb.conv1.bshift, nullptr, nullptr, nullptr, nullptr, 1.f, tb.conv1.act, B, h,
                            w, b.conv1.cout, SUBREG_CONV_LRELU, dt, stream));
        TRY(conv_stash(d, b.conv2, tb.conv2, tb.conv1.act, B, h, w, stream));
        TRY(subreg_bn_apply(tb.conv2.raw, tb.conv2.bscale, tb.conv2.bshift, nullptr, nullptr, nullptr, nullptr, 1.f, tb.conv2.act, B, h,
                            w, b.conv2.cout, SUBREG_CONV_LRELU, dt, stream));
        TRY(conv_stash(d, b.conv3, tb.conv3, tb.conv2.act, B, h, w, stream));
        const void* res = cur;
        const float *rsc = nullptr, *rsh = nullptr;
        if (b.down.w) {
            SUBREG_CHECK_ARG(tb.down.raw != nullptr);
            TRY(conv_stash(d, b.down, tb.down, cur, B, h, w, stream));
            res = tb.down.raw; rsc = tb.down.bscale; rsh = tb.down.bshift;
        }
        TRY(subreg_bn_apply(tb.conv3.raw, tb.conv3.bscale, tb.conv3.bshift, res, rsc, rsh, b.keep_mask, b.mask_scale, tb.out, B, h, w,
                            b.conv3.cout, SUBREG_CONV_LRELU | pflag, dt, stream));
        cur = tb.out;
        if (b.stride == 2) { h /= 2; w /= 2; }
    }
    return subreg_avgpool(cur, feat, B, h, w, d->blocks[d->n_blocks - 1].conv3.cout, dt, stream);
}

extern "C" int subreg_backbone_backward_blocks(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* dfeat, int B,
                                              int H, int W, int first_block, int last_block, void* stream) {
    SUBREG_CHECK_ARG(d && t && d->blocks && t->blocks && d->n_blocks > 0 && dfeat && B > 0 && H > 0 && W > 0);
    SUBREG_CHECK_ARG(t->g[0] && t->g[1] && t->dv && t->dr && t->dt && t->dr2 && t->bn_partial && t->zero_shift);
    const int dt = d->dtype, nb = d->n_blocks;
    SUBREG_CHECK_ARG(0 <= first_block && first_block <= last_block && last_block < nb);
    int hs[64], ws[64];                    // input spatial size of every block
    SUBREG_CHECK_ARG(nb <= 64);
    int h = H, w = W;
    for (int i = 0; i < nb; ++i) { hs[i] = h; ws[i] = w; if (d->blocks[i].stride == 2) { h /= 2; w /= 2; } }
    // d(loss)/d(output of block i) lives in g[(nb - 1 - i) & 1]: the ping-pong index is a function of the block, so the
    // backward can be issued in several calls (descending, contiguous block ranges)
    if (last_block == nb - 1) TRY(subreg_avgpool_bwd(dfeat, t->g[0], B, h, w, d->blocks[nb - 1].conv3.cout, dt, stream));
    for (int i = last_block; i >= first_block; --i) {
        const int gi = (nb - 1 - i) & 1;
        const subreg_block_desc& b = d->blocks[i];
        const subreg_block_train& tb = t->blocks[i];
        const void* xin = i == 0 ? d->col : t->blocks[i - 1].out;
        const int bh = hs[i], bw = ws[i], C = b.conv3.cout;
        const void* res = b.down.w ? tb.down.raw : xin;
        if (t->grad_out_dump && t->grad_out_dump[i]) {
            const int oh = b.stride == 2 ? bh / 2 : bh, ow = b.stride == 2 ? bw / 2 : bw;
            if (hipMemcpyAsync(t->grad_out_dump[i], t->g[gi], (size_t)B * oh * ow * C * (dt == SUBREG_BF16 ? 2 : 4),
                               hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return SUBREG_EHIP;
        }
        // d(pre-activation sum) from d(out): keep mask, max-pool routing, LeakyReLU'
        TRY(subreg_block_tail_bwd(t->g[gi], b.keep_mask, b.mask_scale, tb.conv3.raw, tb.conv3.bscale, tb.conv3.bshift, res,
                                  b.down.w ? tb.down.bscale : nullptr, b.down.w ? tb.down.bshift : nullptr, t->dv, B, bh, bw, C,
                                  b.stride == 2, dt, stream));
        // main branch: bn3/conv3 -> bn2/conv2 -> bn1/conv1
        TRY(bn_and_wgrad(d, t, b.conv3, tb.conv3, t->dv, nullptr, tb.conv2.act, t->dr, B, bh, bw, stream));
        TRY(subreg_conv_fwd(t->dr, tb.conv3.w_dgrad, t->dt, nullptr, t->zero_shift, nullptr, nullptr, nullptr, nullptr, 0, B, bh, bw,
                            b.conv3.cout, b.conv3.cin, b.conv3.ksize_raw, 0, dt, stream));
        TRY(bn_and_wgrad(d, t, b.conv2, tb.conv2, t->dt, tb.conv2.act, tb.conv1.act, t->dr, B, bh, bw, stream));
        TRY(subreg_conv_fwd(t->dr, tb.conv2.w_dgrad, t->dt, nullptr, t->zero_shift, nullptr, nullptr, nullptr, nullptr, 0, B, bh, bw,
                            b.conv2.cout, b.conv2.cin, b.conv2.ksize_raw, 0, dt, stream));
        TRY(bn_and_wgrad(d, t, b.conv1, tb.conv1, t->dt, tb.conv1.act, xin, t->dr, B, bh, bw, stream));
        // shortcut branch
        if (b.down.w) TRY(bn_and_wgrad(d, t, b.down, tb.down, t->dv, nullptr, xin, t->dr2, B, bh, bw, stream));
        if (i == 0) break;                 // no gradient w.r.t. the images
        // d(block input) = dX(conv1) + (dX(shortcut conv) | d(pre-activation sum))
        SUBREG_CHECK_ARG(tb.conv1.w_dgrad && (!b.down.w || tb.down.w_dgrad));
        const int go = gi ^ 1;
        if (b.down.w) {
            TRY(subreg_conv_fwd(t->dr, tb.conv1.w_dgrad, t->g[go], nullptr, t->zero_shift, nullptr, nullptr, t->dr2, tb.down.w_dgrad,
                                b.down.cout, B, bh, bw, b.conv1.cout, b.conv1.cin, b.conv1.ksize_raw, 0, dt, stream));
        } else {
            TRY(subreg_conv_fwd(t->dr, tb.conv1.w_dgrad, t->g[go], nullptr, t->zero_shift, t->dv, nullptr, nullptr, nullptr, 0, B, bh,
                                bw, b.conv1.cout, b.conv1.cin, b.conv1.ksize_raw, 0, dt, stream));
        }
    }
    return SUBREG_OK;
}

extern "C" int subreg_backbone_backward(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* dfeat, int B, int H,
                                        int W, void* stream) {
    SUBREG_CHECK_ARG(d && d->n_blocks > 0);
    return subreg_backbone_backward_blocks(d, t, dfeat, B, H, W, 0, d->n_blocks - 1, stream);
}
