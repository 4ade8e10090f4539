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
               int W, float* stats, float* ws, long long ws_floats, void* stream, bool eval_mode = false) {
    TRY(subreg_conv_fwd_ws(x, c.w, tc.raw, nullptr, nullptr, nullptr, stats, nullptr, nullptr, 0, B, H, W, c.cin, c.cout, c.ksize,
                           SUBREG_CONV_RAW_STATS, d->dtype, ws, ws_floats, stream));
    if (eval_mode)      // eval-mode BatchNorm (running statistics; nothing is updated): the batch statistics just computed are not used
        return subreg_bn_eval_stash(c.bn_weight, c.bn_bias, c.running_mean, c.running_var, d->bn_eps, c.cout, tc.bscale, tc.bshift,
                                    tc.mean, tc.invstd, stream);
    const int rows = subreg_conv_stats_rows(d->dtype, B, H, W, c.cout);
    return subreg_bn_train_finalize(stats, rows, c.cout, (long long)B * H * W, c.bn_weight, c.bn_bias, c.running_mean,
                                    c.running_var, d->bn_momentum, d->bn_eps, tc.bscale, tc.bshift, tc.mean, tc.invstd, stream);
}

// BN backward (+ fused LeakyReLU' of `act`): d(raw conv output) into `draw`, d gamma / d beta
// pre_slices > 0: `partial` already holds that many slices of the reduce pass (subreg_block_tail_bwd_stats wrote them)
int bn_backward(const subreg_backbone_desc* d, const subreg_conv_desc& c, const subreg_conv_train& tc, const void* dy, const void* act,
                void* draw, double* partial, int B, int H, int W, void* stream, bool eval_mode = false, int pre_slices = 0) {
    if (pre_slices > 0)
        return subreg_bn_bwd_partials(dy, act, tc.raw, tc.mean, tc.invstd, c.bn_weight, partial, pre_slices, tc.grad_gamma, tc.grad_beta, draw,
                                      (long long)B * H * W, c.cout, d->dtype, eval_mode ? 1 : 0, stream);
    return (eval_mode ? subreg_bn_bwd_eval : subreg_bn_bwd)(dy, act, tc.raw, tc.mean, tc.invstd, c.bn_weight, partial, tc.grad_gamma,
                                                           tc.grad_beta, draw, (long long)B * H * W, c.cout, d->dtype, stream);
}

// the weight gradient of the conv that produced `raw`, from d(raw) and the conv's input
int weight_grad(const subreg_backbone_desc* d, const subreg_train_desc* t, const subreg_conv_desc& c, const subreg_conv_train& tc,
                const void* conv_input, const void* draw, int B, int H, int W, void* stream) {
    TRY(subreg_conv_wgrad(conv_input, draw, tc.gw_packed, t->pad_x, t->pad_dy, B, H, W, c.cin, c.cout, c.ksize, d->dtype, stream));
    return subreg_unpack_wgrad(tc.gw_packed, tc.grad_w, c.cout, c.cin_raw, c.ksize_raw, c.cin_raw == 3 ? 1 : 0,
                               subreg_conv_wgrad_splits(B, H, W, c.cin, c.cout, c.ksize, d->dtype), stream);
}

// Two-stream schedule of the step (subreg_train_desc.side_stream): the weight-gradient chain of a conv (pad copies, dW kernel,
// unpack) and the whole shortcut branch depend only on d(raw) / d(sum), not on each other or on the dX chain, and at the
// pretraining batch most of these launches cannot fill 256 CUs (52-200 workgroups on the 10x10 / 5x5 maps) - so they go to a
// second stream and overlap with the main stream's BatchNorm-backward -> dX chain.  Ordering is by events only (no host
// synchronisation): `ready` = main produced a buffer the side stream reads, `done` = the side stream finished reading it.
struct Fork {
    hipStream_t main, side;
    hipEvent_t ev[SUBREG_TRAIN_EVENTS];
    bool on;
    Fork(const subreg_train_desc* t, void* stream) : main((hipStream_t)stream), side((hipStream_t)t->side_stream), on(false) {
        on = t->side_stream != nullptr && t->side_stream != stream;
        for (int i = 0; i < SUBREG_TRAIN_EVENTS; ++i) {
            ev[i] = (hipEvent_t)t->events[i];
            if (!ev[i]) on = false;
        }
        if (!on) side = main;
    }
    // the side stream continues after everything the main stream has been given so far
    int main_to_side(int e) { return !on || (hipEventRecord(ev[e], main) == hipSuccess && hipStreamWaitEvent(side, ev[e], 0) == hipSuccess) ? SUBREG_OK : SUBREG_EHIP; }
    int mark_side(int e) { return !on || hipEventRecord(ev[e], side) == hipSuccess ? SUBREG_OK : SUBREG_EHIP; }
    int main_waits(int e) { return !on || hipStreamWaitEvent(main, ev[e], 0) == hipSuccess ? SUBREG_OK : SUBREG_EHIP; }
};
enum { EV_FORK = 0, EV_DOWN = 1, EV_READY0 = 2, EV_READY1 = 3, EV_DONE0 = 4, EV_DONE1 = 5 };

// Every event record / wait between two kernels of a stream costs a ~6 us bubble there (in-trace gaps), so the fork only pays
// where the launches it overlaps leave CUs idle.  First block (counted from the input) whose backward / forward forks; measured
// with tools/bench_train.py (profiles/r03_ab_train_two_streams.txt).  Environment overrides are for such measurements.
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
int fork_bwd_from() { static const int v = env_int("SUBREG_TRAIN_FORK_FROM", 0); return v; }
int fork_fwd_from() { static const int v = env_int("SUBREG_TRAIN_FORK_FWD_FROM", 0); return v; }
// Blocks below this index run their shortcut branch (BN backward + 1x1 dW) on the MAIN stream although their 3x3 dW chains fork:
// in the first two blocks (84x84 / 42x42 maps) the dW chains are longer than the BN-backward -> dX chain and main would wait for them
// (two gaps of ~100 us at the d(raw) buffer reuse in the trace); the shortcut branch is work main can take over.  The 1x1 convs
// have a dW scratch of their own (subreg_hip/train.py), so the two streams never share one.
bool tail_stats_on() { static const int v = env_int("SUBREG_NO_TAIL_STATS", 0); return v == 0; }   // (the switch is for A/B timing)
int down_on_side_from() { static const int v = env_int("SUBREG_TRAIN_DOWN_SIDE_FROM", 2); return v; }

}  // namespace

extern "C" int subreg_backbone_forward_stash(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* x_nchw, int B,
                                             int H, int W, float* feat, void* stream) {
    SUBREG_CHECK_ARG(d && t && d->blocks && t->blocks && d->n_blocks > 0 && x_nchw && feat && B > 0 && H > 0 && W > 0);
    SUBREG_CHECK_ARG(d->col && d->stats);
    const int dt = d->dtype;
    const bool ev = t->eval_mode != 0;     // eval-mode BatchNorm (running statistics), no dropout masks: the caller leaves keep_mask NULL
    Fork f(t, stream);
    if (!t->stats_side) f.on = false, f.side = f.main;
    TRY(subreg_pack_input(x_nchw, d->col, B, H, W, dt, stream));
    const void* cur = d->col;
    int h = H, w = W;
    for (int i = 0; i < d->n_blocks; ++i) {
        const subreg_block_desc& b = d->blocks[i];
        const subreg_block_train& tb = t->blocks[i];
        const int pflag = b.stride == 2 ? SUBREG_CONV_POOL2 : 0;
        SUBREG_CHECK_ARG(tb.conv1.raw && tb.conv1.act && tb.conv2.raw && tb.conv2.act && tb.conv3.raw && tb.out);
        const void* res = cur;
        const float *rsc = nullptr, *rsh = nullptr;
        const bool fork = f.on && i >= fork_fwd_from();
        if (b.down.w) {                    // the 1x1 shortcut conv + its statistics: beside conv1..conv3 on the side stream
            SUBREG_CHECK_ARG(tb.down.raw != nullptr);
            if (fork) TRY(f.main_to_side(EV_FORK));
            TRY(conv_stash(d, b.down, tb.down, cur, B, h, w, fork ? t->stats_side : d->stats, nullptr, 0, fork ? f.side : f.main, ev));
            if (fork) TRY(f.mark_side(EV_DOWN));
            res = tb.down.raw; rsc = tb.down.bscale; rsh = tb.down.bshift;
        }
        TRY(conv_stash(d, b.conv1, tb.conv1, cur, B, h, w, d->stats, t->splitk_ws, t->splitk_ws_floats, stream, ev));
        TRY(subreg_bn_apply(tb.conv1.raw, tb.conv1.bscale, tb.conv1.bshift, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, tb.conv1.act, B, h,
                            w, b.conv1.cout, SUBREG_CONV_LRELU, dt, stream));
        TRY(conv_stash(d, b.conv2, tb.conv2, tb.conv1.act, B, h, w, d->stats, t->splitk_ws, t->splitk_ws_floats, stream, ev));
        TRY(subreg_bn_apply(tb.conv2.raw, tb.conv2.bscale, tb.conv2.bshift, nullptr, nullptr, nullptr, nullptr, 1.f, nullptr, tb.conv2.act, B, h,
                            w, b.conv2.cout, SUBREG_CONV_LRELU, dt, stream));
        TRY(conv_stash(d, b.conv3, tb.conv3, tb.conv2.act, B, h, w, d->stats, t->splitk_ws, t->splitk_ws_floats, stream, ev));
        if (b.down.w && fork) TRY(f.main_waits(EV_DOWN));
        TRY(subreg_bn_apply(tb.conv3.raw, tb.conv3.bscale, tb.conv3.bshift, res, rsc, rsh, b.keep_mask, b.mask_scale, b.mask_scale_dev, tb.out, B, h, w,
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
    Fork f(t, stream);
    if (!t->dr_alt || !t->bn_partial_side) f.on = false, f.side = f.main;
    void* const drb[2] = {t->dr, f.on ? t->dr_alt : t->dr};     // d(raw) ping-pong: the side stream reads one while main fills the other
    bool busy[2] = {false, false};                              // the side stream has been given work that reads drb[k]
    bool down_busy = false;
    int use = 0;
    // BN backward on main into drb[k]; its dW chain on the side stream; returns the buffer for the dX convolution on main
    bool fork = false;                                          // this block's dW chains / shortcut branch go to the side stream
    auto join = [&]() -> int {                                  // main continues after everything the side stream was given
        if (down_busy) TRY(f.main_waits(EV_DOWN));
        for (int k = 0; k < 2; ++k)
            if (busy[k]) TRY(f.main_waits(EV_DONE0 + k));
        down_busy = busy[0] = busy[1] = false;
        return SUBREG_OK;
    };
    auto bn_then_wgrad = [&](const subreg_conv_desc& c, const subreg_conv_train& tc, const void* dy, const void* act,
                             const void* conv_input, int bh, int bw, const void** draw_out, int pre_slices = 0) -> int {
        const int k = use++ & 1;
        if (busy[k]) { TRY(f.main_waits(EV_DONE0 + k)); busy[k] = false; }
        TRY(bn_backward(d, c, tc, dy, act, drb[k], t->bn_partial, B, bh, bw, stream, t->eval_mode != 0, pre_slices));
        if (fork) {
            TRY(f.main_to_side(EV_READY0 + k));
            TRY(weight_grad(d, t, c, tc, conv_input, drb[k], B, bh, bw, f.side));
            TRY(f.mark_side(EV_DONE0 + k));
            busy[k] = true;
        } else {
            TRY(weight_grad(d, t, c, tc, conv_input, drb[k], B, bh, bw, stream));
        }
        *draw_out = drb[k];
        return SUBREG_OK;
    };
    // The blocks, as a callable: whatever it returns, the side stream is joined before this call does (the header's promise;
    // an error return between a fork and its join would otherwise leave side-stream work nobody waits for)
    auto run_blocks = [&]() -> int {
    for (int i = last_block; i >= first_block; --i) {
        const int gi = (nb - 1 - i) & 1;
        const subreg_block_desc& b = d->blocks[i];
        const subreg_block_train& tb = t->blocks[i];
        const void* xin = i == 0 ? d->col : t->blocks[i - 1].out;
        const int bh = hs[i], bw = ws[i], C = b.conv3.cout;
        const void* res = b.down.w ? tb.down.raw : xin;
        const bool fork_now = f.on && i >= fork_bwd_from();
        if (fork && !fork_now) TRY(join());     // the one-stream blocks use the dW scratch buffers (pad_x, pad_dy, gw) on main
        fork = fork_now;
        if (t->grad_out_dump && t->grad_out_dump[i]) {
            const int oh = b.stride == 2 ? bh / 2 : bh, ow = b.stride == 2 ? bw / 2 : bw;
            if (hipMemcpyAsync(t->grad_out_dump[i], t->g[gi], (size_t)B * oh * ow * C * (dt == SUBREG_BF16 ? 2 : 4),
                               hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return SUBREG_EHIP;
        }
        // d(pre-activation sum) from d(out): keep mask, max-pool routing, LeakyReLU'
        // ... and, in the same pass, the reduce pass of the two BatchNorms that consume dV: bn3 (into bn_partial) and the shortcut's
        // (into bn_partial_side: its finalize may run on the side stream, or on main after bn3 .. bn1 reused bn_partial).  The side
        // stream's last use of bn_partial_side (the previous block's shortcut branch) is behind that block's main_waits(EV_DOWN).
        const bool tail_stats = tail_stats_on();
        const bool ds_stats = tail_stats && b.down.w && t->bn_partial_side;
        int tail_slices = 0;
        if (tail_stats) {
            TRY(subreg_block_tail_bwd_stats(t->g[gi], b.keep_mask, b.mask_scale, b.mask_scale_dev, tb.conv3.raw, tb.conv3.bscale, tb.conv3.bshift, res,
                                            b.down.w ? tb.down.bscale : nullptr, b.down.w ? tb.down.bshift : nullptr, t->dv, B, bh, bw, C,
                                            b.stride == 2, dt, tb.conv3.mean, tb.conv3.invstd, t->bn_partial, ds_stats ? tb.down.mean : nullptr,
                                            ds_stats ? tb.down.invstd : nullptr, ds_stats ? t->bn_partial_side : nullptr, &tail_slices, stream));
        } else {
            TRY(subreg_block_tail_bwd(t->g[gi], b.keep_mask, b.mask_scale, b.mask_scale_dev, tb.conv3.raw, tb.conv3.bscale, tb.conv3.bshift, res,
                                      b.down.w ? tb.down.bscale : nullptr, b.down.w ? tb.down.bshift : nullptr, t->dv, B, bh, bw, C,
                                      b.stride == 2, dt, stream));
        }
        // shortcut branch (BN backward -> dr2, dW): needs only dv; all of it on the side stream
        // INVARIANT the two-stream schedule rests on: the shortcut convolution is 1x1, and the 1x1 path of subreg_conv_wgrad never
        // touches the dW scratch (t->pad_x / t->pad_dy) that the 3x3 dW chains of the OTHER stream are using at the same time; its
        // own dW scratch (gw of `down`) is a separate buffer (train.py).  A 3x3 shortcut would race: refused here.
        SUBREG_CHECK_ARG(!b.down.w || b.down.ksize_raw == 1);
        const bool down_side = fork && i >= down_on_side_from();
        if (b.down.w && down_side) {
            TRY(f.main_to_side(EV_FORK));
            TRY(bn_backward(d, b.down, tb.down, t->dv, nullptr, t->dr2, t->bn_partial_side, B, bh, bw, f.side, t->eval_mode != 0,
                            ds_stats ? tail_slices : 0));
            TRY(weight_grad(d, t, b.down, tb.down, xin, t->dr2, B, bh, bw, f.side));
            TRY(f.mark_side(EV_DOWN));
            down_busy = true;
        }
        // main branch: bn3/conv3 -> bn2/conv2 -> bn1/conv1
        const void* dr = nullptr;
        TRY(bn_then_wgrad(b.conv3, tb.conv3, t->dv, nullptr, tb.conv2.act, bh, bw, &dr, tail_slices));
        TRY(subreg_conv_fwd_ws(dr, tb.conv3.w_dgrad, t->dt, nullptr, t->zero_shift, nullptr, nullptr, nullptr, nullptr, 0, B, bh, bw,
                               b.conv3.cout, b.conv3.cin, b.conv3.ksize_raw, 0, dt, t->splitk_ws, t->splitk_ws_floats, stream));
        TRY(bn_then_wgrad(b.conv2, tb.conv2, t->dt, tb.conv2.act, tb.conv1.act, bh, bw, &dr));
        TRY(subreg_conv_fwd_ws(dr, tb.conv2.w_dgrad, t->dt, nullptr, t->zero_shift, nullptr, nullptr, nullptr, nullptr, 0, B, bh, bw,
                               b.conv2.cout, b.conv2.cin, b.conv2.ksize_raw, 0, dt, t->splitk_ws, t->splitk_ws_floats, stream));
        TRY(bn_then_wgrad(b.conv1, tb.conv1, t->dt, tb.conv1.act, xin, bh, bw, &dr));
        if (b.down.w && !down_side) {      // shortcut branch on the main stream
            TRY(bn_backward(d, b.down, tb.down, t->dv, nullptr, t->dr2, ds_stats ? t->bn_partial_side : t->bn_partial, B, bh, bw, stream,
                            t->eval_mode != 0, ds_stats ? tail_slices : 0));
            TRY(weight_grad(d, t, b.down, tb.down, xin, t->dr2, B, bh, bw, stream));
        }
        // the shortcut branch's results (dr2) and its reads of dv: main continues after them
        if (down_busy) { TRY(f.main_waits(EV_DOWN)); down_busy = false; }
        if (i == 0) break;                 // no gradient w.r.t. the images
        // d(block input) = dX(conv1) + (dX(shortcut conv) | d(pre-activation sum))
        SUBREG_CHECK_ARG(tb.conv1.w_dgrad && (!b.down.w || tb.down.w_dgrad));
        const int go = gi ^ 1;
        if (b.down.w) {
            TRY(subreg_conv_fwd(dr, tb.conv1.w_dgrad, t->g[go], nullptr, t->zero_shift, nullptr, nullptr, t->dr2, tb.down.w_dgrad,
                                b.down.cout, B, bh, bw, b.conv1.cout, b.conv1.cin, b.conv1.ksize_raw, 0, dt, stream));
        } else {
            TRY(subreg_conv_fwd(dr, tb.conv1.w_dgrad, t->g[go], nullptr, t->zero_shift, t->dv, nullptr, nullptr, nullptr, 0, B, bh,
                                bw, b.conv1.cout, b.conv1.cin, b.conv1.ksize_raw, 0, dt, stream));
        }
    }
    return SUBREG_OK;
    };
    const int rc = run_blocks();
    // join: every gradient of these blocks is complete in main-stream order when this call's work is
    const int rj = join();
    return rc != SUBREG_OK ? rc : rj;
}

extern "C" int subreg_event_create(void** event) {
    SUBREG_CHECK_ARG(event);
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return SUBREG_EHIP;
    *event = (void*)e;
    return SUBREG_OK;
}

extern "C" int subreg_event_destroy(void* event) {
    SUBREG_CHECK_ARG(event);
    return hipEventDestroy((hipEvent_t)event) == hipSuccess ? SUBREG_OK : SUBREG_EHIP;
}

extern "C" int subreg_backbone_backward(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* dfeat, int B, int H,
                                        int W, void* stream) {
    SUBREG_CHECK_ARG(d && d->n_blocks > 0);
    return subreg_backbone_backward_blocks(d, t, dfeat, B, H, W, 0, d->n_blocks - 1, stream);
}
