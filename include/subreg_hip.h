/* libsubreg_hip — C ABI of the MI355X-native (gfx950) incremental-episode hot path of
 * feyzaakyurek/subspace-reg.
 *
 * The reference has NO native/FFI interface: its seam is "a torch.nn.Module + autograd"
 * (SURVEY.md section 8b).  These entry points are what the host-side mirror
 * (the Python package subspace-reg_amd/subreg_hip, torch.autograd.Function stubs over ctypes) binds;
 * each one names the reference call site it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is CALLER-OWNED device memory (the host
 *     side passes torch.Tensor.data_ptr()); the library allocates nothing and keeps no
 *     mutable global state, so one process per GPU is safe.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no call
 *     synchronises, allocates or copies to the host => all of it is hipGraph-capturable.
 *   - return value: SUBREG_OK (0) or a negative code; nothing throws across the ABI.
 *     -(1000+e) carries hipError_t e of a failed launch.
 *   - activations: compact NHWC [B*H*W][C] in the compute dtype (SUBREG_F32 | SUBREG_BF16);
 *     BN parameters, features, classifier and all loss scalars are fp32.
 *   - single caller thread per process (the reference loop is single-threaded,
 *     --num_workers 0).
 */
#ifndef SUBREG_HIP_H
#define SUBREG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SUBREG_ABI_VERSION 14

#define SUBREG_OK 0
#define SUBREG_EINVAL (-1)       /* bad argument (null pointer, shape not supported by contract) */
#define SUBREG_EUNSUPPORTED (-2) /* valid request the kernels cannot serve (e.g. image too wide for the LDS patch) */
#define SUBREG_EHIP (-3)         /* HIP runtime call failed */

#define SUBREG_F32 0  /* exact-f32 MFMA path: the 1e-4 parity gate */
#define SUBREG_BF16 1 /* bf16 MFMA, fp32 accumulate: the throughput path */

/* flags of subreg_conv_fwd / subreg_bn_apply */
#define SUBREG_CONV_LRELU 1     /* nn.LeakyReLU(0.1), models/resnet_language.py:251 */
#define SUBREG_CONV_POOL2 2     /* nn.MaxPool2d(2) floor mode, :256,290 */
#define SUBREG_CONV_RAW_STATS 4 /* train mode: raw conv output + per-channel (sum,sumsq) partials */
/* kernel selection of subreg_conv_fwd for Cout % 160 == 0 (default: a measured rule): force the general kernel (conv_fwd.hip) or the
 * one-wave-per-SIMD kernel (conv_wide.hip; SUBREG_EUNSUPPORTED where it does not take the problem).  Parity tests and A/B runs.
 * subreg_conv12_first_fused takes the same two flags: the 8-wave kernel (the default) or the one-wave-per-SIMD one
 * (conv64_resident.hip::conv64_fused_first_kernel / conv64_wide_fused_kernel). */
#define SUBREG_CONV_KERNEL_GENERAL 256
#define SUBREG_CONV_KERNEL_WIDE 512
#define SUBREG_CONV_KERNEL_WIDE_ALT 1024  /* with _WIDE: the OTHER MFMA shape of conv_wide.hip than its default for the problem (16x16x32 by default): parity tests, A/B runs */
#define SUBREG_CONV_KERNEL_WIDE_128 2048  /* with _WIDE: the 128-row tiling of conv_wide16_kernel (SUBREG_EUNSUPPORTED where its patch does not fit) */
#define SUBREG_CONV_KERNEL_WIDE_256 4096  /* with _WIDE: the 256-row tiling */

/* flags of subreg_backbone_forward */
#define SUBREG_FWD_TRAIN 1 /* BN batch statistics + running-stat update + keep masks (net.train(), eval/language_eval.py:211) */

int subreg_abi_version(void);
const char* subreg_strerror(int code);

/* ---- layout packing ------------------------------------------------------------------------- */
/* First layer (Cin = 3): x NCHW fp32 [B,3,H,W] -> im2col rows [B*H*W][32], k = 3*(3*ky+kx)+c, zero padded.
 * Turns models/resnet_language.py:249 (conv1 of layer1.0) and :146 (its 1x1 shortcut) into K=32 GEMMs. */
int subreg_pack_input(const float* x_nchw, void* col, int B, int H, int W, int dtype, void* stream);
/* Conv2d.weight OIHW fp32 -> [k*k][Cin/32][Cout][32] (mode 0: one (tap, 32-channel chunk) tile = Cout contiguous rows,
 * the unit subreg_conv_fwd stages) or the K=32 first-layer layout [Cout][32] (mode 1, Cin == 3);
 * fold_scale [Cout] (may be NULL) multiplies output channel o by fold_scale[o] (eval-mode BN scale folded in). */
int subreg_pack_conv_weight(const float* w_oihw, const float* fold_scale, void* out, int Cout, int Cin, int ksize, int mode,
                            int dtype, void* stream);
/* identity weights [C/32][C][32] (the 1x1 layout above) for the identity-shortcut GEMM (layer3.1 / layer4.1, resnet_language.py:271,288) */
int subreg_pack_identity(void* out, int C, int dtype, void* stream);
int subreg_vec_add(float* dst, const float* a, const float* b, int n, void* stream);
int subreg_nchw_to_nhwc(const float* x_nchw, void* y_nhwc, int B, int C, int H, int W, int dtype, void* stream);
int subreg_nhwc_to_nchw(const void* x_nhwc, float* y_nchw, int B, int C, int H, int W, int dtype, void* stream);

/* ---- convolution: replaces nn.Conv2d (conv3x3 :402-405, 1x1 shortcut :146-147) + fused epilogue -------- */
/* y = [pool2]( [lrelu]( (conv(x,w) + x2*w2^T) * scale + shift [+ residual] ) ), or with SUBREG_CONV_RAW_STATS:
 * y = conv(x,w) and stats_partial[rows][Cout][2] (rows = subreg_conv_stats_rows).  scale == NULL: already folded into
 * the packed weights.  (x2 [B*H*W][Cin2], w2 [Cin2/32][Cout][32]) is the fused shortcut GEMM of BasicBlock.forward :286-288
 * (1x1 conv+BN, or the identity with w2 = I); x2 == NULL: none.  Cin, Cin2, Cout multiples of 32; ksize 1 or 3. */
int subreg_conv_fwd(const void* x, const void* w, void* y, const float* scale, const float* shift, const void* residual,
                    float* stats_partial, const void* x2, const void* w2, int Cin2, int B, int H, int W, int Cin, int Cout,
                    int ksize, int flags, int dtype, void* stream);
/* The same with a caller-owned fp32 workspace: small-M 3x3 layers (the 10x10 / 5x5 maps at the pretraining batch: 52-100 tiles
 * for 256 CUs) split K over several workgroups per tile and finish in a second launch.  Used only when workspace_floats >=
 * subreg_conv_splitk_floats(...) > 0 and there is no pooling / residual / fused shortcut; otherwise exactly subreg_conv_fwd. */
int subreg_conv_fwd_ws(const void* x, const void* w, void* y, const float* scale, const float* shift, const void* residual,
                       float* stats_partial, const void* x2, const void* w2, int Cin2, int B, int H, int W, int Cin, int Cout,
                       int ksize, int flags, int dtype, float* workspace, long long workspace_floats, void* stream);
long long subreg_conv_splitk_floats(int B, int H, int W, int Cin, int Cout, int ksize, int dtype);
int subreg_conv_stats_rows(int dtype, int B, int H, int W, int Cout);
/* First layer without an im2col buffer (bf16, Cout = 64, eval mode): y = [lrelu](conv3x3(x_nchw, w) + shift) straight from the
 * fp32 NCHW image of the reference's loaders (models/resnet_language.py:249-251 with the BN scale folded into w);
 * w_packed = the mode-1 layout of subreg_pack_conv_weight.  SUBREG_EUNSUPPORTED for other dtypes / widths. */
int subreg_conv_first_fwd(const float* x_nchw, const void* w_packed, void* y, const float* shift, int B, int H, int W, int Cout,
                          int flags, int dtype, void* stream);
/* conv3 of layer1.0 with its 1x1 shortcut (models/resnet_language.py:146-147,254-256,286-290) fed from the fp32 NCHW image:
 * y = pool2(lrelu(conv3x3(x, w) + conv1x1(img, w2_first) + shift)); x [B*H*W][64] bf16, w2_first = the mode-1 packed 1x1 weights.
 * bf16, Cin = Cout = 64, SUBREG_CONV_POOL2 required; SUBREG_EUNSUPPORTED where subreg_layer1_direct_supported says 0. */
int subreg_conv_fwd_image_shortcut(const void* x, const void* w, void* y, const float* shift, const float* img_nchw,
                                   const void* w2_first, int B, int H, int W, int Cin, int Cout, int flags, int dtype, void* stream);
/* conv1 + BN + LeakyReLU + conv2 + BN + LeakyReLU of layer1.0 in one launch (models/resnet_language.py:249-253, eval mode):
 * y [B*H*W][64] bf16 = lrelu(conv3x3(lrelu(conv3x3(x_nchw, w1) + shift1), w2) + shift2); the 64-channel intermediate stays in
 * LDS.  w1 = mode-1 packed first-layer weights, w2 = mode-0 packed 64 -> 64 weights (both BN-folded).  bf16, 84x84-class images
 * only: SUBREG_EUNSUPPORTED otherwise (the caller then runs subreg_conv_first_fwd + subreg_conv_fwd). */
int subreg_conv12_first_fused(const float* x_nchw, const void* w1_packed, const float* shift1, const void* w2_packed,
                              const float* shift2, void* y, int B, int H, int W, int flags, int dtype, void* stream);
/* 1 if an eval-mode forward of (B, H, W) images runs layer 1 without the im2col buffer (backbone desc `col` may then be NULL) */
int subreg_layer1_direct_supported(int B, int H, int W, int dtype);

/* ---- BatchNorm2d (:148,250,253,255) -------------------------------------------------------------------- */
/* eval: scale = weight/sqrt(running_var+eps), shift = bias - running_mean*scale */
int subreg_bn_fold(const float* weight, const float* bias, const float* running_mean, const float* running_var,
                   float* scale, float* shift, int C, float eps, void* stream);
/* train: reduce the partials -> batch scale/shift; running stats updated in place (momentum, unbiased var) */
/* eval-mode BatchNorm of a forward that keeps a stash (subreg_train_desc.eval_mode): scale = w / sqrt(rv + eps), shift = b - rm * scale,
 * save_mean = rm, save_invstd = 1 / sqrt(rv + eps) (nn.BatchNorm2d in eval mode, models/resnet_language.py:250-255) */
int subreg_bn_eval_stash(const float* weight, const float* bias, const float* running_mean, const float* running_var, float eps,
                         int C, float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
int subreg_bn_train_finalize(const float* stats_partial, int rows, int C, long long count, const float* weight,
                             const float* bias, float* running_mean, float* running_var, float momentum, float eps,
                             float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
/* train second pass: y = keep*mask_scale * [pool2]([lrelu]( x*scale+shift + residual*res_scale+res_shift )) */
int subreg_bn_apply(const void* x, const float* scale, const float* shift, const void* residual, const float* res_scale,
                    const float* res_shift, const unsigned char* keep_mask, float mask_scale,
                    const float* mask_scale_dev /* device override of mask_scale, or NULL */, void* y, int B, int H, int W,
                    int C, int flags, int dtype, void* stream);
/* F.dropout (:299) / DropBlock (:311-325) masks: NCHW {0,1} floats -> NHWC u8 keep mask (invert: keep = 1-mask) */
int subreg_mask_nchw_to_nhwc(const float* mask_nchw, unsigned char* keep_nhwc, int B, int C, int H, int W, int invert,
                             void* stream);
/* counter-based Bernoulli(1-p_drop) keep mask for free-running train forwards; kept_count may be NULL */
int subreg_random_keep_mask(unsigned char* keep, long long n, unsigned long long seed, float p_drop, unsigned int* kept_count,
                            void* stream);
/* the same with the seed and the drop probability read from device memory when the launch runs (a captured hipGraph replays it with
 * fresh randomness and DropBlock's step-dependent gamma, :294-296); subreg_mask_params_set writes n <= SUBREG_MASK_PARAMS_MAX
 * parameter records from HOST values (they travel as a kernel argument: no staging buffer, nothing to keep alive) */
#define SUBREG_MASK_PARAMS_MAX 16
typedef struct subreg_mask_param {
    unsigned long long seed;
    float p_drop;
    unsigned int reserved;
} subreg_mask_param;
int subreg_random_keep_mask_dev(unsigned char* keep, long long n, const subreg_mask_param* param, unsigned int* kept_count, void* stream);
int subreg_mask_params_set(subreg_mask_param* params_dev, int n, const subreg_mask_param* host_values, void* stream);
/* DropBlock's rescale factor numel / max(count, 1) (resnet_language.py:318-323) from the counter the mask kernels fill, written
 * to a device float for subreg_block_desc.mask_scale_dev */
int subreg_mask_scale(const unsigned int* kept_count, long long numel, float* scale, void* stream);
/* DropBlock._compute_block_mask (:327-357) for block_size > 1: sample [B][C][H-bs+1][W-bs+1] u8 (1 = seed, NCHW order like
 * the reference's Bernoulli sample) -> keep mask NHWC u8 and the number of kept elements (scale = numel / kept, :320-323);
 * restates the reference's seed / offset pairing (nz.repeat vs offsets.repeat). */
int subreg_dropblock_mask(const unsigned char* sample, unsigned char* keep_nhwc, int B, int C, int H, int W, int block_size,
                          unsigned int* kept_count, void* stream);
/* AdaptiveAvgPool2d(1) + view (:125,179-181): [B][H*W][C] -> fp32 [B][C] */
int subreg_avgpool(const void* x, float* feat, int B, int H, int W, int C, int dtype, void* stream);

/* ---- whole backbone: ResNet.forward up to `feat` (:170-182), BasicBlock.forward (:268-301) ------------- */
typedef struct subreg_conv_desc {
    const void* w;            /* packed RAW weight in the compute dtype (train mode); NULL = absent (identity shortcut) */
    const void* w_folded;     /* packed weight * eval-mode BN scale (eval mode), refreshed by subreg_backbone_fold */
    const float* w_oihw;      /* the fp32 Conv2d.weight the two packed copies are made from */
    const float* bn_weight;   /* [cout] */
    const float* bn_bias;     /* [cout] */
    float* running_mean;      /* [cout] updated in train mode */
    float* running_var;       /* [cout] */
    float* scale;             /* [cout] folded (eval) / batch (train) scale, written by the library */
    float* shift;             /* [cout] */
    int cin, cout, ksize;     /* as the kernel sees them (first layer: cin 32, ksize 1 over the im2col rows) */
    int cin_raw, ksize_raw;   /* Conv2d shape (3 / 3x3 for the first layer) */
} subreg_conv_desc;

typedef struct subreg_block_desc {
    subreg_conv_desc conv1, conv2, conv3, down;
    const void* w_identity;         /* [cout/32][cout][32] identity (subreg_pack_identity) for an identity shortcut, else NULL */
    float* shift3;                  /* [cout] conv3 epilogue shift: bn3 shift + shortcut-BN shift, written by the fold */
    int stride;                     /* 2: MaxPool2d(2); 1: identity */
    const unsigned char* keep_mask; /* train: NHWC u8 keep mask of the block output, NULL = keep all */
    float mask_scale;               /* 1/(1-p) for dropout, countM/count_ones for DropBlock */
    const float* mask_scale_dev;    /* non-NULL: the factor is read from this DEVICE float instead (subreg_mask_scale: DropBlock's
                                       count_ones stays on the device, as in the reference - no host read per step) */
} subreg_block_desc;

typedef struct subreg_backbone_desc {
    int n_blocks;
    const subreg_block_desc* blocks; /* HOST array */
    int dtype;
    void* col;         /* [B*H*W][32] first-layer im2col rows */
    void* ws[4];       /* activation workspaces, each >= subreg_backbone_ws_bytes(...) */
    float* stats;      /* train: BN partials, >= subreg_backbone_stats_floats(...) floats */
    float bn_eps, bn_momentum;
} subreg_backbone_desc;

long long subreg_backbone_ws_bytes(const subreg_backbone_desc* d, int B, int H, int W);
/* 1 if subreg_backbone_forward of (B, H, W) images in this mode reads the im2col buffer `col` (it must then be non-NULL), 0 if
 * layer 1 reads the fp32 image itself.  The one predicate the library and its callers share: it needs both the shape support of
 * subreg_layer1_direct_supported and a first block of the layer1.0 form (ResNet.forward, models/resnet_language.py:170-192). */
int subreg_backbone_needs_col(const subreg_backbone_desc* d, int B, int H, int W, int train);
long long subreg_backbone_stats_floats(const subreg_backbone_desc* d, int B, int H, int W);
/* (re)pack every conv weight: the raw copy, and the copy with the eval-mode BN scale folded in; writes shift[].
 * Call after the weights or the BN statistics changed. */
int subreg_backbone_fold(const subreg_backbone_desc* d, void* stream);
/* only the raw packed copies (`w`): all a train-mode forward needs after an optimiser step (the folded copies and the
 * eval-mode scale/shift stay stale until the next subreg_backbone_fold) */
int subreg_backbone_pack_raw(const subreg_backbone_desc* d, void* stream);
/* x NCHW fp32 [B,3,H,W] -> feat fp32 [B][C_last].  stage_out[i] (optional, may be NULL) receives block i's
 * output as NCHW fp32 (is_feat=True, :189-190). */
int subreg_backbone_forward(const subreg_backbone_desc* d, const float* x_nchw, int B, int H, int W, float* feat,
                            float* const* stage_out, int flags, void* stream);

/* ---- pretraining step: train_supervised.py:205-268 (output = model(input); loss.backward()) ----------------- */
/* dX of a conv = subreg_conv_fwd on dY with these weights: OIHW fp32 -> [taps][Cout/32][Cin][32], taps flipped */
int subreg_pack_conv_weight_dgrad(const float* w_oihw, void* out, int Cout, int Cin, int ksize, int dtype, void* stream);
/* dW: gw_packed[splits][Cout][taps][Cin] fp32 = per-K-split partial sums of dY[p][o] * X[p+off(tap)][c]
 * (splits = subreg_conv_wgrad_splits(...) >= 1; the caller sizes gw_packed with it), then OIHW via unpack, which adds
 * the splits up (mode 1: the first layer's K=32 im2col layout back to [Cout][3][k][k]).  pad_x / pad_dy: scratch of
 * B*(H+2)*(W+2)*Cin resp. *Cout elements for the bf16 3x3 kernels (zero-bordered copies); NULL selects the exact-f32
 * MFMA kernel that reads the compact tensors (always used for dtype f32). */
int subreg_conv_wgrad_splits(int B, int H, int W, int Cin, int Cout, int ksize, int dtype);
int subreg_conv_wgrad(const void* x, const void* dy, float* gw_packed, void* pad_x, void* pad_dy, int B, int H, int W, int Cin,
                      int Cout, int ksize, int dtype, void* stream);
int subreg_unpack_wgrad(const float* gw_packed, float* grad_oihw, int Cout, int Cin, int ksize, int mode, int splits, void* stream);
/* BatchNorm2d training-mode backward with the LeakyReLU' of `act` fused in (act == NULL: none):
 * g = dy*lrelu'(act); dgamma = sum g*xhat; dbeta = sum g; dx = gamma*invstd*(g - dbeta/N - xhat*dgamma/N).
 * partial: subreg_bn_bwd_slices(npix)*C*2 DOUBLES of scratch (sums in fp64 like the reference's CPU batch_norm backward) */
int subreg_bn_bwd_slices(long long npix);
int subreg_bn_bwd(const void* dy, const void* act, const void* raw, const float* mean, const float* invstd,
                  const float* gamma, double* partial, float* dgamma, float* dbeta, void* dx, long long npix, int C, int dtype,
                  void* stream);
/* the same for an eval-mode BatchNorm (mean / invstd = the running statistics, constants): dx = gamma * invstd * dy * lrelu'(act) */
int subreg_bn_bwd_eval(const void* dy, const void* act, const void* raw, const float* mean, const float* invstd,
                  const float* gamma, double* partial, float* dgamma, float* dbeta, void* dx, long long npix, int C, int dtype,
                  void* stream);
/* backward of  out = keep*mask_scale * [pool2](lrelu(raw3*scale3+shift3 + residual*res_scale+res_shift))  w.r.t. the
 * pre-activation sum (MaxPool2d: first maximum of the window; BasicBlock.forward :288-299) */
int subreg_block_tail_bwd(const void* grad_out, const unsigned char* keep_mask, float mask_scale,
                          const float* mask_scale_dev /* device override or NULL */, const void* raw3,
                          const float* scale3, const float* shift3, const void* residual, const float* res_scale,
                          const float* res_shift, void* dv, int B, int H, int W, int C, int pool, int dtype, void* stream);
/* the same pass + the REDUCE pass of the BatchNorms that consume dV (BasicBlock's bn3 over raw3 and, with mean_d != NULL, the shortcut's
 * BatchNorm over `residual`): partial3 / partial_d receive *slices (< subreg_bn_bwd_slices()) slices of (sum dV, sum dV * xhat), to be
 * handed to subreg_bn_bwd_partials - two launches and four tensor reads less per block than subreg_block_tail_bwd + 2 x subreg_bn_bwd */
int subreg_block_tail_bwd_stats(const void* grad_out, const unsigned char* keep_mask, float mask_scale,
                                const float* mask_scale_dev, const void* raw3, const float* scale3, const float* shift3,
                                const void* residual, const float* res_scale, const float* res_shift, void* dv, int B, int H, int W,
                                int C, int pool, int dtype, const float* mean3, const float* invstd3, double* partial3,
                                const float* mean_d, const float* invstd_d, double* partial_d, int* slices, void* stream);
/* subreg_bn_bwd / subreg_bn_bwd_eval (eval_mode != 0) without their reduce pass: `partial` holds `slices` slices already */
int subreg_bn_bwd_partials(const void* dy, const void* act, const void* raw, const float* mean, const float* invstd,
                           const float* gamma, double* partial, int slices, float* dgamma, float* dbeta, void* dx, long long npix,
                           int C, int dtype, int eval_mode, void* stream);
int subreg_avgpool_bwd(const float* dfeat, void* dx, int B, int H, int W, int C, int dtype, void* stream);
/* torch.optim.SGD step (train_supervised.py:133-136): d = g + wd*p; buf = first ? d : m*buf + d; p -= lr*buf */
/* the same update for n tensors in ONE launch: params / bufs are DEVICE arrays of n pointers, grad_offsets[n] element offsets
 * of every tensor's gradient inside the flat buffer grad_base, ends[n] the inclusive prefix sums of the tensor sizes
 * (device), total = ends[n-1] (host) */
int subreg_sgd_momentum_multi(float* const* params, float* const* momentum_bufs, const float* grad_base,
                              const long long* grad_offsets, const long long* ends, int n, long long total, float lr,
                              float momentum, float weight_decay, int first_step, void* stream);
/* torch.optim.Adam step (train_supervised.py:128-131 with --adam: Adam(lr, weight_decay 5e-4), betas (0.9, 0.999), eps 1e-8;
 * eval/util.py:92-97 builds the same for the incremental loop): L2 weight decay added to the gradient, bias corrections
 * 1 - beta^step (step >= 1, counted by the caller); exp_avg / exp_avg_sq are the caller's state tensors, zero before step 1 */
int subreg_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                float beta2, float eps, float weight_decay, int step, void* stream);
int subreg_sgd_momentum(float* param, const float* grad, float* momentum_buf, long long n, float lr, float momentum,
                        float weight_decay, int first_step, void* stream);

typedef struct subreg_conv_train { /* per conv; every buffer caller-owned */
    void* raw;            /* [B*H*W][cout] raw conv output (pre-BN) */
    void* act;            /* [B*H*W][cout] BN+LeakyReLU output (conv1, conv2); NULL for conv3 / the shortcut conv */
    float* mean;          /* [cout] batch mean */
    float* invstd;        /* [cout] */
    float* bscale;        /* [cout] gamma*invstd of THIS batch */
    float* bshift;        /* [cout] beta - mean*gamma*invstd */
    const void* w_dgrad;  /* subreg_pack_conv_weight_dgrad output; NULL when no input gradient is needed */
    float* gw_packed;     /* [splits][cout][taps][cin as the kernel sees it] fp32 scratch; the 3x3 convs may share one buffer, the 1x1
                             shortcut convs another (with side_stream their branch can run while a 3x3 dW chain uses the first) */
    float* grad_w;        /* OIHW fp32: Conv2d.weight.grad */
    float* grad_gamma;    /* [cout] */
    float* grad_beta;     /* [cout] */
} subreg_conv_train;

typedef struct subreg_block_train {
    subreg_conv_train conv1, conv2, conv3, down;
    void* out;            /* block output [B*Ho*Wo][cout] (after pool and keep mask) */
} subreg_block_train;

#define SUBREG_TRAIN_EVENTS 6
typedef struct subreg_train_desc {
    const subreg_block_train* blocks; /* HOST array, one per block of the backbone desc */
    void* g[2];            /* activation-sized gradient ping-pong (w.r.t. block outputs / inputs) */
    void* dv;              /* activation-sized scratch x4 */
    void* dr;
    void* dt;
    void* dr2;
    double* bn_partial;    /* subreg_bn_bwd_slices(B*H*W)*Cmax*2 doubles */
    void* pad_x;           /* max over convs of B*(H+2)*(W+2)*Cin elements (bf16 wgrad scratch; may be NULL) */
    void* pad_dy;          /* ... *Cout elements */
    const float* zero_shift; /* [Cmax] zeros */
    void* const* grad_out_dump; /* optional HOST array [n_blocks] of device buffers: receives d(loss)/d(block output)
                                   (NHWC, compute dtype) of every block for diagnostics; NULL = off */
    /* Optional two-stream schedule (all NULL = one stream, the order of train_supervised.py:229-244's autograd graph): the
     * weight-gradient chains and the shortcut branch run on `side_stream` beside the BatchNorm-backward -> dX chain (and the
     * shortcut conv of the forward beside conv1..conv3).  Ordering is by the events only; every call joins the side stream
     * before it returns, so callers keep single-stream semantics on `stream`. */
    void* side_stream;       /* a second hipStream_t of the same device */
    void* events[SUBREG_TRAIN_EVENTS]; /* from subreg_event_create */
    void* dr_alt;            /* activation-sized, like dr (the two alternate) */
    double* bn_partial_side; /* like bn_partial, for the shortcut branch's BatchNorm backward */
    float* stats_side;       /* like subreg_backbone_desc.stats, for the shortcut conv's batch statistics in the forward */
    float* splitk_ws;        /* optional workspace of subreg_conv_fwd_ws for the step's 3x3 convolutions (forward and dX) */
    long long splitk_ws_floats; /* >= max over them of subreg_conv_splitk_floats */
    int eval_mode;           /* 1: the forward normalises with the RUNNING statistics (and updates nothing), the backward is that of an
                              * eval-mode BatchNorm - whole-network fine-tuning with the model in eval mode, i.e. the epochs 2 ..
                              * freeze_backbone_at - 1 of eval/language_eval.py:242-295 (validate() leaves the model in eval mode, :19);
                              * the blocks' keep_mask must be NULL (no dropout / DropBlock in eval mode).  0: train mode */
} subreg_train_desc;

/* hipEvent_t (timing disabled) for subreg_train_desc.events; destroy when the descriptor is retired */
int subreg_event_create(void** event);
int subreg_event_destroy(void* event);

/* train-mode forward that keeps what the backward needs (raw conv outputs, activations, batch statistics) */
int subreg_backbone_forward_stash(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* x_nchw, int B,
                                  int H, int W, float* feat, void* stream);
/* After an optimiser step (train_supervised.py:243-244 `optimizer.step()`): re-pack EVERY conv weight for the next step in
 * ONE launch - the raw copies `w` of `d` (what subreg_backbone_pack_raw writes) and the flipped / transposed dX copies
 * `w_dgrad` of `t` (what subreg_pack_conv_weight_dgrad writes; convs with w_dgrad == NULL are skipped). */
int subreg_backbone_pack_train(const subreg_backbone_desc* d, const subreg_train_desc* t, void* stream);
/* The optimiser step of every conv weight AND its re-packing in ONE launch (train_supervised.py:243-244 `optimizer.step()`,
 * torch.optim.SGD with momentum / weight decay, no dampening, no nesterov): per 32x32-channel tile the fp32 weight, its
 * gradient and its momentum buffer are read once in OIHW order, updated, written back, and the two packed copies the
 * next step's kernels read (raw forward layout `w` of `d`, dX layout `w_dgrad` of `t`) are written from the same tile.
 * Gradients: t->blocks[i].convX.grad_w, taken relative to `grad_origin` and read at the same offset from `grad_base`
 * (the buffer the caller's gradient views live in; pass the same pointer twice when they are the stash's own).
 * Momentum buffers: `mom_base` + the same offsets (one flat fp32 buffer laid out like the gradients).
 * Updates d->blocks[i].convX.w_oihw IN PLACE (the fp32 master weights). */
int subreg_sgd_pack_train(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* grad_base,
                          const float* grad_origin, float* mom_base, float lr, float momentum, float weight_decay,
                          int first_step, void* stream);
/* gradients of every conv weight and BN affine parameter given d(loss)/d(feat) [B][C_last] */
int subreg_backbone_backward(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* dfeat, int B, int H,
                             int W, void* stream);
/* The same backward issued block range by block range: blocks last_block .. first_block (descending).  Call with
 * contiguous, descending ranges that start at n_blocks-1 and end at 0; after each call the gradients of that range's
 * parameters are complete on `stream` - a data-parallel caller starts their all-reduce there while the next range runs
 * (train_supervised.py:141-142 nn.DataParallel reduces gradients; here: one process per GPU over RCCL). */
int subreg_backbone_backward_blocks(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* dfeat, int B,
                                    int H, int W, int first_block, int last_block, void* stream);

/* ---- classifier head: nn.Linear (:138-140,187) ----------------------------------------------------------- */
int subreg_linear_fwd(const float* feat, const float* weight, const float* bias, float* logits, int B, int N, int D,
                      void* stream);
/* any of dweight/dbias/dfeat may be NULL */
int subreg_linear_bwd(const float* dlogits, const float* feat, const float* weight, float* dweight, float* dbias,
                      float* dfeat, int B, int N, int D, void* stream);

/* ---- subspace regularizer: LangPuller.get_projected_weight (:92-97), loss1 (:89-90) --------------------- */
/* rows of q = orthonormal basis of span(rows of w_base) (replaces torch.qr of a CONSTANT matrix every epoch);
 * scratch: n_base*D doubles; *info = number of (numerically) dependent rows */
int subreg_subspace_basis(const float* w_base, float* q, double* scratch, int n_base, int D, int* info, void* stream);
/* p[k][D] = w q^T q  (projection of the novel rows onto the base span; also its own backward) */
int subreg_subspace_project(const float* w, const float* q, float* p, int k, int n_base, int D, void* stream);
/* loss = scale*sum (a-b)^2; grad_a = gscale*grad_out*(a-b), grad_b = -grad_a  (any output may be NULL) */
int subreg_sqdiff(const float* a, const float* b, long long n, float scale, float* loss, const float* grad_out,
                  float gscale, float* grad_a, float* grad_b, void* stream);
/* ResNet.regloss / reglossnovel (:229-240): loss = lmbd*||a-b||_F, grad_a = grad_out*lmbd*(a-b)/||a-b|| (0 at 0) */
int subreg_frob(const float* a, const float* b, long long n, float lmbd, float* loss, const float* grad_out,
                float* grad_a, void* stream);

/* ---- fused fine-tune epoch: eval/language_eval.py:252-318 without a host round trip ---------------------- */
typedef struct subreg_loop_state { /* device resident, one per session */
    int epoch;        /* completed fine-tune epochs */
    int stop;         /* stop rule fired (:305,318): later step launches are no-ops */
    int stable;       /* consecutive |dloss| < eps epochs (:301-304) */
    int val_epoch;    /* epoch whose validation has been recorded */
    float train_loss; /* previous epoch's loss, 15 before the first (:234) */
} subreg_loop_state;

typedef struct subreg_step_desc {
    const float* feat;       /* [n_support+n_memory][dim] backbone features: support rows then memory rows */
    const long long* labels; /* [n_support+n_memory] remapped ids */
    int n_support, n_memory, n_classes, dim;
    float* weight;           /* [n_classes][dim] classifier.weight, updated in place */
    float* momentum_buf;     /* [n_classes][dim] */
    const float* w_base;     /* [n_base][dim] frozen base rows (:106-107) */
    const float* w_prev;     /* [n_prev][dim] reserved earlier novel rows (:172-185) or NULL */
    const float* basis;      /* [n_base][dim] from subreg_subspace_basis or NULL */
    int n_base, n_prev, n_old; /* rows [n_old, n_classes) are this session's novel rows */
    float lr, momentum, weight_decay, lmbd_base, lmbd_prev, pull;
    int use_base_reg, use_prev_reg, use_pull;
    float* dlogits;          /* scratch [n_support+n_memory][n_classes] */
    float* rowloss;          /* scratch [n_support+n_memory] */
    int* rowcorrect;         /* scratch [n_support+n_memory] */
    float* norms;            /* scratch [3] */
    float* rowl1;            /* scratch [n_classes] */
    subreg_loop_state* state;
    float* losses;           /* [max_epochs] loss of every epoch (what the reference prints / the stop rule reads) */
    float* train_acc;        /* [max_epochs] support top-1 % */
    int max_epochs, min_epochs, stable_epochs, stable_mode;
    float target_loss, convergence_eps;
    const float* pull_target; /* [n_classes - n_old][dim] CONSTANT pullers of this session's novel rows (semantic subspace
                               * regularizer / linear mapping, LangPuller.forward :75-87), or NULL: project onto `basis` */
    /* optimiser (eval/util.py:92-102 `get_optim`): 0 = SGD(lr, momentum, weight_decay) with momentum_buf; 1 = torch.optim.Adam
     * (lr, betas, eps, weight_decay as L2 added to the gradient; --adam): momentum_buf is exp_avg, exp_avg_sq its second moment
     * (both zero-initialised per session), the step count is state->epoch + 1 */
    int adam;
    float beta1, beta2, adam_eps;
    float* exp_avg_sq;       /* [n_classes][dim], required with adam */
    /* classifier WITH bias (nn.Linear(640, n, bias=opt.linear_bias), resnet_language.py:140; eval_incremental.py:96-103 takes it
     * from the checkpoint): logits + bias, regloss adds lmbd_base*||bias[:n_base] - bias_base||^2 (:231-232, squared), the bias is
     * updated like any parameter.  All four NULL for a classifier without bias.  bias together with use_prev_reg is refused:
     * reglossnovel (:238) indexes the 1-D bias with two indices and raises in the reference. */
    float* bias;              /* [n_classes] classifier.bias, updated in place */
    float* bias_momentum_buf; /* [n_classes] */
    float* bias_exp_avg_sq;   /* [n_classes], required with adam */
    const float* bias_base;   /* [n_base] frozen base bias (:106-107) */
} subreg_step_desc;

/* the validation of ALL query sets so far (language_eval.py:321-326: one `validate` call over the list of sets) in one
 * launch: the rows of set j follow those of set j-1 in feat / labels; set_rows[n_sets] is a HOST array of row counts;
 * correct[slot*n_sets_max + j] += hits of set j, as subreg_validate does per set; correct_top5 (same layout, may be NULL)
 * counts the rows whose label is among the five largest logits (eval/util.py:26-40, topk=(1, 5)) */
#define SUBREG_MAX_QUERY_SETS 32
int subreg_validate_sets(const float* feat, const long long* labels, const float* weight, const float* bias /* [N] or NULL */,
                         const int* set_rows, int n_sets, int N, int D, subreg_loop_state* state, int* correct, int* correct_top5, int n_sets_max, int mark_done,
                         void* stream);
/* nn.CrossEntropyLoss() (mean) + eval/util.py:26-40 accuracy counters of one batch: rowloss[B] (scratch, required with
 * loss), loss[1] = mean, dlogits[B][N] = (softmax - onehot)/B, correct[0] += #(label is the argmax),
 * correct[1] += #(label within the topk largest).  Every output pointer may be NULL. */
int subreg_softmax_ce(const float* logits, const long long* labels, int B, int N, int topk, float* rowloss, float* loss,
                      float* dlogits, int* correct, void* stream);
/* LangPuller.forward, resnet_language.py:75-83 (semantic subspace regularizer): target[n_novel][dim] =
 * softmax(novel_embeds[n_novel][embed_dim] base_embeds[n_base][embed_dim]^T / temperature, dim=1) @ base_weight[n_base][dim];
 * mask_diagonal: scores.fill_diagonal_(-9999) first (:80-81).  probs [n_novel][n_base] (optional) feeds the backward
 * grad_base_weight = probs^T @ grad_target. */
int subreg_semantic_target(const float* novel_embeds, const float* base_embeds, const float* base_weight, int n_novel, int n_base,
                           int embed_dim, int dim, float temperature, int mask_diagonal, float* probs, float* target,
                           void* stream);
int subreg_semantic_target_bwd(const float* probs, const float* grad_target, int n_novel, int n_base, int dim,
                               float* grad_base_weight, void* stream);
int subreg_loop_state_init(subreg_loop_state* state, void* stream);
int subreg_finetune_step(const subreg_step_desc* d, void* stream);
/* validate (:18-43) / eval_base (:46-69): correct[slot*n_sets_max + set_index] += #(argmax == label), slot =
 * state->epoch (0 when state == NULL); skipped once the stopped loop's last epoch is recorded; mark_done records it */
int subreg_validate(const float* feat, const long long* labels, const float* weight, const float* bias /* [N] or NULL */, int B, int N, int D,
                    subreg_loop_state* state, int* correct, int set_index, int n_sets_max, int mark_done, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUBREG_HIP_H */
