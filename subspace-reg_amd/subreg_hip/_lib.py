"""ctypes binding of libsubreg_hip.so (include/subreg_hip.h).

The library is built in-tree (subspace-reg_amd/Makefile, hipcc --offload-arch=gfx950) and
loaded lazily.  There is NO fallback: if the .so is missing or a call returns a negative
code, a RuntimeError is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SUBREG_LIB: another build of the same library (kernel experiments: several variants measured in one process group)
LIB_PATH = os.environ.get("SUBREG_LIB") or os.path.join(_HERE, "libsubreg_hip.so")

F32, BF16 = 0, 1
CONV_LRELU, CONV_POOL2, CONV_RAW_STATS = 1, 2, 4
CONV_KERNEL_GENERAL, CONV_KERNEL_WIDE = 256, 512   # kernel selection of subreg_conv_fwd (Cout % 160 == 0): conv_fwd.hip / conv_wide.hip
CONV_KERNEL_WIDE_ALT = 1024                        # with _WIDE: the other MFMA shape of conv_wide.hip than its default for the problem
CONV_KERNEL_WIDE_128, CONV_KERNEL_WIDE_256 = 2048, 4096   # with _WIDE: the 128- / 256-row tiling of conv_wide16_kernel
FWD_TRAIN = 1
ABI_VERSION = 14
MAX_QUERY_SETS = 32                          # SUBREG_MAX_QUERY_SETS

c_void_p, c_int, c_float, c_longlong = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class ConvDesc(C.Structure):
    _fields_ = [("w", c_void_p), ("w_folded", c_void_p), ("w_oihw", c_void_p), ("bn_weight", c_void_p),
                ("bn_bias", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p), ("scale", c_void_p),
                ("shift", c_void_p), ("cin", c_int), ("cout", c_int), ("ksize", c_int), ("cin_raw", c_int),
                ("ksize_raw", c_int)]


class BlockDesc(C.Structure):
    _fields_ = [("conv1", ConvDesc), ("conv2", ConvDesc), ("conv3", ConvDesc), ("down", ConvDesc),
                ("w_identity", c_void_p), ("shift3", c_void_p), ("stride", c_int), ("keep_mask", c_void_p),
                ("mask_scale", c_float), ("mask_scale_dev", c_void_p)]


class BackboneDesc(C.Structure):
    _fields_ = [("n_blocks", c_int), ("blocks", C.POINTER(BlockDesc)), ("dtype", c_int), ("col", c_void_p),
                ("ws", c_void_p * 4), ("stats", c_void_p), ("bn_eps", c_float), ("bn_momentum", c_float)]


class ConvTrain(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("raw", "act", "mean", "invstd", "bscale", "bshift", "w_dgrad", "gw_packed", "grad_w",
                                         "grad_gamma", "grad_beta")]


class BlockTrain(C.Structure):
    _fields_ = [("conv1", ConvTrain), ("conv2", ConvTrain), ("conv3", ConvTrain), ("down", ConvTrain), ("out", c_void_p)]


class TrainDesc(C.Structure):
    _fields_ = [("blocks", C.POINTER(BlockTrain)), ("g", c_void_p * 2), ("dv", c_void_p), ("dr", c_void_p), ("dt", c_void_p),
                ("dr2", c_void_p), ("bn_partial", c_void_p), ("pad_x", c_void_p), ("pad_dy", c_void_p),
                ("zero_shift", c_void_p), ("grad_out_dump", C.POINTER(c_void_p)),
                ("side_stream", c_void_p), ("events", c_void_p * 6), ("dr_alt", c_void_p), ("bn_partial_side", c_void_p),
                ("stats_side", c_void_p), ("splitk_ws", c_void_p), ("splitk_ws_floats", c_longlong), ("eval_mode", c_int)]


class MaskParam(C.Structure):
    _fields_ = [("seed", C.c_ulonglong), ("p_drop", c_float), ("reserved", C.c_uint)]


MASK_PARAMS_MAX = 16                          # SUBREG_MASK_PARAMS_MAX


class LoopState(C.Structure):
    _fields_ = [("epoch", c_int), ("stop", c_int), ("stable", c_int), ("val_epoch", c_int), ("train_loss", c_float)]


class StepDesc(C.Structure):
    _fields_ = [("feat", c_void_p), ("labels", c_void_p),
                ("n_support", c_int), ("n_memory", c_int), ("n_classes", c_int), ("dim", c_int),
                ("weight", c_void_p), ("momentum_buf", c_void_p), ("w_base", c_void_p), ("w_prev", c_void_p),
                ("basis", c_void_p), ("n_base", c_int), ("n_prev", c_int), ("n_old", c_int),
                ("lr", c_float), ("momentum", c_float), ("weight_decay", c_float), ("lmbd_base", c_float),
                ("lmbd_prev", c_float), ("pull", c_float),
                ("use_base_reg", c_int), ("use_prev_reg", c_int), ("use_pull", c_int),
                ("dlogits", c_void_p), ("rowloss", c_void_p), ("rowcorrect", c_void_p), ("norms", c_void_p),
                ("rowl1", c_void_p), ("state", c_void_p), ("losses", c_void_p), ("train_acc", c_void_p),
                ("max_epochs", c_int), ("min_epochs", c_int), ("stable_epochs", c_int), ("stable_mode", c_int),
                ("target_loss", c_float), ("convergence_eps", c_float), ("pull_target", c_void_p),
                ("adam", c_int), ("beta1", c_float), ("beta2", c_float), ("adam_eps", c_float), ("exp_avg_sq", c_void_p),
                ("bias", c_void_p), ("bias_momentum_buf", c_void_p), ("bias_exp_avg_sq", c_void_p), ("bias_base", c_void_p)]


# name -> (restype, argtypes); every symbol include/subreg_hip.h declares
_P, _I, _F, _L = c_void_p, c_int, c_float, c_longlong
SIGNATURES = {
    "subreg_abi_version": (_I, []),
    "subreg_strerror": (C.c_char_p, [_I]),
    "subreg_pack_input": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "subreg_pack_conv_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "subreg_pack_identity": (_I, [_P, _I, _I, _P]),
    "subreg_vec_add": (_I, [_P, _P, _P, _I, _P]),
    "subreg_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "subreg_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "subreg_conv_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "subreg_conv_fwd_ws": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "subreg_conv_stats_rows": (_I, [_I, _I, _I, _I, _I]),
    "subreg_conv_first_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "subreg_conv_fwd_image_shortcut": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "subreg_conv12_first_fused": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "subreg_layer1_direct_supported": (_I, [_I, _I, _I, _I]),
    "subreg_bn_fold": (_I, [_P, _P, _P, _P, _P, _P, _I, _F, _P]),
    "subreg_bn_train_finalize": (_I, [_P, _I, _I, _L, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "subreg_pack_conv_weight_dgrad": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "subreg_conv_wgrad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "subreg_conv_wgrad_splits": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "subreg_validate_sets": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _P]),
    "subreg_softmax_ce": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "subreg_semantic_target": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P, _P]),
    "subreg_semantic_target_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "subreg_unpack_wgrad": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "subreg_bn_bwd_slices": (_I, [_L]),
    "subreg_bn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "subreg_bn_bwd_eval": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "subreg_bn_eval_stash": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P]),
    "subreg_block_tail_bwd": (_I, [_P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "subreg_block_tail_bwd_stats": (_I, [_P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "subreg_bn_bwd_partials": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _I, _I, _I, _P]),
    "subreg_avgpool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "subreg_sgd_momentum": (_I, [_P, _P, _P, _L, _F, _F, _F, _I, _P]),
    "subreg_adam": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P]),
    "subreg_backbone_forward_stash": (_I, [C.POINTER(BackboneDesc), C.POINTER(TrainDesc), _P, _I, _I, _I, _P, _P]),
    "subreg_conv_splitk_floats": (_L, [_I, _I, _I, _I, _I, _I, _I]),
    "subreg_event_create": (_I, [C.POINTER(c_void_p)]),
    "subreg_event_destroy": (_I, [_P]),
    "subreg_backbone_backward": (_I, [C.POINTER(BackboneDesc), C.POINTER(TrainDesc), _P, _I, _I, _I, _P]),
    "subreg_backbone_backward_blocks": (_I, [C.POINTER(BackboneDesc), C.POINTER(TrainDesc), _P, _I, _I, _I, _I, _I, _P]),
    "subreg_backbone_pack_train": (_I, [C.POINTER(BackboneDesc), C.POINTER(TrainDesc), _P]),
    "subreg_sgd_pack_train": (_I, [C.POINTER(BackboneDesc), C.POINTER(TrainDesc), _P, _P, _P, _F, _F, _F, _I, _P]),
    "subreg_bn_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "subreg_mask_scale": (_I, [_P, _L, _P, _P]),
    "subreg_mask_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "subreg_random_keep_mask": (_I, [_P, _L, C.c_ulonglong, _F, _P, _P]),
    "subreg_random_keep_mask_dev": (_I, [_P, _L, _P, _P, _P]),
    "subreg_mask_params_set": (_I, [_P, _I, _P, _P]),
    "subreg_dropblock_mask": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "subreg_avgpool": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "subreg_backbone_ws_bytes": (_L, [C.POINTER(BackboneDesc), _I, _I, _I]),
    "subreg_backbone_needs_col": (_I, [C.POINTER(BackboneDesc), _I, _I, _I, _I]),
    "subreg_backbone_stats_floats": (_L, [C.POINTER(BackboneDesc), _I, _I, _I]),
    "subreg_backbone_fold": (_I, [C.POINTER(BackboneDesc), _P]),
    "subreg_backbone_pack_raw": (_I, [C.POINTER(BackboneDesc), _P]),
    "subreg_sgd_momentum_multi": (_I, [_P, _P, _P, _P, _P, _I, _L, _F, _F, _F, _I, _P]),
    "subreg_backbone_forward": (_I, [C.POINTER(BackboneDesc), _P, _I, _I, _I, _P, C.POINTER(_P), _I, _P]),
    "subreg_linear_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "subreg_linear_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "subreg_subspace_basis": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "subreg_subspace_project": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "subreg_sqdiff": (_I, [_P, _P, _L, _F, _P, _P, _F, _P, _P, _P]),
    "subreg_frob": (_I, [_P, _P, _L, _F, _P, _P, _P, _P]),
    "subreg_loop_state_init": (_I, [_P, _P]),
    "subreg_finetune_step": (_I, [C.POINTER(StepDesc), _P]),
    "subreg_validate": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _I, _P]),
}

_lib = None


def build(verbose=False):
    """Compile every HIP source for gfx950 into libsubreg_hip.so (hipcc cross-compiles without a GPU)."""
    root = os.path.dirname(_HERE)
    r = subprocess.run(["make", "-C", root, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError("building libsubreg_hip.so failed")
    return LIB_PATH


def load():
    """Load the HIP library (once).  Raises if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsubreg_hip.so not found at %s - run `make -C subspace-reg_amd` "
                           "(or __graft_entry__.build()); the HIP path has no fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.subreg_abi_version() != ABI_VERSION:
        raise RuntimeError("libsubreg_hip.so ABI %d != binding %d" % (lib.subreg_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().subreg_strerror(int(rc)).decode()
        raise RuntimeError("libsubreg_hip %s failed: %s (code %d)" % (what, msg, rc))


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def dtype_code(name):
    if name in (F32, "f32", "fp32", "float32"):
        return F32
    if name in (BF16, "bf16", "bfloat16"):
        return BF16
    raise ValueError("dtype must be 'f32' or 'bf16', got %r" % (name,))
