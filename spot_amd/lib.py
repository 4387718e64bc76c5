"""ctypes binding of libspot_hip.so (the C ABI declared in include/spot_hip.h).

The library is built in-tree by ``make -C spot_amd/csrc`` (hipcc, gfx950) or by
``__graft_entry__.build()``.  There is no fallback: if the shared object is
missing or a symbol cannot be resolved, importing the product ops fails loudly.
"""
import ctypes as C
import os

# PyTorch ships its own HIP runtime (same soname as the system one).  It must be the first one
# mapped into the process so that the kernels launched from this library and torch's allocator /
# streams share a single runtime and device context; loading the system runtime first leaves
# torch without a device.
import torch  # noqa: F401  (side effect: loads torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspot_hip.so")

vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double


class GemmDesc(C.Structure):
    """Mirror of ``spot_gemm_desc``."""
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp),
        ("M", i32), ("N", i32), ("K", i32), ("batch", i32),
        ("transA", i32), ("transB", i32),
        ("lda", i64), ("ldb", i64), ("ldc", i64),
        ("strideA", i64), ("strideB", i64), ("strideC", i64),
        ("kseg", i32), ("segA", i64), ("segB", i64),
        ("splitk", i32), ("alpha", f32), ("accumulate", i32),
        ("bias", vp), ("rowbias", vp), ("rb_rows", i32), ("ldrb", i64),
        ("relu", i32), ("mask", vp), ("ldmask", i64),
        ("residual", vp), ("ldres", i64), ("strideRes", i64),
        ("drop_p", f32), ("drop_seed", C.c_uint), ("tile", i32), ("rb_cols", i32), ("bn_gamma", vp), ("bn_beta", vp), ("bn_mean", vp), ("bn_var", vp), ("bn_eps", f32),
        ("rowsum", vp), ("precision", i32),
    ]


class CropDesc(C.Structure):
    """spot_crop_desc of include/spot_hip.h."""
    _fields_ = [("image", C.c_void_p), ("H", C.c_int), ("W", C.c_int), ("x0", C.c_int), ("y0", C.c_int),
                ("crop", C.c_int), ("pad", C.c_ubyte * 4)]


class BnTrainIn(C.Structure):
    """Mirror of ``spot_bn_train_in``."""
    _fields_ = [("stats_ws", vp), ("gamma", vp), ("beta", vp), ("running_mean", vp), ("running_var", vp),
                ("num_batches_tracked", vp), ("save_mean", vp), ("save_rstd", vp), ("relu_sign", vp),
                ("deferred_stats", vp), ("eps", f32), ("momentum", f32)]


class WinogradOpts(C.Structure):
    """Mirror of ``spot_winograd_opts``."""
    _fields_ = [("v_keep", vp), ("u_keep", vp), ("u_valid", i32), ("bn_gamma", vp), ("bn_beta", vp), ("bn_mean", vp),
                ("bn_var", vp), ("bn_eps", f32), ("relu", i32), ("in_bn", C.POINTER(BnTrainIn))]


# name -> argument ctypes (all return int unless noted); the trailing hipStream_t is a void*
SIGNATURES = {
    "spot_preprocess_u8_f32": [vp, vp, C.POINTER(f32), C.POINTER(f32), vp, i32, i32, i32, i32, vp],
    "spot_posenc_sine_f32": [vp, vp, i32, i32, i32, i32, f32, f32, vp],
    "spot_decode_teacher_f32": [vp, vp, vp, vp, f64, i32, i32, vp, vp, vp, vp, i32, i32, vp],
    "spot_middleware_filter_f32": [vp, vp, vp, vp, vp, vp, i32, f32, vp, vp, vp, i32, vp],
    "spot_decode_filter_f32": [vp, vp, vp, vp, f64, i32, i32, vp, vp, vp, vp, i32, f32,
                               vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "spot_compact_indices": [vp, i32, vp, i32, vp, vp],
    "spot_gather_rows": [vp, vp, vp, i32, i64, vp],
    "spot_tracker_decode": [vp, vp, vp, f64, vp, vp, i32, i32, vp],
    "spot_match_masks": [vp, i32, i32, i32, vp, vp, vp],
    "spot_ema_lerp_f32": [vp, vp, i64, f64, vp],
    "spot_ema_lerp_i64": [vp, vp, i32, f64, vp],
    "spot_box_iou_pair_f32": [vp, vp, i32, vp, vp, vp, i32, vp],
    "spot_box_giou_pair_bwd_f32": [vp, vp, vp, vp, vp, i32, vp],
    "spot_transt_loss_f32": [vp] * 6 + [f32, f32, i32, i32, vp, vp, vp, vp],
    "spot_transt_loss_bwd_f32": [vp] * 6 + [f32, f32] + [vp] * 5 + [i32, i32, vp, vp, vp, vp],
    "spot_xcorr_depthwise_f32": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "spot_groupdw_f32": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "spot_xcorr_depthwise_bwd_f32": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "spot_groupdw_bwd_f32": [vp] * 15 + [i32, i32, i32, i32, i32, vp],
    "spot_batchnorm_fwd_train_f32": [vp] * 13 + [i32, i32, i32, f32, f32, i32, vp],
    "spot_batchnorm_apply_deferred": [vp, i32, vp],
    "spot_batchnorm_fwd_infer_f32": [vp] * 7 + [i32, i32, i32, f32, i32, vp],
    "spot_batchnorm_bwd_f32": [vp] * 11 + [i32, i32, i32, i32, vp],
    "spot_adamw_f32": [vp, vp, vp, vp, vp, i32, vp, vp],
    "spot_clip_grad_norm_f32": [vp, i64, f32, vp, vp, vp],
    "spot_gemm_f32": [C.POINTER(GemmDesc), vp],
    "spot_colsum_f32": [vp, i64, vp, i64, i32, vp],
    "spot_batch_sum_f32": [vp, i64, i32, i64, i32, vp, vp],
    "spot_channel_sum_f32": [vp, i32, i32, i64, vp, vp],
    "spot_groupdw_weight_grad_f32": [vp, i32, vp, vp, vp],
    "spot_im2col_f32": [vp, vp] + [i32] * 12 + [vp],
    "spot_col2im_f32": [vp, vp] + [i32] * 12 + [vp],
    "spot_rotate_flip_nearest_f32": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "spot_ellipse_rotate_boxes_f32": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "spot_channel_scale_f32": [vp, vp, i64, i32, vp],
    "spot_ocean_labels_f32": [vp] * 5 + [i32] * 4 + [f32, i32, vp],
    "spot_ocean_tracker_decode": [vp] * 5 + [f64] * 3 + [i32] * 3 + [vp, vp],
    "spot_conv3x3_winograd_general_f32": [vp] * 6 + [i32] * 9 + [vp],
    "spot_conv3x3_winograd_general_wgrad_f32": [vp] * 5 + [i32] * 9 + [vp],
    "spot_crop_resize_u8": [vp, vp, i32, vp],
    "spot_crop_resize_batch_u8": [vp, i32, vp, i32, vp],
    "spot_deform_im2col_f32": [vp] * 3 + [i32] * 12 + [vp],
    "spot_deform_col2im_f32": [vp] * 6 + [i32] * 12 + [vp],
    "spot_maxpool3x3s2_fwd_f32": [vp, vp, vp, i64, i32, i32, vp],
    "spot_maxpool3x3s2_bwd_f32": [vp, vp, vp, i64, i32, i32, vp],
    "spot_conv3x3_winograd_f32": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "spot_conv3x3_winograd_wgrad_f32": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "spot_mha_fwd_f32": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, i32, f32, f32, C.c_uint, vp, vp],
    "spot_mha_fwd_splitk_f32": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp] + [i32] * 6 + [f32, vp],
    "spot_mha_fwd_splitk_pair_f32": ([vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i32, i32, i32] * 2) + [i32, i32, f32, vp],
    "spot_layernorm_fwd_pair_f32": [vp, vp, vp, vp, i64, f32, vp, vp, vp, vp, i64, f32, i32, vp],
    "spot_gemm_pair_f32": [vp, vp, vp],
    "spot_mha_fwd_splitk_noalloc_f32": [vp, i64, vp, i64, vp, i64, vp, i64, vp] + [i32] * 6 + [f32, vp],
    "spot_mha_bwd_f32": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64,
                         i32, i32, i32, i32, i32, f32, f32, C.c_uint, vp, vp],
    "spot_mha_fwd_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, i32, f32, f32, C.c_uint, vp, vp],
    "spot_mha_bwd_bf16": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64,
                         i32, i32, i32, i32, i32, f32, f32, C.c_uint, vp, vp],
    "spot_mha_dropout_mask": [vp, i32, i32, i32, i32, f32, C.c_uint, vp],
    "spot_dropout_f32": [vp, vp, i64, i32, f32, C.c_uint, vp],
    "spot_layernorm_fwd_f32": [vp, vp, vp, vp, vp, vp, i64, i32, f32, vp],
    "spot_batchnorm_stats_f32": [vp, vp, i32, i32, i32, vp],
    "spot_layernorm_bwd_f32": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, f32, C.c_uint, vp],
}

_lib = None


def load():
    """Load the shared library and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "spot_amd: %s is missing. Build it with `make -C spot_amd/csrc` (hipcc, --offload-arch=gfx950); "
            "there is no CPU fallback for the hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.spot_version.restype = i32
    lib.spot_version.argtypes = []
    lib.spot_last_error.restype = C.c_char_p
    lib.spot_last_error.argtypes = []
    lib.spot_layernorm_bwd_ws_floats.restype = i64
    lib.spot_layernorm_bwd_ws_floats.argtypes = [i64, i32]
    lib.spot_batchnorm_ws_floats.restype = i64
    lib.spot_batchnorm_ws_floats.argtypes = [i32, i32]
    lib.spot_conv3x3_winograd_ws_floats.restype = i64
    lib.spot_conv3x3_winograd_ws_floats.argtypes = [i32, i32, i32, i32, i32]
    lib.spot_conv3x3_winograd_wgrad_ws_floats.restype = i64
    lib.spot_conv3x3_winograd_wgrad_ws_floats.argtypes = [i32, i32, i32, i32, i32]
    lib.spot_conv3x3_winograd_general_ws_floats.restype = i64
    lib.spot_conv3x3_winograd_general_ws_floats.argtypes = [i32] * 8
    lib.spot_conv3x3_winograd_general_v_floats.restype = i64
    lib.spot_conv3x3_winograd_general_v_floats.argtypes = [i32] * 7
    lib.spot_mha_fwd_splitk_ws_floats.restype = i64
    lib.spot_mha_fwd_splitk_ws_floats.argtypes = [i32] * 4
    lib.spot_mha_drop_bits_words.restype = i64
    lib.spot_mha_drop_bits_words.argtypes = [i32] * 4
    lib.spot_transt_loss_ws_floats.restype = i64
    lib.spot_transt_loss_ws_floats.argtypes = []
    lib.spot_clip_grad_norm_ws_floats.restype = i64
    lib.spot_clip_grad_norm_ws_floats.argtypes = []
    lib.spot_deform_col2im_ws_floats.restype = i64
    lib.spot_deform_col2im_ws_floats.argtypes = [i32] * 10
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.restype = i32
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().spot_last_error().decode("utf-8", "replace")
        raise RuntimeError("spot_hip %s failed (%d): %s" % (what, rc, msg))


_recording = None     # a list while launches are being collected instead of issued (see record / run_paired)
_keeping = None       # the tensors whose addresses the collected launches carry (kept alive until they are issued)


def keep(*tensors):
    """While launches are being collected, hold on to the tensors they address: a buffer released before its launch
    goes out could be handed to another collected launch, and the two would then run on the same memory."""
    if _keeping is not None:
        _keeping.extend(t for t in tensors if t is not None)


def call(name, *args):
    """Call a C-ABI entry point and raise on a non-zero status."""
    if _recording is not None:
        _recording.append((name, args))
        return
    rc = getattr(_lib or load(), name)(*args)
    if rc != 0:
        check(rc, name)


class record:
    """Collects the library calls made inside the block instead of issuing them; `.calls` holds (name, args) in order.
    The block must consist of library launches and allocations only: anything else the framework runs on the device
    inside it would execute before the collected launches.  The object keeps the buffers of its calls alive: hold it
    until the calls have been issued."""

    def __enter__(self):
        global _recording, _keeping
        self._prev, self.calls, self.tensors = (_recording, _keeping), [], []
        _recording, _keeping = self.calls, self.tensors
        return self

    def __exit__(self, *exc):
        global _recording, _keeping
        _recording, _keeping = self._prev
        return False


def _pair_gemm(a, b):
    return "spot_gemm_pair_f32", (a[0], b[0], a[1])


def _pair_layernorm(a, b):
    # (x, gamma, beta, y, mean, rstd, rows, C, eps, stream): inference calls keep no statistics
    if a[4] is not None or a[5] is not None or b[4] is not None or b[5] is not None or a[7] != b[7]:
        return None
    return "spot_layernorm_fwd_pair_f32", (a[0], a[1], a[2], a[3], a[6], a[8], b[0], b[1], b[2], b[3], b[6], b[8], a[7], a[9])


def _pair_mha_splitk(a, b):
    # (q, ldq, k, ldk, v, ldv, out, ldo, kpm, ws, nsplit, B, Lq, Lk, nhead, head_dim, scale, stream)
    if a[11] != 1 or b[11] != 1 or a[14:17] != b[14:17]:
        return None
    return "spot_mha_fwd_splitk_pair_f32", (a[:11] + (a[12], a[13]) + b[:11] + (b[12], b[13]) + (a[14], a[15], a[16], a[17]))


_PAIRS = {"spot_gemm_f32": _pair_gemm, "spot_layernorm_fwd_f32": _pair_layernorm, "spot_mha_fwd_splitk_f32": _pair_mha_splitk}


def run_paired(calls_a, calls_b):
    """Issue two recorded call sequences side by side: calls at the same position that name the same entry point go
    out through its two-problem variant (one launch where the library can), everything else one after the other."""
    n = min(len(calls_a), len(calls_b))
    for (na, aa), (nb, ab) in zip(calls_a[:n], calls_b[:n]):
        merged = _PAIRS[na](aa, ab) if na == nb and na in _PAIRS else None
        if merged is not None:
            call(*((merged[0],) + tuple(merged[1])))
        else:
            call(na, *aa)
            call(nb, *ab)
    for name, args in calls_a[n:] + calls_b[n:]:
        call(name, *args)
