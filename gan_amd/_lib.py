"""ctypes binding of libganconv.so (the C ABI declared in include/ganconv.h).

The library is built in-tree by gan_amd/csrc/Makefile (see __graft_entry__.build).  There is no fallback: if
the shared object is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

import torch  # noqa: F401  (loads PyTorch's HIP runtime first so that the library binds to the same runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libganconv.so")

GC_F32, GC_BF16 = 0, 1
GC_PREC_BF16, GC_PREC_F32X3 = 0, 1
GC_ACT_NONE, GC_ACT_RELU, GC_ACT_LRELU, GC_ACT_TANH, GC_ACT_SIGMOID = 0, 1, 2, 3, 4
GC_PASS_FPROP, GC_PASS_DGRAD, GC_PASS_WGRAD = 0, 1, 2

EXPORTS = (
    "gc_version", "gc_last_error", "gc_packed_weight_bytes", "gc_pack_weight", "gc_conv2d_fprop", "gc_conv2d_dgrad",
    "gc_wgrad_workspace_bytes", "gc_workspace_bytes", "gc_fused_upsample_weight", "gc_conv2d_wgrad", "gc_wgrad_sums_bias", "gc_conv2d_wgrad_noalloc", "gc_noalloc_scratch", "gc_dwblur3x3", "gc_dwblur3x3_masked", "gc_profile_enable", "gc_profile_collect",
    "gc_instnorm_act_fwd", "gc_instnorm_act_bwd", "gc_instnorm_act_bwd_bwd",
    "gc_batchnorm_workspace_bytes", "gc_batchnorm_act_fwd", "gc_batchnorm_act_fwd_rows", "gc_batchnorm_act_bwd", "gc_batchnorm_update_running", "gc_channel_sum_workspace_bytes", "gc_channel_sum", "gc_act_backward_channel_sum", "gc_batchnorm_stats", "gc_batchnorm_act_bwd_sums", "gc_batchnorm_act_bwd_apply", "gc_adam_step", "gc_adam_ema_step", "gc_set_split_reduction", "gc_set_patch_kernel", "gc_pack_weights", "gc_pack_weights_div", "gc_attention_fwd", "gc_attention_bwd", "gc_attention_residual_fwd", "gc_attention_residual_workspace_bytes", "gc_attention_residual_bwd", "gc_pool2x2", "gc_replicate2x", "gc_pool2x2_add", "gc_replicate2x_add", "gc_reflect_pad", "gc_reflect_pad_adjoint",
    "gc_cond_batchnorm_workspace_bytes", "gc_cond_batchnorm_act_fwd", "gc_cond_batchnorm_act_fwd_rows", "gc_cond_batchnorm_act_bwd", "gc_noise_lrelu_adain_workspace_bytes", "gc_noise_lrelu_adain_fwd", "gc_noise_lrelu_adain_bwd",
    "gc_wgrad_multi_workspace_bytes", "gc_conv2d_wgrad_multi",
    "gc_channel_scale", "gc_rowdot_workspace_bytes", "gc_rowdot", "gc_modulated_act_backward_workspace_bytes",
    "gc_modulated_act_backward", "gc_act_concat_fwd", "gc_act_concat_bwd",
    "gc_wgan_mix", "gc_gradient_penalty_workspace_bytes", "gc_gradient_penalty_fwd", "gc_gradient_penalty_bwd", "gc_wgan_critic_loss", "gc_image_batch_prepare", "gc_spectral_norm_workspace_bytes", "gc_spectral_norm_power_iteration", "gc_spectral_norm_multi_plan", "gc_spectral_norm_multi_run",
)


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "N", "H", "W", "Cin", "Ho", "Wo", "Cout", "kh", "kw", "stride_h", "stride_w", "pad_h", "pad_w",
        "x_dtype", "y_dtype", "precision", "x_cstride", "x_coff", "y_cstride", "y_coff")] + [("reserved", ctypes.c_int32 * 4)]


class Epilogue(ctypes.Structure):
    _fields_ = [("act", ctypes.c_int32), ("slope", ctypes.c_float), ("bias", ctypes.c_void_p), ("ch_sum", ctypes.c_void_p),
                ("ch_sumsq", ctypes.c_void_p), ("out_bf16", ctypes.c_void_p), ("grad_mask", ctypes.c_void_p),
                ("stat_rows", ctypes.c_void_p), ("stat_rows_capacity", ctypes.c_int32), ("stat_rows_written", ctypes.POINTER(ctypes.c_int32)),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("sample_scale", ctypes.c_void_p), ("sample_scale_out", ctypes.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C gan_amd/csrc` (or __graft_entry__.build()); "
            "gan_amd has no fallback convolution path")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64p = ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)
    L.gc_version.restype = ctypes.c_int
    L.gc_set_split_reduction.argtypes = [ctypes.c_int]
    L.gc_set_patch_kernel.argtypes = [ctypes.c_int]
    L.gc_pack_weights.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
    L.gc_pack_weights_div.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.gc_noise_lrelu_adain_workspace_bytes.argtypes = [i32, i32, i32]
    L.gc_noise_lrelu_adain_workspace_bytes.restype = ctypes.c_size_t
    L.gc_noise_lrelu_adain_fwd.argtypes = [vp] * 9 + [ctypes.c_size_t] + [i32] * 4 + [ctypes.c_float, ctypes.c_float, vp]
    L.gc_noise_lrelu_adain_bwd.argtypes = [vp] * 13 + [ctypes.c_size_t] + [i32] * 4 + [ctypes.c_float, vp]
    L.gc_reflect_pad.argtypes = [vp, vp] + [i32] * 6 + [vp]
    L.gc_reflect_pad_adjoint.argtypes = [vp, vp] + [i32] * 6 + [vp]
    L.gc_pool2x2.argtypes = [vp, vp] + [i32] * 5 + [ctypes.c_float, vp]
    L.gc_channel_scale.argtypes = [vp, vp, vp] + [i32] * 4 + [vp]
    L.gc_rowdot_workspace_bytes.argtypes = [i32, i32, i32]
    L.gc_rowdot_workspace_bytes.restype = ctypes.c_size_t
    L.gc_rowdot.argtypes = [vp, vp, vp, vp, ctypes.c_size_t] + [i32] * 4 + [vp]
    L.gc_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_int]
    L.gc_workspace_bytes.restype = ctypes.c_size_t
    L.gc_fused_upsample_weight.argtypes = [vp, vp, i32, i32, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_int, vp]
    L.gc_fused_upsample_weight.restype = ctypes.c_int
    L.gc_attention_residual_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.gc_attention_residual_workspace_bytes.argtypes = [i32, i32, i32]
    L.gc_attention_residual_workspace_bytes.restype = ctypes.c_size_t
    L.gc_attention_residual_bwd.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_size_t, i32, i32, i32, i32, vp]
    L.gc_wgrad_sums_bias.argtypes = [ctypes.POINTER(ConvDesc), vp, vp]
    L.gc_wgrad_sums_bias.restype = ctypes.c_int
    L.gc_modulated_act_backward_workspace_bytes.argtypes = [i32, i32, i32]
    L.gc_modulated_act_backward_workspace_bytes.restype = ctypes.c_size_t
    L.gc_modulated_act_backward.argtypes = [vp] * 8 + [ctypes.c_size_t] + [i32] * 4 + [ctypes.c_float, vp]
    L.gc_act_concat_fwd.argtypes = [vp, vp, vp, ctypes.c_int64, i32, i32, i32, ctypes.c_float, vp]
    L.gc_act_concat_bwd.argtypes = [vp, vp, vp, vp, ctypes.c_int64, i32, i32, i32, ctypes.c_float, vp]
    L.gc_replicate2x.argtypes = [vp, vp] + [i32] * 5 + [ctypes.c_float, vp]
    L.gc_pool2x2_add.argtypes = [vp, vp, vp] + [i32] * 5 + [ctypes.c_float, vp]
    L.gc_replicate2x_add.argtypes = [vp, vp, vp] + [i32] * 5 + [ctypes.c_float, vp]
    L.gc_attention_fwd.argtypes = [vp] * 5 + [i32] * 4 + [vp]
    L.gc_attention_bwd.argtypes = [vp] * 10 + [i32] * 4 + [vp]
    L.gc_last_error.restype = ctypes.c_char_p
    L.gc_packed_weight_bytes.restype = ctypes.c_size_t
    L.gc_packed_weight_bytes.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_int]
    L.gc_pack_weight.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_int, vp, i64p, ctypes.c_float, vp, vp]
    L.gc_conv2d_fprop.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, ctypes.POINTER(Epilogue), vp]
    L.gc_conv2d_dgrad.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, ctypes.POINTER(Epilogue), vp]
    L.gc_wgrad_workspace_bytes.restype = ctypes.c_size_t
    L.gc_wgrad_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
    L.gc_conv2d_wgrad.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, i64p, ctypes.c_float, ctypes.c_int, vp, vp, ctypes.c_size_t, vp]
    L.gc_conv2d_wgrad_noalloc.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, i64p, ctypes.c_float, ctypes.c_int, vp, vp]
    L.gc_noalloc_scratch.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    L.gc_dwblur3x3.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.gc_dwblur3x3_masked.argtypes = [vp, vp, vp, ctypes.c_float, ctypes.c_int, i32, i32, i32, i32, i32, vp]
    f32 = ctypes.c_float
    L.gc_instnorm_act_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, i32, f32, vp]
    L.gc_instnorm_act_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, f32, vp]
    L.gc_instnorm_act_bwd_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, f32, vp]
    i64, sz = ctypes.c_int64, ctypes.c_size_t
    L.gc_batchnorm_workspace_bytes.restype = sz
    L.gc_batchnorm_workspace_bytes.argtypes = [i64, i32]
    L.gc_batchnorm_act_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i64, i32, i32, f32, f32, i32, f32, i32, vp]
    L.gc_batchnorm_act_fwd_rows.argtypes = [vp] * 12 + [i32, vp, sz, i64, i32, i32, f32, f32, i32, f32, vp]
    L.gc_cond_batchnorm_workspace_bytes.restype = sz
    L.gc_cond_batchnorm_workspace_bytes.argtypes = [i32, i32, i32]
    L.gc_cond_batchnorm_act_fwd.argtypes = [vp] * 11 + [sz, i32, i32, i32, i32, f32, f32, i32, f32, i32, vp]
    L.gc_cond_batchnorm_act_fwd_rows.argtypes = [vp] * 11 + [i32, vp, sz, i32, i32, i32, i32, f32, f32, i32, f32, vp]
    L.gc_cond_batchnorm_act_bwd.argtypes = [vp] * 10 + [sz, i32, i32, i32, i32, i32, f32, vp]
    L.gc_batchnorm_update_running.argtypes = [vp, vp, vp, vp, vp, i32, f32, vp]
    L.gc_batchnorm_act_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i64, i32, i32, i32, f32, vp]
    L.gc_wgrad_multi_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc), i32, vp]
    L.gc_wgrad_multi_workspace_bytes.restype = sz
    L.gc_conv2d_wgrad_multi.argtypes = [ctypes.POINTER(ConvDesc), i32, vp, vp, vp, vp, i64p, f32, ctypes.c_int, vp, sz, vp]
    L.gc_wgan_mix.argtypes = [vp, vp, vp, vp, i32, i64, vp]
    L.gc_image_batch_prepare.argtypes = [vp, i32, i32, i32, i32, i32, ctypes.c_float, vp, vp, vp, i32, vp, vp]
    L.gc_spectral_norm_workspace_bytes.argtypes = [i32, i64]
    L.gc_spectral_norm_workspace_bytes.restype = ctypes.c_size_t
    L.gc_spectral_norm_power_iteration.argtypes = [vp, i32, i64, vp, vp, i32, vp, ctypes.c_size_t, vp]
    L.gc_spectral_norm_multi_plan.argtypes = [vp, i32, vp, vp, vp]
    L.gc_spectral_norm_multi_run.argtypes = [vp, i32, vp, i32, vp, vp]
    L.gc_gradient_penalty_workspace_bytes.argtypes = [i32, i64]
    L.gc_gradient_penalty_workspace_bytes.restype = sz
    L.gc_gradient_penalty_fwd.argtypes = [vp, i32, i32, i64, f32, vp, vp, vp, vp, sz, vp]
    L.gc_gradient_penalty_bwd.argtypes = [vp, i32, i32, i64, vp, vp, vp, vp]
    L.gc_wgan_critic_loss.argtypes = [vp, i32, f32, vp, vp, vp]
    L.gc_channel_sum_workspace_bytes.argtypes = [i64, i32]
    L.gc_channel_sum_workspace_bytes.restype = sz
    L.gc_channel_sum.argtypes = [vp, vp, vp, sz, i64, i32, i32, vp]
    L.gc_channel_sum.restype = ctypes.c_int
    L.gc_act_backward_channel_sum.argtypes = [vp, vp, vp, vp, vp, sz, i64, i32, i32, i32, f32, vp]
    L.gc_act_backward_channel_sum.restype = ctypes.c_int
    L.gc_batchnorm_stats.argtypes = [vp, vp, vp, vp, sz, i64, i32, i32, vp]
    L.gc_batchnorm_stats.restype = ctypes.c_int
    L.gc_batchnorm_act_bwd_sums.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i64, i32, i32, i32, f32, vp]
    L.gc_batchnorm_act_bwd_sums.restype = ctypes.c_int
    L.gc_batchnorm_act_bwd_apply.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, f32, vp]
    L.gc_batchnorm_act_bwd_apply.restype = ctypes.c_int
    L.gc_adam_step.argtypes = [i32, vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, vp]
    L.gc_adam_step.restype = ctypes.c_int
    L.gc_adam_ema_step.argtypes = [i32, vp, vp, vp, vp, vp, f32, vp, vp, f32, f32, f32, f32, vp]
    L.gc_adam_ema_step.restype = ctypes.c_int
    L.gc_profile_enable.argtypes = [ctypes.c_int]
    L.gc_profile_collect.argtypes = [ctypes.c_int, vp, vp, vp, vp]
    for fn in ("gc_pack_weight", "gc_conv2d_fprop", "gc_conv2d_dgrad", "gc_conv2d_wgrad", "gc_conv2d_wgrad_noalloc", "gc_dwblur3x3", "gc_dwblur3x3_masked", "gc_profile_enable",
               "gc_profile_collect", "gc_instnorm_act_fwd", "gc_instnorm_act_bwd", "gc_instnorm_act_bwd_bwd", "gc_batchnorm_act_fwd",
               "gc_batchnorm_act_bwd", "gc_batchnorm_update_running"):
        getattr(L, fn).restype = ctypes.c_int
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().gc_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"ganconv: {what} failed (status {rc}): {msg}")


def strides4(t):
    return (ctypes.c_int64 * 4)(*[int(s) for s in t.stride()])


def profile_enable(on):
    check(lib().gc_profile_enable(1 if on else 0), "gc_profile_enable")


def profile_collect(max_entries=512):
    """Timed launches since profile_enable(True), summed per kernel name (names beyond max_entries are dropped by the
    library; a training step of the larger harnesses launches about 50 different kernels)."""
    names = ((ctypes.c_char * 96) * max_entries)()
    launches = (ctypes.c_int64 * max_entries)()
    total_ms = (ctypes.c_double * max_entries)()
    flops = (ctypes.c_double * max_entries)()
    n = lib().gc_profile_collect(max_entries, ctypes.cast(names, ctypes.c_void_p), ctypes.cast(launches, ctypes.c_void_p),
                                 ctypes.cast(total_ms, ctypes.c_void_p), ctypes.cast(flops, ctypes.c_void_p))
    if n < 0:
        check(n, "gc_profile_collect")
    return {names[i].value.decode(): {"launches": int(launches[i]), "total_ms": float(total_ms[i]), "flops": float(flops[i])}
            for i in range(n)}
