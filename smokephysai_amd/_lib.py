"""ctypes binding of libsmokehip.so (C ABI: include/smokehip.h).  Fails loudly when the library is missing."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMOKEHIP_LIB") or os.path.join(_HERE, "libsmokehip.so")   # SMOKEHIP_LIB: diagnostic builds only

ABI_VERSION = 17                 # include/smokehip.h SMK_ABI_VERSION this binding was written against (tests/test_abi.py holds them equal)
SMK_ERR_TIMEOUT = -5
SMK_F32, SMK_BF16X3, SMK_BF16, SMK_I8X3 = 0, 1, 2, 3
SMK_ACT_NONE, SMK_ACT_GELU, SMK_ACT_RELU = 0, 1, 2
SMK_FMT_F32, SMK_FMT_SPLIT_BF16, SMK_FMT_SPLIT4_INPLACE = 0, 1, 2
STAGE_BUOY_DIFFUSE, STAGE_PROJECT, STAGE_ADVECT_U, STAGE_ADVECT_V, STAGE_ADVECT_D = range(5)
DTYPES = {"f32": SMK_F32, "fp32": SMK_F32, "float32": SMK_F32, "bf16x3": SMK_BF16X3, "bf16": SMK_BF16, "i8x3": SMK_I8X3}


class SmkSimDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("jacobi_iters", C.c_int32),
                ("dt", C.c_double), ("viscosity", C.c_double), ("device_id", C.c_int32),
                ("pitch_c", C.c_int32), ("pitch_v", C.c_int32),
                ("u", C.c_void_p), ("v", C.c_void_p), ("p", C.c_void_p), ("density", C.c_void_p)]


class SmkSource(C.Structure):
    _fields_ = [("grid", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("radius", C.c_int32),
                ("intensity", C.c_double)]


class SmkSim3dDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("depth", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("jacobi_iters", C.c_int32),
                ("dt", C.c_double), ("viscosity", C.c_double), ("device_id", C.c_int32),
                ("pitch_c", C.c_int32), ("pitch_v", C.c_int32),
                ("u", C.c_void_p), ("v", C.c_void_p), ("w", C.c_void_p), ("p", C.c_void_p), ("density", C.c_void_p)]


class SmkSource3d(C.Structure):
    _fields_ = [("grid", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("z", C.c_int32), ("radius", C.c_int32),
                ("intensity", C.c_double)]


class SmkEncoderWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("conv1_w", "conv1_b", "bn1_w", "bn1_b", "bn1_mean", "bn1_var",
                 "conv2_w", "conv2_b", "bn2_w", "bn2_b", "bn2_mean", "bn2_var")]


class SmkDecoderWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("ct1_w", "ct1_b", "bn1_w", "bn1_b", "bn1_mean", "bn1_var",
                 "ct2_w", "ct2_b", "bn2_w", "bn2_b", "bn2_mean", "bn2_var", "conv_w", "conv_b")]


# name -> (argtypes); every function returns int status except the two noted below
_SIGNATURES = {
    "smk_sim_create": [C.POINTER(SmkSimDesc), C.POINTER(C.c_void_p)],
    "smk_sim_destroy": [C.c_void_p],
    "smk_sim_status": [C.c_void_p],
    "smk_sim3d_create": [C.POINTER(SmkSim3dDesc), C.POINTER(C.c_void_p)],
    "smk_sim3d_destroy": [C.c_void_p],
    "smk_sim3d_reset": [C.c_void_p, C.c_char_p, C.c_void_p],
    "smk_sim3d_add_sources": [C.c_void_p, C.POINTER(SmkSource3d), C.c_int32, C.c_void_p],
    "smk_sim3d_step": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p],
    "smk_sim3d_run_stage": [C.c_void_p, C.c_int32, C.c_void_p],
    "smk_conv3d_im2col": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p],
    "smk_conv3d_cl_forward": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p],
    "smk_conv3d_s7_forward": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p],
    "smk_conv3d_cl_zsum_forward": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p],
    "smk_conv3d_s7_march_forward": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p],
    "smk_pool3d_accumulate": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p],
    "smk_sim_reset": [C.c_void_p, C.c_char_p, C.c_void_p],
    "smk_sim_add_sources": [C.c_void_p, C.POINTER(SmkSource), C.c_int32, C.c_void_p],
    "smk_sim_step": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_void_p],
    "smk_sim_run_stage": [C.c_void_p, C.c_int32, C.c_void_p],
    "smk_sim_divergence": [C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_sim_backtrace": [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_sim_fractal": [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)],
    "smk_sim_describe": [C.c_void_p, C.c_char_p, C.c_int64],
    "smk_diffuse": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                    C.c_void_p],
    "smk_apply_fractal": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p],
    "smk_advect": [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                   C.c_int32, C.c_int32, C.c_double, C.c_void_p],
    "smk_fractal_constants": [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_interpolate": [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                        C.c_int64, C.c_int64, C.c_void_p, C.c_void_p],
    "smk_chaos_stats": [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                        C.c_void_p],
    "smk_frame_diff_norms": [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p],
    "smk_encoder_create": [C.POINTER(SmkEncoderWeights), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)],
    "smk_encoder_destroy": [C.c_void_p],
    "smk_encoder_forward": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                            C.c_void_p, C.c_int32, C.c_void_p],
    "smk_encoder_forward_tokens": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_int32, C.c_void_p],
    "smk_encoder_conv1": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p],
    "smk_chaos_addend": [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                         C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_void_p],
    "smk_attention": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                      C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int32, C.c_void_p],
    "smk_chaos_addend_batched": [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p],
    "smk_pooled_head": [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_linear_forward_ln": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p,
                              C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "smk_linear_forward_ln_split": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_void_p,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "smk_attention_ws": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                         C.c_int64, C.c_int64, C.c_double, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p],
    "smk_attention_kv": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                         C.c_int64, C.c_int64, C.c_double, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p],
    "smk_attention_forward_lse": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_void_p],
    "smk_attention_delta": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p],
    "smk_attention_backward": [C.c_void_p] * 9 + [C.c_int32] * 4 + [C.c_int64] * 7 + [C.c_double, C.c_void_p],
    "smk_lorenz_states": [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p],
    "smk_ffn_elementwise": [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_uint64, C.c_void_p],
    "smk_reduce_shards": [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p],
    "smk_conv1_train_forward": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p],
    "smk_conv1_train_wgrad": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_conv2_train_forward": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_conv2_train_dgrad": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_conv2_train_wgrad": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_bn_relu_pool_forward": [C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p, C.c_double, C.c_int32] + [C.c_void_p] * 6,
    "smk_bn_relu_pool_backward": [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 5,
    "smk_bn_relu_pool_phase": [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p, C.c_double] + [C.c_void_p] * 3 +
                              [C.c_int32] + [C.c_void_p] * 4 + [C.c_double, C.c_void_p, C.c_void_p],
    "smk_layernorm": [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int64,
                      C.c_int32, C.c_void_p],
    "smk_layernorm_backward": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_double, C.c_void_p,
                               C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_decoder_create": [C.POINTER(SmkDecoderWeights), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)],
    "smk_decoder_destroy": [C.c_void_p],
    "smk_decoder_forward": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "smk_linear_create": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)],
    "smk_linear_destroy": [C.c_void_p],
    "smk_linear_update": [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p],
    "smk_linear_wgrad": [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                         C.c_int64, C.c_void_p],
    "smk_linear_forward": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                           C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
}
EXPORTS = ["smk_abi_version", "smk_last_error", "smk_linear_wgrad_workspace", "smk_bn_train_workspace", "smk_layernorm_bwd_workspace", "smk_conv2_train_workspace", "smk_conv2_train_wgrad_workspace", "smk_conv1_train_wgrad_workspace", "smk_attention_workspace_bytes", "smk_linear_ln_max_rows"] + list(_SIGNATURES)

_lib = None


class SmkChaosLayer(C.Structure):
    """smk_chaos_layer (include/smokehip.h)."""
    _fields_ = [("noise", C.c_void_p), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("gate_w", C.c_void_p), ("gate_b", C.c_void_p),
                ("addend", C.c_void_p), ("ld_addend", C.c_int64), ("strength", C.c_double)]


class SmokeHipError(RuntimeError):
    pass


def load():
    """dlopen libsmokehip.so (no GPU needed for this); raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C smokephysai_amd/csrc` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). smokephysai_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.smk_abi_version.restype = C.c_int
        L.smk_last_error.restype = C.c_char_p
        have = L.smk_abi_version()
        if have != ABI_VERSION:
            # the .so is git-ignored and travels separately from the sources: argument lists shift between versions, and a stale library
            # would be called with the wrong pointers
            raise ImportError(f"{LIB_PATH} has ABI version {have}, this binding needs {ABI_VERSION}: rebuild it "
                              "(`make -C smokephysai_amd/csrc`, or `python -c 'import __graft_entry__ as g; g.build()'`)")
        L.smk_linear_wgrad_workspace.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        L.smk_linear_wgrad_workspace.restype = C.c_int64          # a byte count, not a status
        L.smk_bn_train_workspace.argtypes = [C.c_int32] * 5
        L.smk_bn_train_workspace.restype = C.c_int64
        L.smk_layernorm_bwd_workspace.argtypes = [C.c_int32]
        L.smk_layernorm_bwd_workspace.restype = C.c_int64
        L.smk_conv2_train_workspace.argtypes = []
        L.smk_conv2_train_workspace.restype = C.c_int64
        L.smk_conv2_train_wgrad_workspace.argtypes = []
        L.smk_conv2_train_wgrad_workspace.restype = C.c_int64
        L.smk_conv1_train_wgrad_workspace.argtypes = []
        L.smk_conv1_train_wgrad_workspace.restype = C.c_int64
        L.smk_attention_workspace_bytes.argtypes = [C.c_int32] * 4
        L.smk_attention_workspace_bytes.restype = C.c_int64
        L.smk_linear_ln_max_rows.argtypes = [C.c_void_p]
        L.smk_linear_ln_max_rows.restype = C.c_int64
        for name, args in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SmokeHipError(f"libsmokehip error {rc}: {load().smk_last_error().decode()}")


def require_cuda(device, what):
    """The product has no CPU path: anything but a ROCm ('cuda') device is an error."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"{what}: device={device!r} is not a ROCm GPU. smokephysai_amd runs its hot path as HIP "
                           "kernels on MI355X only; there is no CPU fallback.")
    if not torch.cuda.is_available():
        raise RuntimeError(f"{what}: no ROCm device is visible (torch.cuda.is_available() is False); "
                           "smokephysai_amd has no CPU fallback.")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
