"""ctypes binding of libvmhip.so (the C ABI in include/vmhip.h).

The product path has NO fallback: if the library is missing or a call fails,
an exception is raised.  PyTorch is used only to own device memory and streams.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvmhip.so")

VM_F32, VM_BF16 = 0, 1


class GemmEpilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p), ("act", C.c_int), ("aux_out", C.c_void_p), ("mul_gelu_z", C.c_void_p),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64), ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("alpha", C.c_float), ("alpha_dev", C.c_void_p), ("out_dtype", C.c_int), ("accumulate", C.c_int), ("split_k", C.c_int),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("dropout_seed_dev", C.c_void_p),
    ]


class WgradProblem(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("ld_dy", C.c_int64), ("X", C.c_void_p), ("ld_x", C.c_int64), ("dW", C.c_void_p), ("ld_dw", C.c_int64),
                ("db", C.c_void_p), ("rows", C.c_int), ("n_out", C.c_int), ("k_in", C.c_int), ("alpha_dev", C.c_void_p), ("overwrite", C.c_int)]


class GemmProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int)]


class DecodeGemmArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("A", C.c_void_p), ("lda", C.c_int64), ("W", C.c_void_p), ("ldw", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("c2", C.c_void_p), ("ldc2", C.c_int64), ("split_n", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("bias", C.c_void_p),
                ("act", C.c_int), ("residual", C.c_void_p), ("ldr", C.c_int64), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
                ("ln_out", C.c_void_p), ("ln_out_ld", C.c_int64)]


class LnReduceProblem(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int)]


_lib = None

_P, _I, _L, _F, _U64, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_size_t

# name -> (restype, argtypes).  Must list every symbol declared in include/vmhip.h (tests check this).
SIGNATURES = {
    "vm_last_error": (C.c_char_p, []),
    "vm_version": (_I, []),
    "vm_build_digest": (C.c_char_p, []),
    "vm_prof_enable": (_I, [_I]),
    "vm_prof_reset": (_I, []),
    "vm_prof_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "vm_reload_env": (None, []),
    "vm_sizeof_gemm_epilogue": (_I, []),
    "vm_prof_dump": (_I, [C.c_char_p]),
    "vm_gemm_bf16": (_I, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, C.POINTER(GemmEpilogue), _P]),
    "vm_wgrad_grouped": (_I, [C.POINTER(WgradProblem), _I, _P]),
    "vm_gemm_grouped": (_I, [_P, _I, _I, _I, _I, _I, _P]),
    "vm_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "vm_layernorm_bwd_ws": (_SZ, [_I, _I]),
    "vm_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "vm_layernorm_bwd_fused": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "vm_layernorm_bwd_partial": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "vm_layernorm_bwd_partial_dropout": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _P, _I, _I, _P, _P]),
    "vm_layernorm_bwd_reduce": (_I, [_P, _P, _P, _I, _I, _P]),
    "vm_layernorm_bwd_reduce_batched": (_I, [_P, _I, _P]),
    "vm_image_pipeline_ws": (_SZ, [_I, _I, _I]),
    "vm_image_pipeline_u8": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _SZ, _P]),
    "vm_attention_fwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _F, _I, _F, _U64, _P, _P, _L, _P]),
    "vm_attention_bwd": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _P, _L, _P, _L,
                              _I, _I, _I, _I, _I, _F, _I, _F, _U64, _P, _P, _P]),
    "vm_embedding_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vm_embedding_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vm_batchnorm_nhwc_ws": (_SZ, [_I, _I, _I]),
    "vm_batchnorm_nhwc_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _I, _I, _P, _SZ, _P]),
    "vm_batchnorm_nhwc_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "vm_batchnorm_nhwc_stats": (_I, [_P, _L, _P, _L, _P, _P, _P, _I, _P, _I, _I, _I, _F, _I, _P, _SZ, _P]),
    "vm_batchnorm_nhwc_apply": (_I, [_P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _F, _F, _I, _I, _I, _I, _I, _P]),
    "vm_batchnorm_nhwc_bwd_ex": (_I, [_P, _P, _L, _P, _P, _P, _P, _P, _I, _P, _L, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "vm_embedding_fwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vm_embedding_bwd_ex": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vm_gelu_bwd_bf16": (_I, [_P, _P, _P, _L, _P]),
    "vm_vit_assemble_ex": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vm_vit_assemble_bwd_ex": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vm_im2col_patches": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "vm_vit_assemble": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "vm_vit_assemble_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "vm_ce_shift_fwd_bwd": (_I, [_P, _L, _P, _I, _I, _I, _P, _P, _P, _F, _P, C.POINTER(C.c_int32), _I, _P, _P]),
    "vm_topk_threshold_bf16": (_I, [_P, _L, _I, _I, _I, C.POINTER(C.c_int32), _I, _P, _P]),
    "vm_transpose_f32": (_I, [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _P]),
    "vm_row_norm_f32": (_I, [_P, _L, _P, _I, _I, _P]),
    "vm_split3_bf16": (_I, [_P, _L, _I, _I, _P, _L, _I, _I, _I, _P]),
    "vm_gloria_attn_fwd": (_I, [_P, _L, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "vm_gloria_cos_fwd": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P]),
    "vm_gloria_cos_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P]),
    "vm_gloria_attn_bwd": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P]),
    "vm_ce_smooth_fwd_bwd": (_I, [_P, _P, _I, _I, _F, _P, _P, _F, _P]),
    "vm_rownorm_cast": (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "vm_contrastive_ws": (_SZ, [_I, _I]),
    "vm_contrastive_loss_fwd": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "vm_contrastive_loss_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "vm_cast_f32_to_bf16": (_I, [_P, _P, _L, _P]),
    "vm_cast_bf16_to_f32": (_I, [_P, _P, _L, _P]),
    "vm_cast_pad_f32_to_bf16": (_I, [_P, _P, _I, _I, _L, _P]),
    "vm_colsum_bf16": (_I, [_P, _L, _P, _I, _I, _P, _P]),
    "vm_add_bf16": (_I, [_P, _P, _P, _L, _P]),
    "vm_dropout_apply_bf16": (_I, [_P, _P, _L, _F, _U64, _P, _P]),
    "vm_feature_mask": (_I, [_P, _P, _I, _I, _P]),
    "vm_adam_step": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _F, _F, _P]),
    "vm_adam_step_dev": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _F, _F, _P, _P, _P, _P]),
    "vm_adam_step_wire": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _F, _F, _P, _P, _P, _P]),
    "vm_logsoftmax_f32": (_I, [_P, _L, _P, _I, _I, _P]),
    "vm_argmax_f32": (_I, [_P, _L, _P, _P, _I, _I, _P]),
    "vm_decode_gemm": (_I, [C.POINTER(DecodeGemmArgs), _P]),
    "vm_beam_topk": (_I, [_P, _L, _I, _I, _I, _P, _I, _P, _P, _P, _SZ, _P]),
    "vm_beam_topk_ws": (_SZ, [_I, _I, _I]),
    "vm_select_tokens": (_I, [_P, _L, _I, _I, _I, C.POINTER(C.c_int32), _I, _I, _U64, _P, _P, _L, _I, _P, _I, _I, _P]),
    "vm_gemm_f32": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _P, _I, _P, _L, _P]),
    "vm_layernorm_f32": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "vm_embedding_fwd_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "vm_attention_decode_f32": (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _F, _P]),
}


class VmHipError(RuntimeError):
    pass


def lib():
    """Load libvmhip.so (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VmHipError(f"{LIB_PATH} not found: build it with `python -m vilmedic_amd.build` "
                         "(the HIP hot path has no fallback)")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.vm_sizeof_gemm_epilogue() != C.sizeof(GemmEpilogue):
        raise VmHipError("vm_gemm_epilogue layout mismatch between include/vmhip.h and vilmedic_amd/_lib.py")
    _lib = L
    return L


def check(status, what=""):
    if status != 0:
        raise VmHipError(f"{what} failed ({status}): {lib().vm_last_error().decode()}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """raw hipStream_t of torch's current stream on the current device (the C accessor: no Stream object per launch)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())          # (a plain int: ctypes converts it for a c_void_p parameter itself)
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise VmHipError("vilmedic_amd ops need device tensors: the HIP path has no CPU fallback")
    return t.data_ptr() or None          # (a plain int -- ctypes converts it for a c_void_p parameter or field; address 0 -> NULL)
