"""ctypes binding of libecgbyte_hip.so (the C ABI declared in include/ecgbyte.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing,
or a device entry point is called without a GPU, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_PKG, "libecgbyte_hip.so")
CSRC = os.path.join(_PKG, "csrc")

ECGB_OK = 0
ERRORS = {-1: "INVALID", -2: "NOMEM", -3: "UNSUPPORTED", -4: "HIP", -5: "NODEVICE"}


class EcgbError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libecgbyte_hip: {ERRORS.get(code, code)}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-j", str(min(8, os.cpu_count() or 1)), "-C", CSRC]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C ecg_byte_amd/csrc`). "
            "There is no CPU fallback for the encode path.")
    L = C.CDLL(SO_PATH)
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    u32p = C.POINTER(C.c_uint32)
    L.ecgb_last_error.restype = C.c_char_p
    L.ecgb_version.restype = u32
    L.ecgb_tokenizer_create.argtypes = [u32p, u32p, u32p, sz, C.POINTER(vp)]
    L.ecgb_tokenizer_create.restype = C.c_int
    L.ecgb_tokenizer_destroy.argtypes = [vp]
    L.ecgb_tokenizer_destroy.restype = None
    L.ecgb_tokenizer_info.argtypes = [vp, u32p, u32p, u32p]
    L.ecgb_tokenizer_info.restype = C.c_int
    L.ecgb_tokenizer_copy_nodes.argtypes = [vp, C.POINTER(C.c_uint64), sz]
    L.ecgb_tokenizer_copy_nodes.restype = sz
    L.ecgb_tokenizer_copy_runbits.argtypes = [vp, u32p, sz]
    L.ecgb_tokenizer_copy_runbits.restype = sz
    L.ecgb_quantize_hip.argtypes = [vp, sz, C.c_double, C.c_double, vp, vp, vp]
    L.ecgb_quantize_hip.restype = C.c_int
    L.ecgb_encode_scratch_bytes.argtypes = [vp, sz, sz]
    L.ecgb_encode_scratch_bytes.restype = sz
    L.ecgb_encode_hip.argtypes = [vp, vp, sz, sz, vp, sz, vp, vp, sz, vp]
    L.ecgb_encode_hip.restype = C.c_int
    L.ecgb_quantize_encode_hip.argtypes = [vp, vp, sz, sz, C.c_double, C.c_double, vp, sz, vp, vp, sz, vp]
    L.ecgb_quantize_encode_hip.restype = C.c_int
    L.ecgb_quantizer_thresholds.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
    L.ecgb_quantizer_thresholds.restype = C.c_int
    L.ecgb_set_encode_plan.argtypes = [C.c_int]
    L.ecgb_set_encode_plan.restype = C.c_int
    L.ecgb_set_bpe_train_grid.argtypes = [C.c_int]
    L.ecgb_set_bpe_train_grid.restype = C.c_int
    L.ecgb_set_bpe_train_form.argtypes = [C.c_int]
    L.ecgb_set_bpe_train_form.restype = C.c_int
    L.ecgb_set_bpe_train_fused.argtypes = [C.c_int]
    L.ecgb_set_bpe_train_fused.restype = C.c_int
    L.ecgb_bpe_train_scratch_bytes.argtypes = [sz, u32]
    L.ecgb_bpe_train_scratch_bytes.restype = sz
    L.ecgb_bpe_train_hip.argtypes = [vp, sz, u32, vp, vp, vp, vp, vp, sz, vp]
    L.ecgb_bpe_train_hip.restype = C.c_int
    L.ecgb_bpe_shard_create.argtypes = [sz, u32, vp, sz]
    L.ecgb_bpe_shard_create.restype = vp
    L.ecgb_bpe_shard_destroy.argtypes = [vp]
    L.ecgb_bpe_shard_destroy.restype = None
    for name in ("ecgb_bpe_shard_table", "ecgb_bpe_shard_slab"):
        getattr(L, name).argtypes = [vp, C.POINTER(sz)]
        getattr(L, name).restype = vp
    for name, args in {"ecgb_bpe_shard_begin": [vp, vp, vp, vp], "ecgb_bpe_shard_count": [vp, vp, C.c_int, C.c_int, vp],
                       "ecgb_bpe_shard_pick": [vp, u32, vp, vp], "ecgb_bpe_shard_merge": [vp, u32, vp, C.c_int, C.c_int, vp],
                       "ecgb_bpe_shard_apply": [vp, vp], "ecgb_bpe_shard_finish": [vp, vp, vp, vp, vp, vp]}.items():
        getattr(L, name).argtypes = args
        getattr(L, name).restype = C.c_int
    L.ecgb_filtfilt_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.ecgb_filtfilt_scratch_bytes.restype = sz
    L.ecgb_filtfilt_f64.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), vp, vp, vp, vp, sz, vp]
    L.ecgb_filtfilt_f64.restype = C.c_int
    L.ecgb_filtfilt_planar_f64.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), vp, vp, vp, vp, sz, vp, vp, vp]
    L.ecgb_filtfilt_planar_f64.restype = C.c_int
    L.ecgb_wavelet_denoise_planar_f64.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, vp]
    L.ecgb_wavelet_denoise_planar_f64.restype = C.c_int
    L.ecgb_resample_cubic_planar_f64.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), vp, sz, vp, vp]
    L.ecgb_resample_cubic_planar_f64.restype = C.c_int
    L.ecgb_set_wavelet_workgroup_kernel.argtypes = [C.c_int]
    L.ecgb_set_wavelet_workgroup_kernel.restype = None
    L.ecgb_resample_cubic_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    L.ecgb_resample_cubic_scratch_bytes.restype = sz
    L.ecgb_resample_cubic_f64.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, sz, vp]
    L.ecgb_resample_cubic_f64.restype = C.c_int
    L.ecgb_wavelet_denoise_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
    L.ecgb_wavelet_denoise_scratch_bytes.restype = sz
    L.ecgb_wavelet_denoise_f64.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, vp, sz, vp]
    L.ecgb_wavelet_denoise_f64.restype = C.c_int
    L.ecgb_nonfinite_records_f64.argtypes = [vp, C.c_int, sz, vp, vp]
    L.ecgb_nonfinite_records_f64.restype = C.c_int
    i32 = C.c_int32
    L.ecgb_assemble_hip.argtypes = [vp, sz, vp, sz, vp, sz, vp, vp, vp, vp, i32, i32, i32, i32, i32, u32,
                                    C.c_int, u32, vp, vp, vp, vp, vp, vp]
    L.ecgb_assemble_hip.restype = C.c_int
    # ---- decoder ops (include/ecgbyte_decoder.h)
    f32, ll, ci = C.c_float, C.c_longlong, C.c_int
    sigs = {
        "ecgb_embed_fwd": [vp, vp, vp, sz, ci, f32, vp],
        "ecgb_embed_bwd": [vp, vp, vp, sz, ci, f32, vp],
        "ecgb_set_attn_fwd_staging": [ci],
        "ecgb_set_attn_d256_pass_p": [ci],
        "ecgb_set_ce_in_registers": [ci],
        "ecgb_set_rmsnorm_fwd_rows": [ci],
        "ecgb_set_rmsnorm_bwd_rows_per_wg": [ci],
        "ecgb_set_attn_lean_waves": [ci],
        "ecgb_transpose_multi_bf16": [vp, vp, vp, vp, vp, ci, ci, vp],
        "ecgb_embed_bwd_sorted": [vp, vp, vp, vp, sz, ci, f32, C.c_int64, vp],
        "ecgb_rmsnorm_fwd": [vp, vp, vp, vp, vp, vp, sz, ci, f32, ci, vp],
        "ecgb_rmsnorm_lora_fwd": [vp, vp, vp, vp, vp, vp, sz, ci, f32, ci, vp, ll, ci, f32, vp, ll, vp],
        "ecgb_rmsnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, sz, ci, ci, vp, vp],
        "ecgb_rope": [vp, vp, vp, sz, ci, ci, sz, ci, vp],
        "ecgb_rope_append": [vp, vp, vp, ci, ci, ci, ci, sz, vp, ll, ci, vp, vp],
        "ecgb_glu_fwd": [vp, vp, sz, ci, ci, vp],
        "ecgb_glu_bwd": [vp, vp, vp, sz, ci, ci, vp],
        "ecgb_add_bf16": [vp, vp, vp, sz, vp],
        "ecgb_dropout_bf16": [vp, vp, sz, f32, C.c_uint64, vp],
        "ecgb_transpose_bf16": [vp, vp, ci, ci, vp],
        "ecgb_f32_to_bf16": [vp, vp, sz, vp],
        "ecgb_gemm_nt_bf16": [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, ci, ci, ll, ll, ll, vp],
        "ecgb_gemm_nt_bf16_heads": [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, ci, ci, ci, ll, ll, ci, ll, ll, ci, ll, ll, vp],
        "ecgb_transpose_bf16_strided": [vp, vp, ci, ci, ll, ll, ci, ci, ll, ll, ll, ll, vp],
        "ecgb_count_labels": [vp, sz, ci, vp, vp],
        "ecgb_ce_fwd_bwd": [vp, vp, vp, vp, vp, sz, ci, sz, vp],
        "ecgb_sumsq": [vp, sz, ci, vp, vp],
        "ecgb_sumsq_multi_bf16": [vp, vp, vp, vp, ci, vp, vp, vp],
        "ecgb_adam_step": [vp, vp, ci, vp, vp, sz, vp, f32, f32, f32, f32, f32, f32, ci, vp],
        "ecgb_adam_multi_bf16": [vp, vp, vp, vp, vp, vp, vp, ci, vp, f32, f32, f32, f32, f32, f32, ci, vp],
        "ecgb_softmax_causal_fwd": [vp, vp, ci, ci, ci, f32, vp],
        "ecgb_softmax_bwd": [vp, vp, ci, ci, f32, vp],
        "ecgb_set_gemm_tile": [ci],
        "ecgb_set_gemm_backward_persistent": [ci],
        "ecgb_get_gemm_backward_persistent": [],
        "ecgb_set_gemm_group_m": [ci],
        "ecgb_attn_decode": [vp, vp, vp, ll, ll, vp, ll, vp, ci, ci, ci, ci, ci, f32, vp],
        "ecgb_attn_decode_dyn": [vp, vp, vp, ll, ll, vp, ll, vp, ci, vp, ci, ci, ci, f32, vp],
        "ecgb_attn_decode_split": [vp, vp, vp, ll, ll, vp, ll, vp, ci, ci, ci, ci, ci, f32, ci, vp, sz, vp],
        "ecgb_kv_append": [vp, ll, ll, ci, vp, ll, ci, vp, vp],
        "ecgb_argmax_bf16": [vp, ll, ci, ci, vp, vp],
        "ecgb_decode_advance": [vp, ci, vp, vp, vp, vp, vp, ll, vp, ll, vp, ll, vp, ci, vp],
        "ecgb_decode_advance_e": [vp, ci, vp, vp, vp, vp, vp, ll, vp, ll, vp, ll, vp, ci, vp, vp],
        "ecgb_gemm_nt_bf16_lora_decode": [vp, ll, vp, ll, vp, ll, f32, vp, ll, ci, vp, vp, ll, ci, ci, ci, vp, vp],
        "ecgb_rope_table": [vp, ci, vp, ci, vp, vp, vp],
        "ecgb_gemm_nt_w4_bf16": [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, vp],
        "ecgb_gemm_nn_w4_bf16": [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, vp],
        "ecgb_gemm_tn_w4_bf16": [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, vp],
        "ecgb_gemm_nt_bf16_rope": [vp, ll, vp, ll, vp, ll, vp, ll, ci, vp, ll, ci, ci, ci, f32, vp, vp, ci, vp],
        "ecgb_set_gemm_w4": [ci],
        "ecgb_set_gemm_w4_group_m": [ci],
        "ecgb_gemm_w4_span_ok": [ci, ll, ll, ll],
        "ecgb_decode_norm_gemv": [vp, vp, vp, f32, ci, ci, ci, vp, vp, ll, ci, vp, ll, ci, f32, vp, ll, vp, ll, ci, vp],
        "ecgb_decode_gemv": [vp, ll, ci, ci, vp, ll, ci, vp, ll, ci, f32, vp, vp, ll, vp, ll, vp],
        "ecgb_decode_lora_t": [vp, ll, ci, ci, vp, ll, ci, f32, vp, vp],
        "ecgb_decode_attn": [vp, ll, vp, vp, vp, ll, vp, ll, vp, ci, ci, vp, ci, ci, ci, f32, ci, vp, sz, vp],
        "ecgb_attn_decode_one": [vp, ll, vp, vp, vp, ll, ll, vp, ll, vp, ci, ci, vp, ci, ci, ci, f32, ci, vp, sz, vp],
        "ecgb_set_gemm_w4_sched": [ci],
        "ecgb_set_stream_grid_cap": [ci],
        "ecgb_set_rmsnorm_bwd_grid_cap": [ci],
        "ecgb_set_gemm_w4_min_ktiles": [ci],
        "ecgb_attn_decode_split_dyn": [vp, vp, vp, ll, ll, vp, ll, vp, ci, vp, ci, ci, ci, f32, ci, vp, sz, vp],
        "ecgb_gemm_tn_bf16": [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, ci, vp],
        "ecgb_sum_slabs_bf16": [vp, ll, ci, vp, sz, ci, vp],
        "ecgb_gemm_nn_bf16": [vp, ll, vp, ll, vp, ll, ci, ci, ci, f32, ci, vp],
        "ecgb_gemm_nn_glu_bwd_bf16": [vp, ll, vp, ll, vp, ll, vp, ll, ci, ci, ci, ci, vp],
        "ecgb_gemm_nn_glu_bwd_lora_bf16": [vp, ll, vp, ll, vp, ll, vp, vp, vp, ll, ci, ci, ci, ci, f32, f32, C.c_uint64, vp],
        "ecgb_gemm_nn_lora_bf16": [vp, ll, vp, ll, vp, vp, vp, ll, ci, ci, ci, f32, f32, C.c_uint64, vp],
        "ecgb_gemm_nn_splitk_bf16": [vp, ll, vp, ll, vp, vp, ci, ci, ci, ci, f32, vp],
        "ecgb_gemm_nt_glu_bf16": [vp, ll, vp, ll, vp, ll, vp, ll, ci, vp, ll, vp, ll, ci, ci, ci, f32, ci, vp],
        "ecgb_gemm_nt_bf16_cat": [vp, ll, vp, ll, vp, ll, vp, ll, ci, vp, ll, ci, ci, ci, f32, ci, vp],
        "ecgb_layernorm_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, sz, ci, f32, vp],
        "ecgb_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, ci, vp, vp],
        "ecgb_bias_act": [vp, vp, vp, sz, ci, ci, vp],
        "ecgb_gelu_new_bwd": [vp, vp, vp, sz, vp],
        "ecgb_colsum": [vp, vp, sz, ci, vp, vp],
        "ecgb_lora_down": [vp, vp, vp, vp, ci, ci, ci, ci, f32, f32, C.c_uint64, vp],
        "ecgb_lora_dx": [vp, vp, vp, ci, ci, ci, ci, f32, f32, C.c_uint64, vp],
        "ecgb_lora_da": [vp, vp, vp, ci, ci, ci, ci, f32, f32, C.c_uint64, ci, vp, C.c_size_t, vp],
        "ecgb_lora_dx_glu": [vp, vp, vp, vp, vp, ci, ci, ci, ci, f32, f32, C.c_uint64, ci, vp],
        "ecgb_attn_fwd": [vp, ll, vp, ll, vp, ll, vp, vp, ll, vp, ci, ci, ci, ci, ci, f32, vp],
        "ecgb_attn_bwd": [vp, ll, vp, ll, vp, ll, vp, vp, vp, ll, vp, vp, vp, ll, vp, ll, vp, ll, ci, ci, ci, ci, ci, f32, vp, sz, vp],
        "ecgb_attn_bwd_rope": [vp, ll, vp, ll, vp, ll, vp, vp, vp, ll, vp, vp, vp, ll, vp, ll, vp, ll, vp, vp, ci, ci, ci, ci, ci, f32, vp, sz, vp],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name in ("ecgb_layernorm_bwd_scratch_floats", "ecgb_colsum_scratch_floats"):
        getattr(L, name).argtypes = [sz, ci]
        getattr(L, name).restype = sz
    L.ecgb_partial_rows_sum_f32.argtypes = [vp, ci, ci, ll, vp, vp]
    L.ecgb_partial_rows_sum_f32.restype = C.c_int
    L.ecgb_decode_attn_scratch_floats.argtypes = [ll, ci, ci, ci, ci, ci]
    L.ecgb_decode_attn_scratch_floats.restype = sz
    L.ecgb_attn_decode_one_scratch_floats.argtypes = [ci, ci, ci, ci]
    L.ecgb_attn_decode_one_scratch_floats.restype = sz
    L.ecgb_rmsnorm_bwd_scratch_floats.argtypes = [sz, ci]
    L.ecgb_rmsnorm_bwd_scratch_floats.restype = sz
    L.ecgb_lora_da_scratch_bytes.argtypes = [ci, ci, ci]
    L.ecgb_lora_da_scratch_bytes.restype = sz
    L.ecgb_attn_bwd_scratch_bytes.argtypes = [ci, ci, ci, ci, ci]
    L.ecgb_attn_bwd_scratch_bytes.restype = sz
    L.ecgb_attn_decode_split_scratch_bytes.argtypes = [ll, ci, ci, ci, ci]
    L.ecgb_attn_decode_split_scratch_bytes.restype = sz
    _lib = L
    return L


def require_current(device) -> None:
    """The C ABI takes raw pointers and launches on torch's current stream of the CURRENT device, so tensors on another
    device would be touched by kernels running on the wrong GPU.  Fail loudly instead (ecg_byte/main.py's `--device cuda:N`
    works because main() makes N current)."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if device.type != "cuda" or idx != torch.cuda.current_device():
        raise RuntimeError(f"tensor on {device} but the current device is cuda:{torch.cuda.current_device()}: call "
                           f"torch.cuda.set_device({device!s}) first (kernels launch on the current device's stream)")


def check(rc: int) -> None:
    if rc != ECGB_OK:
        raise EcgbError(rc, (lib().ecgb_last_error() or b"").decode("utf-8", "replace"))
