"""ctypes binding of libfishmi.so (the C ABI in include/fishmi.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the product path
raises.  (The CPU oracle lives in /oracle and is test infrastructure only.)"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfishmi.so")


class FishmiError(RuntimeError):
    pass


class DualARConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "vocab_size", "n_layer", "n_head", "n_local_heads", "head_dim", "dim", "intermediate_size",
        "n_fast_layer", "fast_dim", "fast_n_head", "fast_n_local_heads", "fast_head_dim",
        "fast_intermediate_size", "codebook_size", "num_codebooks", "semantic_begin_id",
        "semantic_end_id", "im_end_id", "max_seq_len", "attention_qk_norm", "fast_attention_qk_norm",
        "scale_codebook_embeddings", "norm_fastlayer_input")] + [("rope_base", C.c_float),
                                                                 ("norm_eps", C.c_float),
                                                                 ("weight_int8", C.c_int32)]


class SamplingC(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32),
                ("seed", C.c_uint32), ("use_ras", C.c_int32)]


class DacConfigC(C.Structure):
    _fields_ = [("encoder_dim", C.c_int32), ("encoder_rates", C.c_int32 * 4),
                ("decoder_dim", C.c_int32), ("decoder_rates", C.c_int32 * 4),
                ("latent_dim", C.c_int32), ("n_codebooks", C.c_int32), ("codebook_size", C.c_int32),
                ("semantic_codebook_size", C.c_int32), ("codebook_dim", C.c_int32),
                ("downsample", C.c_int32 * 2), ("tf_layers", C.c_int32), ("tf_heads", C.c_int32),
                ("tf_ffn", C.c_int32), ("tf_window", C.c_int32), ("enc_tf_layers", C.c_int32),
                ("enc_tf_window", C.c_int32), ("sample_rate", C.c_int32)]


_P = C.c_void_p
_I = C.c_int
_SIGS = {
    "fmi_version": (C.c_int, []),
    "fmi_last_error": (C.c_char_p, []),
    "fmi_device_arch": (C.c_int, [C.c_char_p, C.c_size_t]),
    "fmi_dualar_arena_bytes": (C.c_int64, [C.POINTER(DualARConfigC)]),
    "fmi_dualar_create": (C.c_int, [C.POINTER(DualARConfigC), _P, C.c_int64, C.POINTER(_P)]),
    "fmi_dualar_destroy": (None, [_P]),
    "fmi_dualar_load_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, C.c_int64, _I, _P]),
    "fmi_dualar_load_tensor_int8": (C.c_int, [_P, C.c_char_p, _P, _P, C.c_int64, C.c_int64, _I, _P]),
    "fmi_dualar_finalize_weights": (C.c_int, [_P, _P]),
    "fmi_dualar_weights_ready": (C.c_int, [_P, _P]),
    "fmi_dualar_setup_caches": (C.c_int, [_P, _I, _I]),
    "fmi_dualar_prefill": (C.c_int, [_P, _I, C.POINTER(C.c_int32), _P, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.POINTER(SamplingC), _P]),
    "fmi_dualar_prefill_resume": (C.c_int, [_P, _I, C.POINTER(C.c_int32), _P, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.POINTER(SamplingC), _P]),
    "fmi_dualar_decode": (C.c_int, [_P, _I, C.POINTER(C.c_int32), _I, _P]),
    "fmi_dualar_read": (C.c_int, [_P, _I, C.POINTER(C.c_int32), _I, C.POINTER(_I), C.POINTER(_I), _P]),
    "fmi_dualar_poll_done": (C.c_int, [_P, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    "fmi_dualar_release": (C.c_int, [_P, _I]),
    "fmi_dualar_out_ptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I)]),
    "fmi_dualar_wait": (C.c_int, [_P, _P]),
    "fmi_dualar_synchronize": (C.c_int, [_P]),
    "fmi_dualar_step": (C.c_int, [_P, _I, _P, _I, _I, C.POINTER(SamplingC), _P, C.c_int32, _P, _P]),
    "fmi_dualar_forward_slow": (C.c_int, [_P, _I, _P, _I, _I, _P, _P, _P]),
    "fmi_dualar_forward_fast": (C.c_int, [_P, _I, _P, _I, _P, _P]),
    "fmi_dualar_table_ptr": (C.c_int, [_P, _I, C.POINTER(_P), C.POINTER(_I), C.POINTER(_I)]),
    "fmi_dualar_derived_info": (C.c_int, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "fmi_dualar_debug_ptrs": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I), C.POINTER(_I), C.POINTER(_P),
                                        C.POINTER(_P), C.POINTER(_P)]),
    "fmi_dualar_set_trace": (C.c_int, [_P, _I, C.POINTER(_P)]),
    "fmi_dualar_fast_chain_forced": (C.c_int, [_P, _I, C.POINTER(C.c_int32), _P, _P, _I, _P, _P]),
    "fmi_dualar_set_graph": (C.c_int, [_P, _I]),
    "fmi_dualar_set_attn_impl": (C.c_int, [_P, _I]),
    "fmi_dualar_set_fast_merge": (C.c_int, [_P, _I]),
    "fmi_dualar_set_stream_priority": (C.c_int, [_P, _I]),
    "fmi_dualar_set_attn_long_threshold": (C.c_int, [_P, _I]),
    "fmi_dualar_set_ignore_eos": (C.c_int, [_P, _I]),
    "fmi_dualar_last_decode_stats": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(_I)]),
    "fmi_op_linear_int8": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, C.c_float, _I, _I, _P]),
    "fmi_op_linear_bf16": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, C.c_float, _I, _I, _P]),
    "fmi_op_sample": (C.c_int, [_P, _I, _I, _I, _P, C.POINTER(SamplingC), _I, _I, _P, _I, _I, _P, _P]),
    "fmi_dac_arena_bytes": (C.c_int64, [C.POINTER(DacConfigC)]),
    "fmi_dac_create": (C.c_int, [C.POINTER(DacConfigC), _P, C.c_int64, C.POINTER(_P)]),
    "fmi_dac_destroy": (None, [_P]),
    "fmi_dac_load_tensor": (C.c_int, [_P, C.c_char_p, _P, _I, C.POINTER(C.c_int64), _I, _P]),
    "fmi_dac_finalize_weights": (C.c_int, [_P, _P]),
    "fmi_dac_weights_ready": (C.c_int, [_P, _P]),
    "fmi_dac_set_precision": (C.c_int, [_P, _I]),
    "fmi_dac_set_async": (C.c_int, [_P, _I]),
    "fmi_dac_set_background": (C.c_int, [_P, _I]),
    "fmi_dac_wait": (C.c_int, [_P, _P]),
    "fmi_dac_synchronize": (C.c_int, [_P]),
    "fmi_dac_set_stream_options": (C.c_int, [_P, _I, _I, C.POINTER(C.c_uint32)]),
    "fmi_dac_fp16_overflow": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "fmi_dac_decode": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "fmi_dac_decode_latent": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "fmi_dac_decode_tail": (C.c_int, [_P, _P, _I, _I, _I, _P, _P]),
    "fmi_dac_decode_tail_cached": (C.c_int, [_P, _P, _I, _I, _I, C.c_int64, _P, _P]),
    "fmi_dac_stream_reset": (C.c_int, [_P]),
    "fmi_dac_stream_close": (C.c_int, [_P, C.c_int64]),
    "fmi_dac_context_frames": (C.c_int, [_P]),
    "fmi_dac_encode": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "fmi_dac_frame_length": (C.c_int, [_P]),
    "fmi_dac_debug_z": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I), C.POINTER(_I)]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)
_lib = None


def load() -> C.CDLL:
    """dlopen libfishmi.so and type its entry points.  Raises FishmiError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FishmiError(
            f"{LIB_PATH} not found: build it with `python -m fish_speech_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch ships its own libamdhip64; the process must hold ONE HIP runtime, and the tensors / streams handed to the
    # library are torch's, so torch's copy has to be the one that is resident when libfishmi.so resolves its
    # dependency (loading the library first pulled in /opt/rocm's runtime and every hipStreamCreate then failed)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().fmi_last_error()
        raise FishmiError(f"libfishmi error {rc}: {msg.decode(errors='replace') if msg else ''}")


def stream_ptr(device=None) -> C.c_void_p:
    import torch

    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
