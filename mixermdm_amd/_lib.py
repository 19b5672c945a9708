"""ctypes binding of libmmdm_hip.so (C ABI: include/mmdm.h).  Fails loudly; never falls back to CPU code."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MMDMError(RuntimeError):
    pass


def lib_path():
    """The library this process binds.  MMDM_LIB (DECLARED here and in INTEGRATION.md, "Diagnostics and environment"): an alternative build of
    the same ABI for A/B experiments of tools/ (e.g. one translation unit compiled with another macro; tools/ab_lib.sh).  bench.py stamps the path
    it ran and `"lib_override": true` into every line when the variable is set, so that a number measured on another build cannot pass for the
    in-tree library's."""
    return os.environ.get("MMDM_LIB") or os.path.join(_HERE, "libmmdm_hip.so")


def lib_override():
    return bool(os.environ.get("MMDM_LIB"))


class Config(C.Structure):
    """mmdm_config (include/mmdm.h)."""
    _fields_ = [(n, C.c_int) for n in ("d_latent", "d_ff", "d_layers", "d_heads", "m_latent", "m_ff", "m_layers", "m_heads",
                                        "nfeats", "text_dim", "mixing_mode", "align", "xstart_align", "model2_kind", "use_force")] + \
               [("force_val", C.c_float), ("cfg_scale", C.c_float)] + \
               [(n, C.c_int) for n in ("max_batch", "max_frames", "single_only")] + \
               [("cfg_scale_interaction", C.c_float), ("cfg_scale_individual", C.c_float), ("precision", C.c_int)] + \
               [(n, C.c_int) for n in ("model1_kind", "d1_latent", "d1_ff", "d1_layers", "d1_heads")]


class EncoderLayerWeights(C.Structure):
    """mmdm_encoder_layer_weights (include/mmdm.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("in_proj_weight", "in_proj_bias", "out_proj_weight", "out_proj_bias", "linear1_weight", "linear1_bias",
                                           "linear2_weight", "linear2_bias", "norm1_weight", "norm1_bias", "norm2_weight", "norm2_bias")]


# every symbol include/mmdm.h declares: name -> (restype, argtypes)
_I, _VP = C.c_int, C.c_void_p
SYMBOLS = {
    "mmdm_last_error": (C.c_char_p, []),
    "mmdm_version": (C.c_char_p, []),
    "mmdm_linear_f32": (_I, [_VP, _I, _VP, _I, _VP, _VP, _I, _I, _I, _I, _I, _VP, _I, _I, _VP]),
    "mmdm_linear_bf16": (_I, [_VP, _I, _VP, _I, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _I, _I, _VP]),
    "mmdm_f32_to_bf16": (_I, [_VP, _VP, C.c_int64, _VP]),
    "mmdm_linear_split": (_I, [_VP, _I, C.c_int64, _VP, _I, C.c_int64, _VP, _VP, _I, C.c_int64, _I, _I, _I, _I, _I, _VP, _I, _I, _VP]),
    "mmdm_linear_split_packed": (_I, [_VP, _I, C.c_int64, _VP, C.c_int64, _VP, _VP, _I, C.c_int64, _I, _I, _I, _I, _I, _VP, _I, _I, _VP]),
    "mmdm_split_pack_weight": (_I, [_VP, _I, C.c_int64, _VP, C.c_int64, _I, _I, _VP]),
    "mmdm_f32_split": (_I, [_VP, _VP, C.c_int64, C.c_int64, _VP]),
    "mmdm_adaln_f32": (_I, [_VP, _VP, _I, _I, _VP, _I, _I, _I, _VP]),
    "mmdm_adaln_ex": (_I, [_VP, _VP, _I, _I, _VP, _I, _I, _I, _I, _VP]),
    "mmdm_attention_ex": (_I, [_VP, _I, _VP, _I, _VP, _I, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "mmdm_attention_f32": (_I, [_VP, _I, _VP, _I, _VP, _I, _VP, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "mmdm_cond_silu_f32": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP]),
    "mmdm_mixer_pre_f32": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "mmdm_influence_head_f32": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "mmdm_mean_time_f32": (_I, [_VP, _VP, _I, _I, _I, _VP]),
    "mmdm_blend_cfg_f32": (_I, [_VP, _VP, _VP, _I, _I, C.c_float, C.c_float, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "mmdm_xstart_ddim_f32": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "mmdm_cfg_ddim_f32": (_I, [_VP, _VP, _I, _VP, C.c_float, _VP, _VP, _I, _I, _I, _VP]),
    "mmdm_gaussian_filter1d_f32": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "mmdm_cfg4_ddim_f32": (_I, [_VP, _VP, _I, _VP, C.c_float, C.c_float, C.c_float, _VP, _VP, _I, _I, _I, _VP]),
    "mmdm_attention_opts": (_I, [_VP, _I, _VP, _I, _VP, _I, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "mmdm_attention_planes": (_I, [_VP, _I, C.c_int64, _VP, _I, C.c_int64, _I, _VP, _I, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "mmdm_attention_bf16": (_I, [_VP, _I, _VP, _I, _VP, _I, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "mmdm_attention_split": (_I, [_VP, _I, C.c_int64, _VP, _I, C.c_int64, _VP, _I, C.c_int64, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP]),
    "mmdm_dual_ddim_f32": (_I, [_VP, _VP, _VP, _I, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _I, _I, _I, _VP]),
    "mmdm_layernorm_f32": (_I, [_VP, _VP, _VP, _VP, _I, _I, C.c_float, _VP]),
    "mmdm_token_embed_f32": (_I, [_VP, _I, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "mmdm_gather_rows_f32": (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    "mmdm_encoder_layer_workspace": (C.c_size_t, [_I, _I, _I, _I]),
    "mmdm_encoder_layer_f32": (_I, [_VP, C.POINTER(EncoderLayerWeights), _I, _I, _I, _I, _I, _I, _I, _I, C.c_float, _VP, C.c_size_t, _VP]),
    "mmdm_linear_fp8": (_I, [_VP, _I, _VP, _VP, _I, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _I, _I, _VP]),
    "mmdm_pack_weight_frag": (_I, [_VP, C.c_int64, _VP, _I, _I, _VP]),
    "mmdm_linear_bf16_packed": (_I, [_VP, _I, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _I, _I, _VP]),
    "mmdm_linear_fp8_packed": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _I, _I, _VP]),
    "mmdm_quantize_rows_fp8": (_I, [_VP, _I, _VP, _I, _VP, _I, _I, _VP]),
    "mmdm_adaln_fp8": (_I, [_VP, _VP, _I, _I, _VP, _VP, _I, _I, _I, _VP]),
    "mmdm_set_dual_weights": (_I, [_VP, _VP, _I]),
    "mmdm_create": (_I, [C.POINTER(Config), C.POINTER(_VP)]),
    "mmdm_create_shared": (_I, [_VP, _I, _I, C.POINTER(_VP)]),
    "mmdm_destroy": (None, [_VP]),
    "mmdm_handle_error": (C.c_char_p, [_VP]),
    "mmdm_set_weight": (_I, [_VP, C.c_char_p, _VP, C.c_int64, C.c_int64, _VP]),
    "mmdm_set_norm_stats": (_I, [_VP, _VP]),
    "mmdm_set_schedule": (_I, [_VP, _VP, _VP, _I, _VP]),
    "mmdm_prepare": (_I, [_VP]),
    "mmdm_begin": (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    "mmdm_begin_ragged": (_I, [_VP, _VP, _VP, _I, _VP, _VP]),
    "mmdm_call_rows": (_I, [_VP, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "mmdm_attention_ragged_f32": (_I, [_VP, _I, _VP, _I, _VP, _I, _VP, _I, _I, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "mmdm_set_history": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I]),
    "mmdm_run": (_I, [_VP, _I, _I, _VP]),
    "mmdm_seek": (_I, [_VP, _I, _VP]),
    "mmdm_graph_stats": (_I, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(_I)]),
    "mmdm_graph_parked": (_I, []),
    "mmdm_last_gemm_kernel": (C.c_char_p, []),
    "mmdm_get_state": (_I, [_VP] + [C.POINTER(_VP)] * 5),
    "mmdm_copy_result": (_I, [_VP, _VP, _VP]),
    "mmdm_module_forward": (_I, [_VP, _I, _VP, _VP, _VP, _I, _VP, _I, _I, _VP]),
    "mmdm_profile_enable": (_I, [_VP, _I]),
    "mmdm_diag_set": (_I, [C.c_char_p, C.c_longlong]),
    "mmdm_profile_read": (_I, [_VP, _I, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}


def load_library():
    """dlopen the in-tree library and bind every declared symbol (no GPU needed for this)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm ships its own libamdhip64.so (soname libamdhip64.so.7) and finds it by file name through an RPATH.
    # Load torch FIRST so that this library's NEEDED libamdhip64.so.7 binds to the runtime torch uses (one HIP runtime
    # per process: device pointers and streams are shared with torch).  Loaded the other way round, two runtimes
    # coexist and every HIP call from here reports "no ROCm-capable device".
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise MMDMError(f"{path} not found: build it with `python -m mixermdm_amd.build` (hipcc, gfx950). "
                        "There is no CPU fallback for the denoising path.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def diag(key, value):
    """mmdm_diag_set (include/mmdm.h section 4): process-global diagnostic switches of tools/ and bench.py's clock measurement."""
    check(load_library().mmdm_diag_set(key.encode(), int(value)))


def check(rc, handle=None):
    if rc == 0:
        return
    lib = load_library()
    msg = (lib.mmdm_handle_error(handle) if handle else None) or lib.mmdm_last_error() or b""
    msg = msg.decode(errors="replace")
    if "Mixing mode not recognized" in msg or "Mode not recognized" in msg:
        raise ValueError(msg)        # the reference raises ValueError here (mixermdm.py:786, influence.py:90)
    raise MMDMError(f"[mmdm status {rc}] {msg}")
