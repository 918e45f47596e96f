"""ctypes binding of libfgdm_hip.so (the C ABI declared in include/fgdm.h).

There is deliberately NO fallback: if the shared library is missing or a symbol is
absent, importing the product path fails loudly."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libfgdm_hip.so')

MAX_LEVELS = 8
MAX_CONTROLNETS = 4

FLAG_USE_ORIGINAL = 1
FLAG_ONLY_MID_CONTROL = 2
FLAG_NO_CONTROL = 4
FLAG_CFG_PAIRS = 8

ACT_NONE, ACT_SILU, ACT_RELU, ACT_GEGLU, ACT_QGELU = 0, 1, 2, 3, 4
OUT_F16, OUT_F32, OUT_F32_NCHW, OUT_F16_T = 0, 1, 2, 3


class FgdmConfig(C.Structure):
    _fields_ = [
        ('in_channels', C.c_int32), ('out_channels', C.c_int32), ('model_channels', C.c_int32),
        ('num_res_blocks', C.c_int32), ('n_levels', C.c_int32), ('channel_mult', C.c_int32 * MAX_LEVELS),
        ('n_attention_resolutions', C.c_int32), ('attention_resolutions', C.c_int32 * MAX_LEVELS),
        ('num_heads', C.c_int32), ('context_dim', C.c_int32), ('use_adapter', C.c_int32),
        ('n_controlnets', C.c_int32), ('hint_channels', C.c_int32), ('workspace_bytes', C.c_int64),
        ('vae_ch', C.c_int32), ('vae_n_levels', C.c_int32), ('vae_ch_mult', C.c_int32 * MAX_LEVELS),
        ('vae_num_res_blocks', C.c_int32), ('vae_z_channels', C.c_int32), ('vae_out_ch', C.c_int32),
        ('clip_layers', C.c_int32), ('clip_width', C.c_int32), ('clip_heads', C.c_int32), ('clip_mlp', C.c_int32),
        ('clip_vocab', C.c_int32), ('clip_max_len', C.c_int32),
        ('n_extra_adapters', C.c_int32),
    ]


_p = C.c_void_p
_f = C.c_float
_i = C.c_int
_i64 = C.c_int64

# name -> (restype, argtypes); every symbol include/fgdm.h declares
SIGNATURES = {
    'fgdm_create': (_i, [C.POINTER(FgdmConfig), _i, C.POINTER(_p)]),
    'fgdm_destroy': (None, [_p]),
    'fgdm_last_error': (C.c_char_p, [_p]),
    'fgdm_param_count': (_i, [C.POINTER(FgdmConfig)]),
    'fgdm_param_info': (_i, [C.POINTER(FgdmConfig), _i, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(_i)]),
    'fgdm_load_tensor': (_i, [_p, C.c_char_p, _p, _i, C.POINTER(_i64), _i]),
    'fgdm_finalize_weights': (_i, [_p]),
    'fgdm_set_hint': (_i, [_p, _i, _p, _i, _i, _i, _p]),
    'fgdm_set_adapter_conds': (_i, [_p, _p, _i, _i, _i, _i, _p]),
    'fgdm_set_context': (_i, [_p, _p, _i, _p]),
    'fgdm_apply_model': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    'fgdm_clip_encode': (_i, [_p, _p, _i, _i, _p, _p]),
    'fgdm_run_block': (_i, [_p, C.c_char_p, _p, _i, _p, _i, _p, _p, _i, _i, _i, _p, _i64, C.POINTER(_i64), _p]),
    'fgdm_vae_decode': (_i, [_p, _p, _i, _i, _i, _f, _p, _p]),
    'fgdm_image_to_uint8': (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    'fgdm_resize_linear_uint8': (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    'fgdm_uint8_to_hint': (_i, [_p, _i, _i, _i, _i, _p, _p]),
    'fgdm_controlnet': (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _p, _i64, _p]),
    'fgdm_ddim_step': (_i, [_p, _p, _p, _f, _f, _f, _f, _f, _p, _p, _p, _p, _i64, _p]),
    'fgdm_plms_combine': (_i, [_p, _p, _p, _p, _i, _p, _i64, _p]),
    'fgdm_axpby': (_i, [_p, _f, _p, _f, _p, _i64, _p]),
    'fgdm_mask_blend': (_i, [_p, _p, _p, _p, _i64, _p]),
    'fgdm_ancestral_step': (_i, [_p, _p, _f, _f, _f, _f, _f, _p, _p, _i64, _p]),
    'fgdm_sample_ddim': (_i, [_p, _p, _p, _p, _f, _i, C.POINTER(_i64), C.POINTER(_f), C.POINTER(_f), C.POINTER(_f),
                              _p, _i, _i, _i, _i, _p]),
    'fgdm_profile_begin': (_i, [_p, _i]),
    'fgdm_profile_end': (_i, [_p, C.POINTER(C.c_double)]),
    'fgdm_workspace_stats': (_i, [_p, C.POINTER(_i64), C.POINTER(_i64)]),
    'fgdm_launch_stats': (_i, [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    'fgdm_op_conv2d': (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p]),
    'fgdm_op_linear': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    'fgdm_debug_force_igemm_cfg': (_i, [_i]),
    'fgdm_bench_igemm': (_i, [_i] * 13 + [C.POINTER(_f)]),
    'fgdm_op_linear_ln_linear': (_i, [_p] * 8 + [_i] * 5 + [_p, _p, C.POINTER(_i), _p]),
    'fgdm_bench_norm': (_i, [_i] * 7 + [C.POINTER(_f)]),
    'fgdm_bench_attention': (_i, [_i] * 6 + [C.POINTER(_f)]),
    'fgdm_bench_ff': (_i, [_i] * 4 + [C.POINTER(_f)]),
    'fgdm_op_groupnorm': (_i, [_p, _i, _p, _i, _i, _i, _p, _p, _f, _i, _p, _p]),
    'fgdm_op_layernorm': (_i, [_p, _i, _i, _p, _p, _f, _p, _p]),
    'fgdm_op_attention': (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
}

_lib = None


def load():
    """Load the shared library and bind every symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # The library binds to whichever libamdhip64 the process already holds.  PyTorch-ROCm ships its own copy of the HIP
    # runtime: if this library were loaded first it would pull in the system copy, the process would then hold two
    # runtimes, and the second one sees "no ROCm-capable device".  Let torch (device memory / streams plumbing of the
    # Python host layer) load its runtime first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # FGDM_LIB: another BUILD of the same library (tools/ab_lib.sh alternates two builds on one box without touching the in-tree
    # file); it must export every symbol like the in-tree one, and there is still no fallback
    path = os.environ.get('FGDM_LIB') or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f'{path} not found: build it with `python -m fgdm_amd.build` '
                           '(the HIP engine is mandatory, there is no CPU fallback)')
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
