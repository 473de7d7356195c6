"""ctypes binding of libdinodet.so (include/dinodet.h).  PyTorch is used only for device
memory (tensor.data_ptr()) and the current HIP stream; no torch type crosses the ABI.

There is NO fallback: if the library is missing or a GPU op is requested without it,
loading raises.  Build with `python -m dinov2_od_amd._build`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DINODET_LIB") or os.path.join(_HERE, "lib", "libdinodet.so")      # DINODET_LIB: another build of the same ABI (A/B runs)

DOD_F32, DOD_BF16 = 0, 1
PREC = {"fp32": 0, "bf16": 1, "fp8": 2, "bf16x3": 3, "fp16x2": 4}
ACT = {"none": 0, "relu": 1, "gelu": 2, "sigmoid": 3, "swiglu_pairs": 4}


class DodConfig(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("swiglu", C.c_int32),
        ("patch", C.c_int32), ("pos_grid", C.c_int32), ("ffn_hidden", C.c_int32), ("ln_eps", C.c_float),
        ("lora_r", C.c_int32), ("lora_alpha", C.c_float), ("target_dim", C.c_int32),
        ("num_queries", C.c_int32), ("dec_hidden", C.c_int32), ("dec_heads", C.c_int32),
        ("dec_layers", C.c_int32), ("num_classes", C.c_int32), ("dim_feedforward", C.c_int32),
        ("n_points", C.c_int32), ("use_deformable", C.c_int32), ("dec_ln_eps", C.c_float),
        ("precision", C.c_int32),
    ]


_P, _I, _F, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class DodDetection(C.Structure):
    """struct dod_detection (include/dinodet.h): one COCO-style record of evaluate_coco, utils.py:225-233"""
    _fields_ = [("image_id", C.c_int64), ("category_id", C.c_int32), ("query", C.c_int32), ("bbox", C.c_float * 4),
                ("score", C.c_float), ("reserved", C.c_int32)]

# struct dod_dec_train_params: 31 device pointers in this order (the same struct carries the gradient accumulators)
DEC_TRAIN_FIELDS = ["query_embed", "class_w", "class_b", "bb0_w", "bb0_b", "bb2_w", "bb2_b", "in_proj_w", "in_proj_b", "out_proj_w",
                    "out_proj_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b", "norm3_w", "norm3_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b",
                    "refp_w", "refp_b", "off_w", "off_b", "aw_w", "aw_b", "vp_w", "vp_b", "op_w", "op_b"]


class DodDecTrainParams(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in DEC_TRAIN_FIELDS]


DENSE_LAYER_FIELDS = ["sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ca_in_w", "ca_in_b", "ca_out_w", "ca_out_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b",
                      "norm1_w", "norm1_b", "norm2_w", "norm2_b", "norm3_w", "norm3_b"]


class DodDenseLayerParams(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in DENSE_LAYER_FIELDS]


class DodDenseDecTrainParams(C.Structure):
    _fields_ = [("nlayers", C.c_int32), ("reserved", C.c_int32), ("layers", C.POINTER(DodDenseLayerParams))] + \
               [(f, C.c_void_p) for f in ("query_embed", "class_w", "class_b", "bb0_w", "bb0_b", "bb2_w", "bb2_b")]


class DodLoraLinear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("A", C.c_void_p), ("Bm", C.c_void_p)]


class DodBbBlockParams(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "ls1", "ls2")] + \
               [(f, DodLoraLinear) for f in ("q", "k", "v", "o", "fc1", "fc2")]


class DodBbTailParams(C.Structure):
    _fields_ = [("nblocks", C.c_int32), ("reserved", C.c_int32), ("blocks", C.POINTER(DodBbBlockParams)),
                ("lnf_w", C.c_void_p), ("lnf_b", C.c_void_p), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p)]


class DodLnFold(C.Structure):
    """struct dod_ln_fold (include/dinodet.h): the folded-LayerNorm legs of dod_op_linear_ln"""
    _fields_ = [("stats", C.c_void_p), ("csum", C.c_void_p), ("op_out", C.c_void_p), ("part", C.c_void_p), ("shift", C.c_void_p),
                ("part_in", C.c_void_p), ("stats_out", C.c_void_p), ("eps", C.c_float)]


# include/dinodet_tuning.h: exported by -DDINODET_TUNING builds only (tools/ load one through DINODET_LIB); bound when present
TUNING_SYMBOLS = {
    "dod_debug_gemm_stamps": (_I, [_P]),
    "dod_debug_pp_stamps": (_I, [_P]),
    "dod_debug_attn_stamps": (_I, [_P]),
    "dod_debug_mfma_peak": (_I, [_I, _I, _I, _P, _P]),
    "dod_debug_mfma_valu_probe": (_I, [_I, _I, _I, _I, _P, _P]),
}

# name -> (restype, argtypes): every symbol include/dinodet.h declares
SYMBOLS = {
    "dod_create": (_I, [C.POINTER(DodConfig), C.POINTER(_P)]),
    "dod_destroy": (None, [_P]),
    "dod_last_error": (C.c_char_p, [_P]),
    "dod_set_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I, _I]),
    "dod_finalize_weights": (_I, [_P, _P]),
    "dod_prepare": (_I, [_P, _I, _I, _P]),
    "dod_workspace_bytes": (_SZ, [_P, _I, _I, _I]),
    "dod_num_tokens": (_I, [_P, _I, _I]),
    "dod_forward": (_I, [_P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "dod_forward_u8": (_I, [_P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "dod_backbone_forward": (_I, [_P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "dod_backbone_prefix": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    "dod_decoder_forward": (_I, [_P, _P, _I, _I, _P, _P, _SZ, _P]),
    "dod_decoder_workspace_bytes": (_SZ, [_P, _I, _I]),
    "dod_set_tap": (_I, [_P, _I, _P]),
    "dod_profile": (_I, [_P, _I]),
    "dod_profile_read": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "dod_op_linear": (_I, [_I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "dod_op_gemm_f32x": (_I, [_P, _I, _I, C.c_longlong, C.c_longlong, _P, _I, _I, C.c_longlong, C.c_longlong, _P, _I, C.c_longlong, C.c_longlong,
                         _I, _I, _I, _I, _I, _F, _I, _I, _P]),
    "dod_op_linear_fp8": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "dod_op_quant_rows_fp8": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "dod_op_quant_mx_fp8": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "dod_op_linear_fp8_glu_mx": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _P, _I, _P, _P]),
    "dod_op_linear_fp8_mx": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "dod_op_linear_fp8_mx2": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P]),
    "dod_op_split_pair": (_I, [_P, _I, _I, _I, _P, _P]),
    "dod_op_linear_x3": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "dod_op_attention_x3": (_I, [_P, _P, _I, _I, _I, _F, _P]),
    "dod_op_split_h2": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "dod_op_linear_h2": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "dod_op_layernorm": (_I, [_P, _P, _P, _P, _F, _I, _I, _P, _I, _P]),
    "dod_op_linear_ln": (_I, [_I, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, C.POINTER(DodLnFold), _P]),
    "dod_op_rowstats": (_I, [_P, _I, _I, _F, _P, _I, _P, _P]),
    "dod_op_ln_finalize": (_I, [_P, _I, _I, _F, _P, _P]),
    "dod_op_attention_bf16": (_I, [_P, _P, _I, _I, _I, _F, _P]),
    "dod_op_attention_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "dod_op_deform_sample": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dod_op_pos_resize": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "dod_op_im2col": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "dod_postprocess_workspace_bytes": (_SZ, [_I, _I, _I]),
    "dod_postprocess": (_I, [_P, _I, _I, _I, _P, _F, _P, C.c_int64, _P, _P, _SZ, _P]),
    "dod_preprocess": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "dod_preprocess_u8": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "dod_match_cost": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _F, _F, _F, _F, _F, _I, _P, _P]),
    "dod_decoder_train_tape_bytes": (_SZ, [C.POINTER(DodConfig), _I, _I]),
    "dod_decoder_train_workspace_bytes": (_SZ, [C.POINTER(DodConfig), _I, _I]),
    "dod_decoder_train_forward": (_I, [C.POINTER(DodConfig), _P, _P, _I, _I, _F, C.c_uint64, _P, _P, _SZ, _P, _SZ, _P]),
    "dod_decoder_train_backward": (_I, [C.POINTER(DodConfig), _P, _P, _I, _I, _F, C.c_uint64, _P, _P, _SZ, _P, _P, _P, _SZ, _P]),
    "dod_decoder_train_last_error": (C.c_char_p, []),
    "dod_dense_decoder_train_tape_bytes": (_SZ, [C.POINTER(DodConfig), _I, _I]),
    "dod_dense_decoder_train_workspace_bytes": (_SZ, [C.POINTER(DodConfig), _I, _I]),
    "dod_dense_decoder_train_forward": (_I, [C.POINTER(DodConfig), _P, _P, _I, _I, _F, C.c_uint64, _P, _P, _SZ, _P, _SZ, _P]),
    "dod_dense_decoder_train_backward": (_I, [C.POINTER(DodConfig), _P, _P, _I, _I, _F, C.c_uint64, _P, _P, _SZ, _P, _P, _P, _SZ, _P]),
    "dod_backbone_tail_tape_bytes": (_SZ, [C.POINTER(DodConfig), _I, _I, _I]),
    "dod_backbone_tail_workspace_bytes": (_SZ, [C.POINTER(DodConfig), _I, _I, _I]),
    "dod_backbone_tail_train_forward": (_I, [C.POINTER(DodConfig), _P, _P, _I, _I, _P, _P, _SZ, _P, _SZ, _P]),
    "dod_backbone_tail_train_backward": (_I, [C.POINTER(DodConfig), _P, _I, _I, _P, _P, _SZ, _P, _P, _SZ, _P]),
    "dod_reserve_gemm_scratch": (_I, [C.c_size_t]),
    "dod_test_set_option": (_I, [C.c_char_p, _I]),
    "dod_test_counter": (C.c_long, [C.c_char_p]),
    "dod_version": (C.c_char_p, []),
    "dod_abi_version": (_I, []),
    "dod_device_count": (_I, []),
}

_lib = None
ABI_VERSION = 4     # include/dinodet.h DOD_ABI_VERSION


def lib():
    """Load the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # libdinodet.so must share ONE HIP runtime with PyTorch (device pointers and streams cross the
        # ABI).  The PyTorch wheel bundles its own libamdhip64.so and publishes it in the global symbol
        # scope; importing torch first makes libdinodet's hip* references bind to that copy.  Loaded the
        # other way round, the process would hold two HIP runtimes.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python -m dinov2_od_amd._build`.")
        L = C.CDLL(LIB_PATH)
        other = bool(os.environ.get("DINODET_LIB"))      # an A/B build may be an earlier ABI revision: its missing entry points stay unbound
        for name, (res, args) in SYMBOLS.items():
            if other and not hasattr(L, name):
                continue
            fn = getattr(L, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in TUNING_SYMBOLS.items():
            if hasattr(L, name):
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
        if L.dod_abi_version() != ABI_VERSION and not other:
            raise RuntimeError(f"{LIB_PATH} has ABI revision {L.dod_abi_version()}, this binding expects {ABI_VERSION}: rebuild it "
                               "(`python -m dinov2_od_amd._build --force`)")
        try:
            if torch.cuda.is_available() and L.dod_device_count() <= 0:
                raise RuntimeError("libdinodet.so is bound to a different HIP runtime than PyTorch "
                                   "(it sees no device): import torch before loading it")
        except AttributeError:
            pass
        _lib = L
    return _lib


def set_option(name, value):
    """dod_test_set_option (include/dinodet.h): a process-wide test option; -1 hands the shipped behaviour back"""
    check(lib().dod_test_set_option(name.encode(), int(value)))


class DodError(RuntimeError):
    pass


_EXC = {1: ValueError, 2: KeyError, 3: RuntimeError, 4: RuntimeError}


def check(rc, handle=None):
    if rc != 0:
        msg = lib().dod_last_error(handle)
        msg = msg.decode() if msg else f"dinodet error {rc}"
        raise _EXC.get(rc, DodError)(msg)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
