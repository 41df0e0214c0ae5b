"""ctypes binding of libclip_amd.so (include/clip_amd.h).  There is no fallback: if the HIP
library is missing or a call fails, this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLIP_AMD_LIB") or os.path.join(_HERE, "lib", "libclip_amd.so")

CLIP_OK, CLIP_ERR_ARG, CLIP_ERR_HIP, CLIP_ERR_STATE, CLIP_ERR_CAPACITY, CLIP_ERR_SEP_COUNT = 0, -1, -2, -3, -4, -5
KINDS = ("embed_ln", "gemm_qkv", "attention", "gemm_attn_out", "layernorm", "gemm_ffn_up",
         "gemm_ffn_down", "head", "last_layer_rows")
PRECISION_BF16, PRECISION_FP32, PRECISION_F16, PRECISION_F16R = 0, 1, 2, 3
PRECISIONS = {"bf16": PRECISION_BF16, "fp32": PRECISION_FP32, "f16": PRECISION_F16, "f16r": PRECISION_F16R}
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_BIAS_TANH = 0, 1, 2, 3


class ClipConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab_size", "hidden_size", "num_layers", "num_heads",
                                         "intermediate_size", "max_position", "type_vocab_size",
                                         "num_labels")] + \
               [("layer_norm_eps", C.c_float), ("max_tokens", C.c_int32), ("max_seqs", C.c_int32)]


class ClipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libclip_amd error {code}: {msg}")
        self.code = code


_p, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes; every symbol include/clip_amd.h declares (tests check the export list)
SIGNATURES = {
    "clip_create": [C.POINTER(ClipConfig), C.c_int, C.POINTER(_p)],
    "clip_destroy": [_p],
    "clip_load_weight": [_p, C.c_char_p, _p, _i64],
    "clip_finalize_weights": [_p],
    "clip_set_precision": [_p, _i32],
    "clip_focus_index": [_p, _p, _i32, _i32, _i32, _p, _p, _p],
    "clip_prepare_packed": [_p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p],
    "clip_forward_packed": [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p, _p],
    "clip_forward_padded": [_p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p],
    "clip_epilogue": [_p, _p, _f, _i32, _i32, _p, _p, _p, _p, _p],
    "clip_bce_loss": [_p, _p, _i32, _p, _p],
    "clip_mlm_head": [_p, _p, _p, _i32, _p, _p, _i32, _p],
    "clip_ce_loss": [_p, _p, _i32, _i32, _p, _p, _i32, _p],
    "clip_gemm_bf16": [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    "clip_gemm_f16": [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    "clip_gemm_bf16_lnres": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    "clip_gemm_bf16_lnout": [_p, _p, _p, _p, _i32, _p, _p, _f, _p, _p, _p, _p, C.c_uint32, _p,
                             _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    "clip_gemm_bf16_lnout_noalloc": [_p, _p, _p, _p, _i32, _p, _p, _f, _p, _p, _p,
                                     _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    "clip_gemm_f16_lnout_inplace": [_p, _p, _p, _p, _p, C.c_float, _p, _p, _p, C.c_uint32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    "clip_gemm_f16_lnout_res32": [_p, _p, _p, _p, _p, C.c_float, _p, _p, _p, _p, C.c_uint32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    "clip_gemm_f32": [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    "clip_attention_varlen": [_p, _p, _p, _i32, _i32, _i32, _i32, _p],
    "clip_attention_varlen_f16": [_p, _p, _p, _i32, _i32, _i32, _i32, _p],
    "clip_layernorm": [_p, _p, _p, _f, _p, _i32, _i32, _p],
    "clip_layernorm_stats": [_p, _p, _p, _f, _p, _p, _i32, _i32, _p],
    "clip_embed_layernorm": [_p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i32, _i32, _i32, _i32, _i32, _p],
    "clip_set_trim_last_layer": [_p, _i32],
    "clip_set_fused_layernorm": [_p, _i32],
    "clip_poll_errors": [_p, _p],
    "clip_preferred_step_tokens": [_p, _p],
    "clip_profile_enable": [_p, C.c_uint32],
    "clip_profile_read": [_p, _p, _p, _i32],
    "clip_mfma_peak": [_i32, _i32, _i32, _p, _p, _p],
    "clip_last_error": [],
    "clip_abi_version": [],
}

_lib = None


def load():
    """Load the in-tree library (built by __graft_entry__.build() / make -C clip_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_char_p if name == "clip_last_error" else C.c_int
    _lib = lib
    return lib


ERR_SEP_COUNT, ERR_LN_TIMEOUT, ERR_ROW_TABLE, ERR_NONFINITE = 1, 2, 4, 8     # bits of the device error word (clip_poll_errors)


def poll_errors(engine):
    """Read-and-clear the engine's device error word behind its latest forward; raises if non-zero."""
    flags = C.c_int32(0)
    check(load().clip_poll_errors(engine, C.byref(flags)))
    raise_for_flags(flags)


def raise_for_flags(flags):
    """the exception of a non-zero device error word (clip_poll_errors, clip_forward_padded's error_flags)"""
    if flags.value & ERR_LN_TIMEOUT:
        raise RuntimeError(f"engine error word {flags.value}: a fused-LayerNorm exchange timed out (GPU shared "
                           "with other work? set CLIP_AMD_FUSED_LN=0 there); results of the call are void")
    if flags.value & ERR_ROW_TABLE:
        raise ValueError(f"engine error word {flags.value}: inconsistent row table (a row longer than max_seqlen, "
                         "an empty row, or cu_seqlens not ending at total_tokens); results of the call are void")
    if flags.value & ERR_SEP_COUNT:
        # reference: prints and returns None, then classifier(None) raises (bert_model.py:51-55,91)
        print("error when pulling out end of sentence indices: a row does not hold exactly 2 * n_context_sentences + 1 "
              "[SEP] tokens")
        raise TypeError("focus-sentence representation is None: wrong [SEP] count in a row")
    if flags.value & ERR_NONFINITE:
        raise FloatingPointError(f"engine error word {flags.value}: non-finite logits (an activation left the operand "
                                 "type's range — fp16 tops out at 65504: use precision 'bf16' or 'fp32' for this "
                                 "checkpoint — or a weight is NaN); results of the call are void")
    if flags.value:
        raise RuntimeError(f"engine error word {flags.value}")


def check(rc):
    if rc != CLIP_OK:
        raise ClipError(rc, load().clip_last_error().decode())


def ptr(t):
    """device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
