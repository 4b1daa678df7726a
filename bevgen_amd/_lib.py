"""ctypes binding of libbevgen_hip.so (C ABI declared in include/bevgen_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is raised.
``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C bevgen_amd/csrc``) builds it for gfx950.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BEVGEN_LIB_PATH") or os.path.join(_HERE, "csrc", "libbevgen_hip.so")   # override: A/B runs of two builds on one GPU box
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "bevgen_hip.h")

ABI_VERSION = 6
ROUTE_MASKGIT, ROUTE_AR = 0, 1
PRECISION_FP32, PRECISION_BF16, PRECISION_F16X3 = 0, 1, 2
KV_F32, KV_F16 = 0, 1
DECODE_FUSED, DECODE_PER_OP, DECODE_SPLIT, DECODE_AUTO = 0, 1, 2, 3
VQ_OUT_RAW, VQ_OUT_DENORM, VQ_OUT_U8 = 0, 1, 2
W_F32, W_F16 = 0, 1
DTYPE_F32, DTYPE_I64, DTYPE_U8, DTYPE_F64 = 0, 1, 2, 3
ERR_NUMERIC = -5
STATUS_MLP_BARRIER, STATUS_MLP_PLACEMENT, STATUS_NONFINITE_LOGITS, STATUS_F16_RANGE, STATUS_NONFINITE_PIXELS = 1, 2, 4, 8, 16


class BevgenError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libbevgen_hip error {code}: {message}")
        self.code = code


class bevgen_cfg(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("route", C.c_int32), ("precision", C.c_int32),
        ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("dim", C.c_int32), ("vocab_size", C.c_int32), ("cond_vocab_size", C.c_int32),
        ("num_cams", C.c_int32), ("cam_latent_h", C.c_int32), ("cam_latent_w", C.c_int32),
        ("num_cond_tokens", C.c_int32), ("seq_len", C.c_int32), ("sparse_block_size", C.c_int32),
        ("image_embed", C.c_int32), ("bev_embed", C.c_int32), ("camera_bias", C.c_int32),
        ("ff_inner", C.c_int32), ("max_batch", C.c_int32),
        ("vq_ch", C.c_int32), ("vq_num_res_blocks", C.c_int32), ("vq_z_channels", C.c_int32), ("vq_embed_dim", C.c_int32),
        ("vq_n_embed", C.c_int32), ("vq_resolution", C.c_int32), ("vq_out_ch", C.c_int32), ("vq_num_levels", C.c_int32),
        ("vq_ch_mult", C.c_int32 * 8), ("vq_attn_resolution", C.c_int32), ("vq_in_channels", C.c_int32), ("kv_cache_dtype", C.c_int32), ("decode_path", C.c_int32), ("decode_weight_dtype", C.c_int32), ("weight_dtype", C.c_int32), ("decode_chains", C.c_int32), ("reserved", C.c_int32 * 10),
    ]


_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_l = C.c_long

# name -> (restype, argtypes); kept in the same order as include/bevgen_hip.h
SIGNATURES = {
    "bevgen_create": (_i, [C.POINTER(bevgen_cfg), _i, C.POINTER(_p)]),
    "bevgen_destroy": (None, [_p]),
    "bevgen_last_error": (C.c_char_p, [_p]),
    "bevgen_abi_version": (_i, []),
    "bevgen_synchronize": (_i, [_p, _p]),
    "bevgen_status": (_i, [_p, C.POINTER(C.c_uint)]),
    "bevgen_load_tensor": (_i, [_p, C.c_char_p, _p, _i, _i, C.POINTER(C.c_int64)]),
    "bevgen_set_tables": (_i, [_p, _p, _p, _p, _p, _p]),
    "bevgen_finalize": (_i, [_p]),
    "bevgen_muse_forward": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "bevgen_maskgit_generate": (_i, [_p, _p, _p, _p, _i, _i, C.POINTER(C.c_int32), _f, _i, _f, _p, _p, _p, _p, C.c_uint64, _p]),
    "bevgen_maskgit_generate_ex": (_i, [_p, _p, _p, _p, _i, _i, C.POINTER(C.c_int32), _f, _i, _f, _p, _p, _p, _p, C.c_uint64, _i, _i, _p]),
    "bevgen_op_philox_uniform": (_i, [_p, C.c_uint64, C.c_uint, C.c_uint, _i, _l, _p, _p]),
    "bevgen_sparse_self_attention": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "bevgen_ar_prefill": (_i, [_p, _p, _p, _p, _i, _p]),
    "bevgen_ar_logits": (_i, [_p, _p, _p]),
    "bevgen_ar_decode_step": (_i, [_p, _p, _p]),
    "bevgen_ar_sample": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _i, _p, _p, _p]),
    "bevgen_ar_sample_forced": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _i, _p, _p, _p, _p]),
    "bevgen_vq_decode": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "bevgen_vq_decode_latents": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "bevgen_vq_encode": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "bevgen_op_gemm": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "bevgen_op_ln_gemm": (_i, [_p, _p, _p, _p, _f, _p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(_i), _p]),
    "bevgen_op_mlp_fused": (_i, [_p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _p, _i, _i, _p]),
    "bevgen_op_layernorm": (_i, [_p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "bevgen_op_geglu_layernorm": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "bevgen_op_attention": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p]),
    "bevgen_op_attention_ex": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p, _p]),
    "bevgen_op_decode_attention": (_i, [_p, _p, _p, _p, _i, _p, _i, _p, _i, _l, _i, _i, _i, _i, _f, _p, _p]),
    "bevgen_op_ar_attn_fused": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "bevgen_op_conv3x3": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "bevgen_op_groupnorm": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "bevgen_decode_attention_splits": (_i, [_i, _i, _i]),
    "bevgen_profile_begin": (_i, [_p]),
    "bevgen_profile_end": (_i, [_p, C.POINTER(C.c_double)]),
    "bevgen_ar_step_timing": (_i, [_p, _i]),
    "bevgen_ar_step_times": (_i, [_p, C.POINTER(C.c_float), _i, C.POINTER(_i)]),
    "bevgen_set_trace_buffer": (_i, [_p, _p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the library (once) and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc --offload-arch=gfx950) or `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback."
        )
    # torch ships its own libamdhip64: it has to be in the process BEFORE this library is mapped, so that the NEEDED entry of libbevgen_hip.so resolves to
    # the runtime torch uses (one HIP runtime per process; with the opposite order the second runtime reports "no ROCm-capable device")
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        except AttributeError:
            if os.environ.get("BEVGEN_LIB_PATH"):   # A/B runs against an OLDER build (tools/ab_*.sh): entry points it lacks stay unbound
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if lib.bevgen_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libbevgen_hip ABI {lib.bevgen_abi_version()} != binding ABI {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(ctx, code: int) -> None:
    if code != 0:
        msg = load().bevgen_last_error(ctx)
        raise BevgenError(code, msg.decode() if msg else "unknown error")
