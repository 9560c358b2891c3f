"""ctypes binding of libmmae_hip.so (the C ABI declared in include/mmae.h).

The library is built in-tree by `make` (or `__graft_entry__.build()`); there is no fallback: if it is
missing, importing the kernels raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmae_hip.so")

_p, _i, _l, _f, _z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/mmae.h one to one
SIGNATURES = {
    "mmae_last_error": (C.c_char_p, []),
    "mmae_version": (_i, []),
    "mmae_gemm_nt": (_i, [_i, _p, _l, _p, _l, _i, _i, _i, _p, _p, _l, _p, _p, _l, _p]),
    "mmae_gemm_nt_resid_ln": (_i, [_p, _l, _p, _l, _i, _i, _i, _p, _p, _p, _l, _p, _p, _f, _p, _p, _p, _p]),
    "mmae_gemm_tn_set_variant": (None, [_i]),
    "mmae_gemm_nt_set_half_tiles": (None, [_i]),
    "mmae_gemm_nt_set_phase": (None, [_i]),
    "mmae_gemm_nt_last_full_rows": (_i, []),
    "mmae_gemm_nt_last_tail_rows": (_i, []),
    "mmae_gemm_nt_last_kernel": (_i, []),
    "mmae_gemm_tn_last_kernel": (_i, []),
    "mmae_gemm_tn_set_phase": (None, [_i]),
    "mmae_gemm_nt_set_variant": (None, [_i]),
    "mmae_patch_embed_gemm": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _i, _p, _l, _p]),
    "mmae_unpatchify_gemm": (_i, [_p, _l, _p, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mmae_gemm_tn_workspace_bytes": (_z, [_i, _i, _i]),
    "mmae_gemm_tn": (_i, [_p, _l, _p, _l, _i, _i, _i, _p, _i, _p, _i, _p, _z, _p]),
    "mmae_colsum_workspace_bytes": (_z, [_i, _i]),
    "mmae_colsum_bf16": (_i, [_p, _l, _i, _i, _p, _i, _p, _z, _p]),
    "mmae_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "mmae_layernorm_bwd_workspace_bytes": (_z, [_i, _i]),
    "mmae_layernorm_param_reduce": (_i, [_p, _z, _i, _i, _p, _p, _i, _p]),
    "mmae_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "mmae_layernorm_bwd_set_variant": (_i, [_i]),
    "mmae_attn_fwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _p, _i, _i, _i, _i, _i, _f, _p]),
    "mmae_attn_fwd_set_variant": (None, [_i]),
    "mmae_attn_fwd_set_one_pass": (None, [_i]),
    "mmae_attn_max_len": (_i, [_i, _i]),
    "mmae_attn_bwd_workspace_bytes": (_z, [_i, _i, _i]),
    "mmae_attn_bwd_set_variant": (None, [_i]),
    "mmae_attn_bwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _p, _p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _i, _f, _p, _z, _p]),
    "mmae_attn_bwd_chunk": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _p, _p, _p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _i, _f, _p, _z,
                                 _p]),
    "mmae_mask_generate": (_i, [_p, _p, _p, C.POINTER(_i), _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "mmae_pack_patches": (_i, [C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_p), C.POINTER(_i), _i, _i, _i, _i, _p, _i,
                                _p, _i, _p]),
    "mmae_build_token_table": (_i, [C.POINTER(_p), C.POINTER(_p), _i, _i, _i, _p, _p]),
    "mmae_append_global": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "mmae_embed_bwd_reduce": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, C.POINTER(_p), _p, _p]),
    "mmae_decoder_assemble_fwd": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "mmae_decoder_assemble_bwd_workspace_bytes": (_z, [_i, _i, _i]),
    "mmae_decoder_assemble_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _l, _p, C.POINTER(_p), _p, _z, _p]),
    "mmae_build_decoder_emb": (_i, [C.POINTER(_p), _p, _i, _i, _i, _p, _p]),
    "mmae_patchify_bf16": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "mmae_class_emb_bwd_workspace_bytes": (_z, [_i, _i]),
    "mmae_class_emb_bwd": (_i, [_p, _l, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _z, _p]),
    "mmae_masked_loss_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "mmae_masked_ce_workspace_bytes": (_z, [_i, _i, _i]),
    "mmae_masked_ce": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _z, _p]),
    "mmae_masked_loss": (_i, [_i, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "mmae_masked_loss_static": (_i, [_i, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "mmae_standardize_depth": (_i, [_p, _p, _i, _l, C.c_double, C.c_double, _f, _p]),
    "mmae_adamw_flat": (_i, [_p, _p, _p, _p, _l, _l, _f, _f, _f, _f, _f, _i, _f, _p, _p, _f, _f, _p, _p]),
    "mmae_adamw_flat_parts": (_i, [_l]),
    "mmae_adamw_flat_ex": (_i, [_p, _p, _p, _p, _l, _l, _f, _f, _f, _f, _f, _i, _f, _p, _p, _f, _f, _p, _i, _p, _p]),
    "mmae_sum_partials": (_i, [_p, _i, _p, _p]),
    "mmae_sumsq": (_i, [_p, _l, _p, _p, _z, _p]),
    "mmae_cast_transpose": (_i, [_p, _i, _i, _p, _p, _p]),
    "mmae_transpose_multi": (_i, [_p, _p, _p, _i, _i, _p]),
    "mmae_cast_bf16": (_i, [_p, _p, _l, _p]),
    "mmae_split_f32": (_i, [_p, _l, _i, _i, _p, _i, _i, _i, _p]),
    "mmae_gelu_f32": (_i, [_p, _p, _l, _p]),
    "mmae_gelu_bwd_f32": (_i, [_p, _p, _p, _l, _p]),
    "mmae_layernorm_fwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "mmae_layernorm_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _z, _p]),
    "mmae_attn_f32_fwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _p, _i, _i, _i, _i, _i, _f, _p]),
    "mmae_attn_f32_bwd_workspace_bytes": (_z, [_i, _i, _i]),
    "mmae_attn_f32_bwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _p, _p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _i, _f, _p, _z, _p]),
    "mmae_comm_unique_id": (_i, [_p]),
    "mmae_comm_init": (_i, [_p, _i, _i, C.POINTER(_p)]),
    "mmae_comm_allreduce_bucket": (_i, [_p, _p, _z, _i, _p]),
    "mmae_comm_broadcast": (_i, [_p, _p, _z, _i, _i, _p]),
    "mmae_comm_info": (_i, [_p, C.POINTER(_i), C.POINTER(_i)]),
    "mmae_comm_destroy": (_i, [_p]),
    "mmae_spin_us": (_i, [_i, _p]),
    "mmae_set_cu_budget": (_i, [_i]),
    "mmae_get_cu_budget": (_i, []),
}


class MmaeError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library and bind every declared symbol; raises if either is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmaeError(f"{LIB_PATH} is missing: build it with `make` (there is no non-HIP fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mmae_last_error()
        raise MmaeError(f"{what or 'mmae'} failed (code {rc}): {msg.decode() if msg else ''}")
