"""ctypes binding of libmeerqat_hip.so (C ABI: include/meerqat_hip.h).

The library is the product; if it is missing this module raises -- nothing falls back to a CPU
path.  Loading it and resolving its symbols needs no GPU; calling a compute entry does.
"""
import ctypes
import os

from . import build as _build

_LIB = None

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_sz = ctypes.c_size_t
c_ptr = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/meerqat_hip.h
SIGNATURES = {
    "mq_version": (ctypes.c_char_p, []),
    "mq_strerror": (ctypes.c_char_p, [c_int]),
    "mq_last_hip_error": (c_int, []),
    "mq_padded_rows": (c_i64, [c_i64]),
    "mq_padded_dim": (c_int, [c_int]),
    "mq_packed_bytes": (c_sz, [c_i64, c_int]),
    "mq_pack_rows_f32": (c_int, [c_ptr, c_i64, c_int, c_i64, c_int, c_ptr, c_i64, c_ptr, c_ptr]),
    "mq_unpack_rows_f32": (c_int, [c_ptr, c_i64, c_int, c_i64, c_i64, c_ptr, c_ptr]),
    "mq_l2norm_rows_f32": (c_int, [c_ptr, c_i64, c_int, c_ptr]),
    "mq_l2norm_rows_form_f32": (c_int, [c_ptr, c_i64, c_int, c_int, c_ptr]),
    "mq_knn_workspace_bytes": (c_sz, [c_i64, c_int, c_int, c_int]),
    "mq_knn_workspace_bytes_metric": (c_sz, [c_i64, c_int, c_int, c_int, c_int]),
    "mq_knn_search_f32": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_int, c_int, c_int, c_int, c_i64,
                                  c_ptr, c_ptr, c_ptr, c_sz, c_ptr]),
    "mq_knn_search_f32_ev": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_int, c_int, c_int, c_int, c_i64,
                                     c_ptr, c_ptr, c_ptr, c_sz, c_ptr, c_ptr, c_ptr]),
    "mq_knn_screen_bytes": (c_sz, [c_i64, c_int, c_int]),
    "mq_knn_screen_prepare": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_int, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mq_diag_mfma_bf16_loop": (c_int, [c_int, c_int, c_int, c_ptr, c_ptr]),
    "mq_diag_mfma_bf16_dot": (c_int, [c_ptr, c_ptr, c_int, c_ptr, c_ptr]),
    "mq_knn_screen_add_rows_f32": (c_int, [c_ptr, c_i64, c_int, c_i64, c_int, c_int, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mq_knn_search_screened_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_int, c_int, c_int, c_int,
                                           c_i64, c_ptr, c_ptr, c_ptr, c_sz, c_ptr, c_ptr, c_ptr]),
    "mq_knn_screen_stats": (c_int, [c_i64, c_int, c_int, c_int, c_ptr, ctypes.POINTER(c_i64), c_ptr]),
    "mq_knn_launch_info": (c_int, [c_i64, c_int, c_int, c_int, ctypes.POINTER(c_i64)]),
    "mq_knn_screen_scan_kind": (c_int, [c_i64, c_int, c_int, c_int, c_int]),
    "mq_knn_set_option": (c_int, [c_int, c_int]),
    "mq_knn_get_option": (c_int, [c_int]),
    "mq_im2col_split_f32": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr]),
    "mq_warp_affine_faces_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr]),
    "mq_gemm_nt_bf16x3s_respair_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "mq_gemm_nt_bf16x3s_splitk_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "mq_stem_conv3x3_f32": (c_int, [c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mq_conv3x3_pair_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                    c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr]),
    "mq_gemm_nt_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "mq_split_bf16_f32": (c_int, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "mq_split_bf16_tiled_elems": (c_i64, [c_int, c_int]),
    "mq_split_bf16_tiled_f32": (c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "mq_gemm_nt_bf16x3_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "mq_layernorm_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, ctypes.c_float, c_ptr]),
    "mq_bert_embed_ln_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int,
                                     ctypes.c_float, c_ptr]),
    "mq_attention_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, ctypes.c_float, c_ptr]),
    "mq_clip_patchify_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "mq_clip_assemble_ln_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, ctypes.c_float,
                                        c_ptr]),
    "mq_attention_causal_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, ctypes.c_float, c_int, c_ptr]),
    "mq_clip_text_embed_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr]),
    "mq_clip_text_embed_packed_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr]),
    "mq_clip_eos_pool_ln_f32": (c_int, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, ctypes.c_float, c_ptr]),
    "mq_gemm_nt_bf16x3s_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int,
                                       c_ptr]),
    "mq_layernorm_split_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, ctypes.c_float, c_ptr]),
    "mq_bert_embed_ln_split_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int,
                                           c_int, ctypes.c_float, c_ptr]),
    "mq_attention_split_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, ctypes.c_float, c_int,
                                       c_int, c_ptr]),
    "mq_bert_embed_ln_packed_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int,
                                            ctypes.c_float, c_ptr]),
    "mq_attention_packed_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_int, c_int, ctypes.c_float, c_int,
                                        c_int, c_ptr]),
    "mq_sum_groups_f32": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr]),
    "mq_topk_merge_f32": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "mq_shard_record_bytes": (c_sz, [c_int, c_int]),
    "mq_shard_record_ids_offset": (c_sz, [c_int, c_int]),
    "mq_topk_merge_records_f32": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "mq_image_plan": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "mq_image_preprocess_u8": (c_int, [c_ptr, c_ptr, c_int, c_ptr, c_int, c_int, c_int, c_int, ctypes.c_double, c_ptr, c_ptr,
                                       c_ptr, c_ptr, c_sz, c_ptr]),
    "mq_jpeg_probe": (c_int, [c_ptr, c_sz, c_ptr]),
    "mq_jpeg_read_coefficients": (c_int, [c_ptr, c_sz, c_ptr, c_sz]),
    "mq_jpeg_decode_rgb_u8": (c_int, [c_ptr, c_ptr, c_int, c_int, c_i64, c_ptr]),
    "mq_fuse_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "mq_fuse_wsum_f64": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_sz,
                                 c_ptr]),
    "mq_gemm_set_option": (c_int, [c_int, c_int]),
    "mq_run_metrics_f64": (c_int, [c_ptr, c_int, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "mq_fuse_fit_wsum_f64": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_int,
                                     c_int, c_ptr, c_ptr, c_ptr, c_sz, c_ptr]),
    "mq_format_run_json": (c_i64, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_int, c_i64, c_ptr, c_ptr, c_i64, c_int, c_ptr]),
}


KNN_OPT_SMALL_SCAN, KNN_OPT_SMALL_MIN_TILES, KNN_OPT_PARTITIONS, KNN_OPT_SMALL_WAVES = 0, 1, 2, 3  # MQ_KNN_OPT_* (include/meerqat_hip.h)


GEMM_OPT_WIDE, GEMM_OPT_STAGGER = 0, 1  # MQ_GEMM_OPT_*


class gemm_option:
    """``with _lib.gemm_option(GEMM_OPT_WIDE, 0): ...`` -- A/B switch of the split-bf16 GEMM kernels for the block."""

    def __init__(self, key, value):
        self.key, self.value = key, value

    def __enter__(self):
        self.old = load().mq_gemm_set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        load().mq_gemm_set_option(self.key, self.old)
        return False


class knn_option:
    """``with _lib.knn_option(KNN_OPT_SMALL_SCAN, 0): ...`` -- an A/B switch of the search paths set for the block through the C
    ABI (mq_knn_set_option), restored afterwards; the environment is only read once, when the library first needs a switch."""

    def __init__(self, key, value):
        self.key, self.value = int(key), int(value)

    def __enter__(self):
        self.previous = load().mq_knn_set_option(self.key, self.value)
        check(min(self.previous, 0), "mq_knn_set_option")
        return self

    def __exit__(self, *exc):
        load().mq_knn_set_option(self.key, self.previous)
        return False


class MeerqatHipError(RuntimeError):
    pass


def lib_path():
    # MEERQAT_HIP_LIB: load another build of the same C ABI (A/B benchmarking of kernel variants)
    return os.environ.get("MEERQAT_HIP_LIB") or _build.SO


def load():
    """Load the HIP library (raises if it has not been built; never builds implicitly on import)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise MeerqatHipError(
            f"{path} is missing: build it with `python -m viquae_amd.build` (hipcc, gfx950). "
            "viquae_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    ab = bool(os.environ.get("MEERQAT_HIP_LIB"))
    for name, (res, args) in SIGNATURES.items():
        if ab and not hasattr(lib, name):
            continue  # A/B against an OLDER build of the ABI (developer switch): entries it lacks stay unbound
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(code, what=""):
    if code != 0:
        lib = load()
        msg = lib.mq_strerror(code).decode()
        extra = f" (hipError {lib.mq_last_hip_error()})" if code == -3 else ""
        raise MeerqatHipError(f"{what or 'libmeerqat_hip'}: {msg}{extra}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise MeerqatHipError("no GPU visible: viquae_amd runs its arithmetic on MI355X only (no CPU fallback)")
