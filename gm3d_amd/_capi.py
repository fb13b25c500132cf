"""ctypes binding of libgm3d_hip.so -- the C ABI declared in include/gm3d.h.

There is NO CPU fallback anywhere in this package: if the library is missing the
import fails loudly, and every op rejects non-GPU tensors.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GM3D_HIP_LIB") or os.path.join(_HERE, "lib", "libgm3d_hip.so")

GM3D_OK, GM3D_EINVAL, GM3D_EUNSUPPORTED, GM3D_ELAUNCH = 0, -1, -2, -3
GM3D_F32, GM3D_BF16 = 0, 1

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> argtypes; mirrors include/gm3d.h one to one (tests/test_capi_symbols.py checks it)
SIGNATURES = {
    "gm3d_abi_version": [],
    "gm3d_strerror": [_i],
    "gm3d_fps": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "gm3d_gather_points": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "gm3d_gather_points_grad": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "gm3d_knn": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "gm3d_knn_group": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "gm3d_chamfer_fwd": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "gm3d_chamfer_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "gm3d_patch_chamfer_loss_fwd": [_vp, ctypes.c_longlong, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "gm3d_patch_chamfer_loss_bwd": [_vp, ctypes.c_longlong, _vp, _vp, ctypes.c_longlong, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp],
    "gm3d_attention_fwd": [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "gm3d_attention_qkv_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "gm3d_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "gm3d_ln_plain_partial_rows": [_i],
    "gm3d_ln_set_grid_cap": [_i],
    "gm3d_ln_plain_fwd": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gm3d_ln_plain_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gm3d_residual_ln_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gm3d_residual_ln_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gm3d_ln_partial_rows": [_i],
    "gm3d_colsum_finish": [_vp, _i, _i, _i, _vp, _i, _vp],
    "gm3d_colsum_finish_batched": [_vp, _i, ctypes.c_longlong, _i, _i, _i, _vp, _i, _vp],
    "gm3d_bias_gelu_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "gm3d_bias_gelu_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gm3d_gelu_partial_rows": [_i],
    "gm3d_embed_partial_rows": [_i, _i, _i],
    "gm3d_moments3": [_vp, _i, _vp, _vp],
    "gm3d_pn_layer1_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gm3d_group_max_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "gm3d_group_max_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "gm3d_bn_bcast_stats": [_vp, _vp, _i, _i, _i, _vp, _i, _vp],
    "gm3d_bn_bcast_apply_relu": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "gm3d_bn_bcast_bwd_stats": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _i, _vp],
    "gm3d_bn_bcast_bwd_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "gm3d_head_fold": [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp],
    "gm3d_head_rowdot": [_vp, _vp, _vp, _i, _i, _vp, _i, _vp],
    "gm3d_head_fold_bwd": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "gm3d_head_outer": [_vp, _vp, _i, _i, _vp, _i, _vp],
    "gm3d_scale_translate": [_vp, _vp, _f, _f, _f, _i, _i, _vp],
    "gm3d_group_select_maps": [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "gm3d_bn_bcast_apply_relu_sel": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "gm3d_bn_bcast_bwd_stats_sel": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _i, _vp],
    "gm3d_bn_bcast_bwd_apply_sel": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "gm3d_lin3_gelu_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gm3d_lin3_gelu_bwd": [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp],
    "gm3d_lin3_finish": [_vp, _i, _i, _vp, _vp, _vp],
    "gm3d_rank_loss": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "gm3d_rank_loss_tail": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "gm3d_rank_loss_tail_bwd": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "gm3d_ema_counters": [_vp, _vp, _i, _f, _f, _vp],
    "gm3d_drop_path_scales": [_vp, _vp, _i, _i, _vp, _vp],
    "gm3d_pad_cols": [_vp, ctypes.c_longlong, _i, _i, _vp, _i, _i, _vp],
    "gm3d_patch_chamfer_loss_bwd_full": [_vp, ctypes.c_longlong, _vp, _vp, ctypes.c_longlong, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp],
    "gm3d_mask_select_b": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "gm3d_pn1_bwd_finalize": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "gm3d_gemm_tn_bf16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_gelu": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_pool": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_gelu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_res": [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_lna": [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tile_rows": [_i],
    "gm3d_gemm_tn_bf16_ring": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_ring_set_depth": [_i, _i],
    "gm3d_gemm_tn_bf16_ws": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_ws_pool": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_ws_supported": [_i, _i, _i],
    "gm3d_gemm_tn_bf16_ws_poolg": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_nt_set_big_tiles": [_i],
    "gm3d_gemm_nt_set_order": [_i],
    "gm3d_gemm_nt_bf16_sum": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _i, _vp],
    "gm3d_gemm_nt_tiles": [_i, _i],
    "gm3d_gemm_nt_bf16_multi": [_i] + [_vp] * 14 + [_vp],
    "gm3d_gemm_tn_bf16_ws_bn_apply": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_ws_bn_stats": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_ws_stats_rows": [_i, _i, _i],
    "gm3d_gemm_tn_bf16_ws_bn_apply_g": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_ws_bn_stats_g": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_ws_set_occupancy": [_i],
    "gm3d_gemm_tn_bf16_ring96": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_dma": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_dma_gelu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_dma_gelu": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_transpose_bf16_batched": [_vp, _vp, _i, _i, _i, ctypes.c_longlong, _vp],
    "gm3d_transpose_bf16_multi": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gm3d_token_assemble_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "gm3d_token_assemble_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "gm3d_sum_few_rows": [_vp, _i, _i, ctypes.c_longlong, _vp, _vp],
    "gm3d_gemm_nt_bf16": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _i,
                          ctypes.c_longlong, _vp],
    "gm3d_gemm_nt_splits": [_i, _i, _i, _i],
    "gm3d_radius_mask_bits": [_vp, _vp, _f, _i, _i, _vp, _vp],
    "gm3d_radius_mask_bits_m": [_vp, _vp, _i, _f, _i, _i, _vp, _vp],
    "gm3d_attention_masked_set_wide": [_i],
    "gm3d_attention_masked_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "gm3d_attention_masked_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "gm3d_mask_select": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "gm3d_bn_finalize": [_vp, ctypes.c_double, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "gm3d_pn1_finalize": [_vp, ctypes.c_double, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "gm3d_flat_partial_rows": [ctypes.c_longlong],
    "gm3d_adamw_ema_flat_step": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _f, _f, _f, _f,
                                 _vp, _f, _vp, _vp, _vp, _vp],
    "gm3d_adamw_ema_flat_step_lrd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _vp, _f, _f,
                                     _f, _f, _vp, _f, _vp, _vp, _vp, _vp],
    "gm3d_group_scatter_add": [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp],
    "gm3d_pn_layer1_bwd_stats": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp],
    "gm3d_colsum_finish_f64": [_vp, _i, _i, _i, _vp, _vp],
    "gm3d_colsum_partial": [_vp, _i, _i, _vp, _i, _vp],
    "gm3d_gather_inverse": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "gm3d_gather_rows_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "gm3d_add_ln_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "gm3d_add_ln_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "gm3d_add_ln_bwd_acc": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "gm3d_partition_visible": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gm3d_select_rows": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "gm3d_back_project": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "gm3d_interp3_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gather_rows_bwd_w": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "gm3d_where_rows": [_vp, _i, _vp, _vp, _i, _vp, ctypes.c_longlong, _i, _i, _vp],
    "gm3d_take_rows": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "gm3d_colsum_partial_w": [_vp, _vp, _i, _i, _vp, _i, _vp],
    "gm3d_gemm_tn_bf16_dma_pool": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "gm3d_gemm_tn_bf16_dmaw": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
}


class Gm3dError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "gm3d_amd: %s not found. Build it with `python -m gm3d_amd.build` "
            "(hipcc, --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name == "gm3d_strerror" else ctypes.c_int
    return lib


lib = _load()


def check(rc, what):
    if rc != GM3D_OK:
        raise Gm3dError("%s failed: %s (code %d)" % (what, lib.gm3d_strerror(rc).decode(), rc))
