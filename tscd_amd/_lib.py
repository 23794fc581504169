"""ctypes binding of the C-ABI in include/tscd_hip.h.  No fallback: a missing library is an ImportError."""
import ctypes as C
import os

import torch  # noqa: F401  -- load PyTorch-ROCm's bundled libamdhip64.so.7 FIRST so this library binds to the same HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSCD_HIP_LIB") or os.path.join(_HERE, "lib", "libtscd_hip.so")      # (TSCD_HIP_LIB: A/B builds of the timing tools)

P, I, L, F = C.c_void_p, C.c_int, C.c_long, C.c_float

# name -> argtypes (every entry point returns int status, 0 = ok)
SIGNATURES = {
    "tscd_conv2d_nhwc_f16": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "tscd_conv2d_nhwc_f16_ex": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "tscd_conv_bn_act_train_multi_f16": [P, P, P, I, P, I, I, I, I, I, I, I, I, I, P, P, F, F, P, P, P, P, I, I, P],
    "tscd_conv_wgrad_nhwc_f16_ex": [P, P, P, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "tscd_bn_silu_bwd_f16_ex": [P, L, P, L, I, P, L, P, P, P, P, P, I, L, I, P, L, P, P, L, P],
    "tscd_pack_conv_weight": [P, I, P, I, I, I, I, I, I, I, I, P],
    "tscd_pack_conv_weights_batched": [P, P, L, P],
    "tscd_bn_finalize": [P, I, I, I, L, P, P, F, F, P, P, P, P, P, P],
    "tscd_bn_fold": [P, P, P, P, F, I, P, P],
    "tscd_bn_act_nhwc_f16": [P, I, P, I, P, P, I, L, I, I, P],
    "tscd_focus_nchw_to_nhwc16": [P, I, P, I, I, I, P],
    "tscd_copy_slice_nhwc_f16": [P, I, P, I, I, I, I, I, I, P],
    "tscd_maxpool_nhwc_f16": [P, I, P, I, I, I, I, I, I, P],
    "tscd_haar_hf_nhwc_f16": [P, I, P, I, I, I, I, P],
    "tscd_haar_idwt_mul_nhwc_f16": [P, P, I, P, I, I, I, I, I, P],
    "tscd_haar_idwt_mul_bwd_nhwc_f16": [P, P, P, I, P, P, I, I, I, I, P],
    "tscd_ema_update_f32": [P, L, F, F, P],
    "tscd_zero_insert2_nhwc_f16": [P, I, P, I, I, I, I, I, P],
    "tscd_maxpool5_bwd_nhwc_f16": [P, I, P, I, P, I, I, I, I, I, P],
    "tscd_upsample2_bwd_nhwc_f16": [P, I, P, I, I, I, I, I, P],
    "tscd_clip_preproc_u8": [P, I, I, I, P, P, I, I, I, P],
    "tscd_sgd_ema_step_f32": [P, L, P, P, I, F, I, P, P, F, F, P],
    "tscd_grad_gather_check_f32": [P, L, P, F, P, P],
    "tscd_wgrad_batch_f16": [P, I, P, L, P],
    "tscd_colsum_batch_f16": [P, I, P],
    "tscd_sgd_ema_step_indirect_f32": [P, L, P, P, P, I, F, I, P, P, F, F, P],
    "tscd_conv_bn_act_train_f16": [P, P, P, I, P, I, I, I, I, I, I, I, I, I, P, P, F, F, P, P, P, P, I, P, I, I, P],
    "tscd_decode_score": [P, P, P, P, P, P, I, I, P, P, P, P],
    "tscd_select_proposals": [P, I, I, F, I, I, I, P, P, P],
    "tscd_gather_rows": [P, P, P, I, P, I, I, I, I, I, P, P, P, I, P],
    "tscd_gather_det": [P, I, I, P, I, P, I, I, P, P],
    "tscd_gather_det_rows": [P, I, I, P, I, P, I, I, P, P, P, P],
    "tscd_mca_forward": [P, P, P, P, P, P, I, I, I, I, I, I, F, F, F, P, L, P, P, P, P, P, P, I, P],
    "tscd_mca_backward": [P, P, P, P, P, P, I, I, I, I, I, I, F, F, F, P, P, P, P, P, I, P, P, I, P, L, P, P, P, P, P, P],
    "tscd_mca_backward_fws": [P, P, P, P, P, P, I, I, I, I, I, I, F, F, F, P, P, P, P, P, I, P, P, I, P, L, P, P, P, P, P, P, L, P],
    "tscd_cos_attn_f16": [P, P, P, I, I, P, I, P, P],
    "tscd_se_fuse_f16": [P, I, P, I, P, P, P, L, I, P, P],
    "tscd_add_layernorm_f16": [P, P, P, P, I, I, F, P, P],
    "tscd_cos_attn_bwd_f16": [P, P, P, P, I, I, I, I, P, I, P, L, P, P, P, P],
    "tscd_cos_attn_bwd_finish_f16": [P, P, I, I, I, P, P, P],
    "tscd_cos_attn_mfma_f16": [P, P, P, I, I, P, I, P, P, P],
    "tscd_cos_attn_mfma_bwd_f16": [P, P, P, P, P, P, I, I, I, I, P, I, P, L, P, P],
    "tscd_cos_attn_mfma_bwd2_f16": [P, P, P, P, P, P, I, I, I, I, P, I, P, I, P, P, P, P, P],
    "tscd_se_fuse_bwd_f16": [P, I, P, I, P, P, P, L, I, P, P, P, P, P, P],
    "tscd_add_layernorm_bwd_f16": [P, P, P, P, I, I, F, P, P, P, P],
    "tscd_se_fuse_bwd_acc_f16": [P, I, P, I, P, P, P, L, I, P, P, P, P, P, P],
    "tscd_add_layernorm_bwd_acc_f16": [P, P, P, P, I, I, F, P, P, P, P],
    "tscd_row_normalize_f16": [P, I, I, I, I, F, P, I, P],
    "tscd_conv2d_jobs_nhwc_f16": [P, I, P],
    "tscd_match_cost_pairs_f16": [P, I, I, P, I, P, P],
    "tscd_se_fuse_ex_f16": [P, I, P, I, P, P, P, L, I, P, P, P],
    "tscd_add_layernorm_ex_f16": [P, P, P, P, I, I, F, P, P, P],
    "tscd_se_fuse_bwd_ex_f16": [P, I, P, I, P, P, P, L, I, P, P, P, P, P, P, P],
    "tscd_add_layernorm_bwd_ex_f16": [P, P, P, P, I, I, F, P, P, P, P, P],
    "tscd_batched_nms": [P, I, P, I, P, P, P, I, F, F, I, I, I, I, P, L, P, P, P, P],
    "tscd_conv_wgrad_nhwc_f16": [P, P, P, I, I, I, I, I, I, I, I, P, L, P],
    "tscd_bn_silu_bwd_f16": [P, P, P, P, P, P, L, I, P, P, P, L, P],
    "tscd_bias_relu_bwd_f16": [P, I, P, I, P, I, L, I, P, P, L, P],
    "tscd_linear_bwd_f16": [P, I, P, I, P, I, I, I, I, I, P, P, P, P, L, P],
    "tscd_linear_bwd_packed_f16": [P, I, P, I, P, I, P, I, I, I, I, P, P, P, P, L, P],
    "tscd_simota_assign": [P, P, I, I, I, I, I, P, P, P, I, I, P, L, P, P, P, P, P, P],
    "tscd_simota_assign_r": [P, P, I, I, I, I, I, P, P, P, I, I, F, P, L, P, P, P, P, P, P],
    "tscd_loss_dense_f16": [P, P, P, P, P, P, I, I, I, I, P, L, P, P, P],
    "tscd_loss_combine_f32": [P, P, P, P, P, I, I, P, P, P],
    "tscd_scale_grad_f16": [P, P, I, I, I, L, I, I, P, P],
    "tscd_stream_host_func": [P, P, P],
    "tscd_cls_score_gather_f16": [P, I, L, I, P, P, P, I, I, L, P, P],
    "tscd_cls_score_scatter_f16": [P, P, I, L, I, P, P, P, I, I, L, P, P],
    "tscd_pred_grad_split_f16": [P, L, I, L, I, I, P, P, P, P, L, P],
    "tscd_loss_dense_reg_f16": [P, P, P, P, P, P, I, I, I, I, P, P, P, L, P, P, P, P],
    "tscd_loss_rows_f16": [P, P, I, I, I, I, P, P, P, P, P, P, P, P, I, I, P, L, P, P, P, P, P],
}
PLAIN = {"tscd_conv2d_jobs_max": ([], I), "tscd_conv2d_jobs_supported": ([I, I], I), "tscd_wgrad_batch_max_jobs": ([], I), "tscd_wgrad_batch_workspace_bytes": ([P, I], L), "tscd_wgrad_batch_supported": ([I] * 9, I), "tscd_pred_grad_split_workspace_bytes": ([L, I], L), "tscd_ema_chunk_elems": ([], I), "tscd_pack_block_elems": ([], I), "tscd_bias_relu_bwd_workspace_bytes": ([L, I], L), "tscd_conv_wgrad_splits": ([L, I, I, I], I), "tscd_conv_wgrad_workspace_bytes": ([L, I, I, I], L),
         "tscd_bn_silu_bwd_workspace_bytes": ([L, I], L), "tscd_linear_bwd_workspace_bytes": ([I, I, I], L),
         "tscd_mca_dp": ([I], I), "tscd_mca_workspace_bytes": ([I, I, I, I], L), "tscd_mca_rows_bytes": ([I, I, I], L),
         "tscd_mca_backward_workspace_bytes": ([I, I, I, I, I], L),
         "tscd_nms_workspace_bytes": ([I, I, I, I], L), "tscd_simota_workspace_bytes": ([I, I, I, I], L), "tscd_loss_workspace_bytes": ([L, L, L], L),
         "tscd_conv_stat_tiles": ([L, I, I, I, I, I], I), "tscd_conv_stat_tiles_ex": ([L, I, I, I, I, I, I, I, I], I),
         "tscd_conv_wgrad_workspace_bytes_ex": ([L, I, I, I, I], L), "tscd_cos_attn_mfma_supported": ([I], I), "tscd_conv_tile_n": ([I], I), "tscd_conv_tile_m": ([], I), "tscd_abi_version": ([], I),
         "tscd_last_error": ([], C.c_char_p)}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m tscd_amd.build` "
            "(or __graft_entry__.build()). There is no CPU / PyTorch fallback for the TSCD kernels.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = I
    for name, (argtypes, restype) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


lib = _load()


def check(status, name):
    if status != 0:
        raise RuntimeError(f"{name} failed (status {status}): {lib.tscd_last_error().decode()}")


_FN = {}


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib, name)
    st = fn(*args)
    if st != 0:
        check(st, name)
