"""ctypes binding of libgd4d.so (C ABI declared in include/gd4d.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C graph-detr4d_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GD4D_LIB_PATH') or os.path.join(_HERE, 'libgd4d.so')   # env override: dev A/B builds
ABI_VERSION = 55
PIXEL_MAJOR, HEAD_MAJOR = 0, 1

F32, BF16 = 0, 1

_c = ctypes
_vp, _i, _f = _c.c_void_p, _c.c_int, _c.c_float

# name -> (restype, argtypes); must list every symbol include/gd4d.h declares
SIGNATURES = {
    'gd4d_abi_version': (_i, []),
    'gd4d_error_string': (_c.c_char_p, [_i]),
    'gd4d_last_hip_error': (_c.c_char_p, []),
    'gd4d_trace_enable': (_i, [_vp]),
    'gd4d_cross_attn_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp,
                                 _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'gd4d_pyramid_channels_last_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'gd4d_cross_attn_agg_fwd': (_i, [_vp] * 8 + [_f, _f] + [_vp] * 4 + [_i] * 9 + [_vp] * 5),
    'gd4d_cross_attn_plan_bytes': (_c.c_size_t, [_i] * 5),
    'gd4d_cross_attn_plan_fwd': (_i, [_vp] * 6 + [_f, _f, _vp, _vp, _c.c_int64, _vp, _c.c_size_t, _vp, _vp, _vp] + [_i] * 7 + [_vp, _vp]),
    'gd4d_cross_attn_agg_sliced_fwd': (_i, [_vp, _c.c_int64, _vp, _vp] + [_i] * 8 + [_vp, _i, _i, _vp]),
    'gd4d_cross_attn_agg_items_fwd': (_i, [_vp, _vp, _vp, _c.c_int64, _c.c_int64, _vp, _vp, _vp] + [_i] * 8 + [_vp, _i, _i, _vp]),
    'gd4d_cross_attn_agg_items_coarse_fwd': (_i, [_vp, _vp, _vp, _c.c_int64, _c.c_int64, _vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 8 + [_vp, _vp]),
    'gd4d_chain_guest_bytes': (_c.c_size_t, []),
    'gd4d_value_proj_image_bytes': (_c.c_size_t, []),
    'gd4d_value_proj_image': (_i, [_vp, _vp, _vp, _vp]),
    'gd4d_value_proj_guest_fwd': (_i, [_vp, _i, _vp]),
    'gd4d_row_chain_guest_fwd': (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    'gd4d_row_chain_fill_fwd': (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'gd4d_cross_attn_agg_items_count_fwd': (_i, [_vp, _vp, _vp, _c.c_int64, _c.c_int64, _vp, _vp, _vp] + [_i] * 8 + [_vp, _vp, _vp, _vp, _c.c_size_t, _vp]),
    'gd4d_pyramid_slice_planar_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'gd4d_value_proj_heads_fwd': (_i, [_vp] * 5 + [_i, _i, _i, _vp]),
    'gd4d_value_proj_heads_bwd': (_i, [_vp] * 5 + [_i, _i, _i, _vp]),
    'gd4d_value_proj_heads_bwd_weight_workspace_bytes': (_c.c_size_t, []),
    'gd4d_value_proj_heads_bwd_weight': (_i, [_vp] * 6 + [_c.c_size_t, _i, _i, _i, _i, _vp]),
    'gd4d_value_proj_heads_bwd_weight_group': (_i, [_vp] * 6 + [_i, _vp, _c.c_size_t, _i, _i, _i, _vp]),
    'gd4d_cross_attn_dot_bytes': (_c.c_size_t, [_i] * 5),
    'gd4d_cross_attn_dot_sliced': (_i, [_vp, _c.c_int64, _vp, _vp, _vp, _c.c_size_t] + [_i] * 8 + [_vp, _vp]),
    'gd4d_cross_attn_dot_sliced_wgrad': (_i, [_vp, _c.c_int64, _vp, _vp, _vp, _c.c_size_t] + [_i] * 8 + [_vp] + [_vp] * 5 + [_i, _i, _vp]),
    'gd4d_cross_attn_plan_bwd': (_i, [_vp] * 6 + [_f, _f] + [_vp] * 9 + [_c.c_size_t, _vp] + [_i] * 7 + [_vp, _vp]),
    'gd4d_pyramid_grad_chunks': (_c.c_int64, [_vp, _i, _i]),
    'gd4d_pyramid_grad_slots_bytes': (_c.c_size_t, [_i] * 5),
    'gd4d_pyramid_grad_count': (_i, [_vp, _vp, _vp, _c.c_int64, _vp, _vp, _c.c_size_t] + [_i] * 6 + [_vp]),
    'gd4d_pyramid_grad_scan_workspace_bytes': (_c.c_size_t, [_c.c_int64]),
    'gd4d_pyramid_grad_scan': (_i, [_vp, _vp, _vp, _c.c_size_t, _c.c_int64, _vp]),
    'gd4d_pyramid_grad_fill': (_i, [_vp, _vp, _vp, _vp, _c.c_uint32, _vp] + [_i] * 5 + [_vp]),
    'gd4d_pyramid_grad_chunk_geometry': (_i, [_vp, _i, _i, _vp]),
    'gd4d_pyramid_grad_sort': (_i, [_vp] * 5 + [_c.c_int64, _vp]),
    'gd4d_pyramid_grad_reduce': (_i, [_vp] * 7 + [_i, _i, _i, _i, _vp]),
    'gd4d_query_order_fwd': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'gd4d_refine_reference_order_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'gd4d_cross_attn_bwd': (_i, [_vp] * 8 + [_f, _f] + [_vp] * 6 + [_i] * 10 + [_vp, _vp, _c.c_size_t, _vp]),
    'gd4d_cross_attn_bwd_workspace_bytes': (_c.c_size_t, [_i] * 5),
    'gd4d_detr3d_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp,
                             _i, _i, _i, _i, _i, _i, _vp]),
    'gd4d_detr3d_bwd': (_i, [_vp] * 6 + [_f, _f] + [_vp] * 4 + [_i] * 6 + [_vp]),
    'gd4d_detr3d_v2_fwd': (_i, [_vp] * 7 + [_f, _f, _vp, _vp] + [_i] * 7 + [_vp]),
    'gd4d_detr3d_v2_bwd': (_i, [_vp] * 7 + [_f, _f] + [_vp] * 5 + [_i] * 6 + [_vp]),
    'gd4d_value_proj_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _c.c_size_t, _i, _vp]),
    'gd4d_value_proj_workspace_bytes': (_c.c_size_t, [_i]),
    'gd4d_linear_fwd': (_i, [_vp] * 7 + [_i] * 9 + [_vp, _vp]),
    'gd4d_linear_group_fwd': (_i, [_vp] * 6 + [_i] * 4 + [_vp, _vp]),
    'gd4d_linear_ln_fwd': (_i, [_vp] * 9 + [_i] * 5 + [_f] + [_i] * 4 + [_vp]),
    'gd4d_small_linear_layernorm_fwd': (_i, [_vp] * 6 + [_i, _i, _i, _f, _i, _vp]),
    'gd4d_layernorm_fwd': (_i, [_vp] * 5 + [_i, _i, _f, _i, _vp]),
    'gd4d_mha_core_fwd': (_i, [_vp] * 5 + [_i] * 10 + [_f, _vp, _f, _vp, _vp]),
    'gd4d_mha_core_bwd': (_i, [_vp] * 11 + [_i] * 14 + [_f, _f, _vp, _vp]),
    'gd4d_mha_core_bwd_fill': (_i, [_vp] * 11 + [_i] * 14 + [_f, _f, _vp] + [_vp, _i, _vp, _vp] + [_i] * 4 + [_vp]),
    'gd4d_layernorm_bwd_workspace_bytes': (_c.c_size_t, [_i, _i]),
    'gd4d_layernorm_bwd': (_i, [_vp] * 9 + [_c.c_size_t, _i, _i, _f, _i, _vp]),
    'gd4d_inverse_sigmoid_fwd': (_i, [_vp, _vp, _c.c_int64, _vp]),
    'gd4d_inverse_sigmoid_bwd': (_i, [_vp, _vp, _vp, _vp, ctypes.c_int64, _vp]),
    'gd4d_layernorm_bwd_reduce_group': (_i, [_vp] * 4 + [_i, _i, _vp]),
    'gd4d_refine_reference_fwd': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'gd4d_frustum_pe_input_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _i, _i, _vp]),
    'gd4d_se_fuse_chlast_fwd': (_i, [_vp] * 5 + [_i] * 7 + [_vp]),
    'gd4d_sine_pe3d_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'gd4d_se_fuse_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _c.c_size_t, _vp]),
    'gd4d_split_bf16_fwd': (_i, [_vp, _vp, _vp, _c.c_size_t, _vp]),
    'gd4d_gemm_bf16x3_fwd': (_i, [_vp] * 5 + [_i] * 6 + [_vp]),
    'gd4d_image_job_bytes': (ctypes.c_size_t, []),
    'gd4d_chain_weight_image_group': (_i, [_vp, _i, _i, _vp]),
    'gd4d_gemm_tn_bf16x3_workspace_bytes': (ctypes.c_size_t, [ctypes.c_longlong, _i, _i]),
    'gd4d_gemm_tn_bf16x3': (_i, [_vp] * 5 + [ctypes.c_longlong] + [_i] * 5 + [_vp]),
    'gd4d_se_fuse_chlast_bwd': (_i, [_vp] * 6 + [_i] * 5 + [_vp]),
    'gd4d_knn_farthest_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'gd4d_edge_conv_max_fwd': (_i, [_vp] * 6 + [_i] * 5 + [_vp]),
    'gd4d_box_head_fwd': (_i, [_vp, _vp, _vp, _f, _vp, _i, _i, _vp]),
    'gd4d_nms_free_decode_fwd': (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'gd4d_value_proj_multi_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _c.c_size_t, _i, _vp]),
    'gd4d_chain_op_bytes': (_c.c_size_t, []),
    'gd4d_row_chain_fwd': (_i, [_vp, _i, _i, _vp]),
    'gd4d_row_chain2_fwd': (_i, [_vp, _i, _vp, _i, _i, _vp]),
    'gd4d_mha_core_presplit_fwd': (_i, [_vp] * 4 + [_i] * 5 + [_c.c_longlong, _c.c_longlong, _vp, _i, _f, _vp, _f, _vp, _vp]),
    'gd4d_xcd_placement_probe': (_i, [_vp, _i, _vp]),
    'gd4d_mlp2_image_bytes': (_c.c_size_t, [_i] * 3),
    'gd4d_mlp2_image': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'gd4d_mlp2_bf16x3_fwd': (_i, [_vp] * 4 + [_i] * 6 + [_vp]),
    'gd4d_mlp2_se_fuse_fwd': (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'gd4d_mlp2_frustum_fwd': (_i, [_vp, _vp, _i, _i, _c.c_float, _c.c_float, _i, _c.c_float, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'gd4d_mlp2_pe_se_fwd': (_i, [_vp, _vp, _vp, _i, _i, _c.c_float, _c.c_float, _i, _c.c_float, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp,
                                  _vp, _vp]),
    'gd4d_adamw_flat_workspace_bytes': (_c.c_size_t, []),
    'gd4d_adamw_flat': (_i, [_vp] * 6 + [_c.c_size_t, _c.c_int64] + [_f] * 6 + [_vp]),
    'gd4d_chain_weight_image_bytes': (_c.c_size_t, [_i, _i]),
    'gd4d_chain_weight_image': (_i, [_vp, _i, _i, _vp, _vp]),
    'gd4d_chain_weight_image_exact_bytes': (_c.c_size_t, [_i, _i]),
    'gd4d_chain_weight_image_exact': (_i, [_vp, _i, _i, _vp, _vp]),
    'gd4d_linear_sum_assignment_batch': (_i, [_vp] * 4 + [_i, _vp, _vp, _i]),
    'gd4d_hungarian_assign_workspace_bytes': (_c.c_size_t, [_i] * 4),
    'gd4d_hungarian_assign_fwd': (_i, [_vp] * 5 + [_c.c_size_t] + [_i] * 5 + [_vp]),
    'gd4d_match_cost_fwd': (_i, [_vp] * 6 + [_i] * 8 + [_f] * 3 + [_vp]),
    'gd4d_head_loss_fwd_bwd': (_i, [_vp] * 10 + [_i] * 7 + [_f] * 3 + [_vp]),
    'gd4d_linear_bwd_weight': (_i, [_vp] * 4 + [_i] * 6 + [_vp]),
    'gd4d_linear_bwd_weight_group': (_i, [_vp] * 5 + [_i, _i, _vp]),
    'gd4d_value_proj_bwd_input': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'gd4d_value_proj_bwd_weight_workspace_bytes': (_c.c_size_t, []),
    'gd4d_value_proj_bwd_weight': (_i, [_vp] * 6 + [_c.c_size_t, _i, _i, _i, _vp]),
}

_lib = None


class Gd4dError(RuntimeError):
    pass


def load():
    """dlopen libgd4d.so once; raises Gd4dError if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Gd4dError(f'{LIB_PATH} not found: the HIP extension is not built '
                        '(run __graft_entry__.build()); there is no CPU fallback')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.gd4d_abi_version() != ABI_VERSION:
        raise Gd4dError(f'libgd4d.so ABI {lib.gd4d_abi_version()} != expected {ABI_VERSION}')
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        lib = load()
        msg = lib.gd4d_error_string(code).decode()
        hip = lib.gd4d_last_hip_error().decode()
        raise Gd4dError(f'{what} failed: {msg} (code {code})' + (f'; HIP: {hip}' if hip else ''))
