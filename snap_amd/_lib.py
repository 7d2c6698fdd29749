"""ctypes binding of libsnap_hip.so (the C ABI declared in include/snap_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent the
import of :mod:`snap_amd.ops` raises.  Build it with ``make -C snap_amd/csrc`` or
``python -c "import __graft_entry__ as g; g.build()"``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNAP_HIP_LIB: alternative build of the same ABI (A/B experiments); default = in-tree.
LIB_PATH = os.environ.get('SNAP_HIP_LIB') or os.path.join(_HERE, 'lib', 'libsnap_hip.so')

c_int = ctypes.c_int32
c_i64 = ctypes.c_int64
c_u64 = ctypes.c_uint64
c_float = ctypes.c_float
c_size = ctypes.c_size_t
ptr = ctypes.c_void_p


class SnapConvDesc(ctypes.Structure):
  _fields_ = [
      ('N', c_int), ('H', c_int), ('W', c_int), ('Cin', c_int),
      ('Cin_stride', c_int),
      ('KH', c_int), ('KW', c_int), ('stride', c_int), ('pad_t', c_int),
      ('pad_l', c_int),
      ('Ho', c_int), ('Wo', c_int), ('Cout', c_int), ('Cout_stride', c_int),
      ('prologue', c_int), ('epilogue', c_int),
      ('in_scale', c_float), ('in_shift', c_float), ('tile_hint', c_int),
  ]


class SnapConvExtras(ctypes.Structure):
  _fields_ = [
      ('rows_in', ptr), ('rows_out', ptr), ('row_count', ptr), ('gn_partial', ptr),
      ('gn_partial_bytes', c_size), ('gn_partial_relu', c_int),
      ('workspace', ptr), ('workspace_bytes', c_size),
      ('w_bf16', ptr), ('w_bf16_bytes', c_size), ('w_split_parts', c_int), ('w_split_root', c_int),
      ('gn_partial2', ptr), ('gn_partial2_bytes', c_size), ('gn_partial2_done', c_int),
      ('x_presplit', c_int), ('ps_tile', c_int), ('ps_res_init', c_int),
      ('bk_hint', c_int), ('tune_flags', c_int), ('gn_partial_rows', c_int), ('w_half', c_int), ('x_half', c_int), ('y_half', ptr),
      ('gnb_x', ptr), ('gnb_mu', ptr), ('gnb_rstd', ptr), ('gnb_gamma', ptr), ('gnb_beta', ptr), ('gnb_mode', c_int),
  ]


class SnapLiftDesc(ctypes.Structure):
  _fields_ = [
      ('B', c_int), ('V', c_int), ('h', c_int), ('w', c_int), ('C', c_int),
      ('feature_dim', c_int), ('num_bins', c_int), ('N', c_int), ('K', c_int),
      ('fisheye', c_int), ('out_stride', c_int),
      ('depth_min', c_float), ('depth_max', c_float),
      ('max_view_distance', c_float),
      ('weighted', c_int), ('use_variance', c_int), ('add_minmax', c_int),
      ('grid_y', c_int), ('grid_z', c_int), ('valid_rows_only', c_int), ('out_split', c_int),
      ('class_rows', c_int), ('tune_flags', c_int),
  ]


# name -> (restype, argtypes): every symbol include/snap_hip.h declares.
SIGNATURES = {
    'snap_abi_version': (c_int, []),
    'snap_status_string': (ctypes.c_char_p, [c_int]),
    'snap_build_arch': (ctypes.c_char_p, []),
    'snap_conv2d_nhwc_f32': (
        c_int,
        [ctypes.POINTER(SnapConvDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr,
         ptr, ptr, ptr],
    ),
    'snap_conv2d_nhwc_ex_f32': (
        c_int,
        [ctypes.POINTER(SnapConvDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr,
         ptr, ptr, ctypes.POINTER(SnapConvExtras), ptr],
    ),
    'snap_conv2d_gn_partial_bytes': (c_size, [ctypes.POINTER(SnapConvDesc)]),
    'snap_conv2d_workspace_bytes': (c_size, [ctypes.POINTER(SnapConvDesc)]),
    'snap_conv2d_tile_rows': (c_int, [ctypes.POINTER(SnapConvDesc)]),
    'snap_conv2d_pack_weights_blocks': (c_int, [c_int, c_int, c_int]),
    'snap_conv2d_pack_weights_multi_bf16': (c_int, [ptr, c_int, c_int, ptr]),
    'snap_conv2d_pack_weights_multi_f16': (c_int, [ptr, c_int, c_int, ptr]),
    'snap_conv2d_stationary_kind': (c_int, [ctypes.POINTER(SnapConvDesc), c_int]),
    'snap_conv2d_tile_rows_ex': (c_int, [ctypes.POINTER(SnapConvDesc), c_int]),
    'snap_conv2d_gn_partial_bytes_ex': (c_size, [ctypes.POINTER(SnapConvDesc), c_int]),
    'snap_conv2d_splitk_gn_partial_bytes': (c_size, [ctypes.POINTER(SnapConvDesc)]),
    'snap_conv2d_presplit_tile_rows': (c_int, [ctypes.POINTER(SnapConvDesc), c_int]),
    'snap_conv2d_presplit_gn_partial_bytes': (c_size, [ctypes.POINTER(SnapConvDesc), c_int]),
    'snap_conv2d_presplit_workspace_bytes': (c_size, [ctypes.POINTER(SnapConvDesc), c_int]),
    'snap_conv2d_presplit_supported': (c_int, [ctypes.POINTER(SnapConvDesc)]),
    'snap_gn_norm_split_f32': (
        c_int, [ptr, ptr, c_int, c_int, c_int, c_int, c_float, c_int, ptr, ptr, ptr, ptr, ptr, ptr]),
    'snap_presplit_f32': (c_int, [ptr, c_i64, c_int, ptr, ptr]),
    'snap_interpolate_nd_f32': (c_int, [ptr, ptr, c_int, c_int, ptr, ptr, c_i64, ptr, ptr, ptr]),
    'snap_expectation_nd_f32': (c_int, [ptr, c_i64, ptr, c_int, ptr, ptr]),
    'snap_semantic_embed_f32': (
        c_int, [ptr, c_i64, c_int, ptr, c_int, ptr, c_int, ptr, ptr, c_int, ptr, ptr]),
    'snap_semantic_onehot_f32': (c_int, [ptr, c_i64, c_int, ptr, c_int, ptr, c_int, ptr, c_int, ptr]),
    'snap_voting_fft_workspace_bytes': (c_size, [c_int] * 6),
    'snap_voting_fft_f32': (
        c_int, [ptr, ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, ptr, c_size,
                ptr, ptr]),
    'snap_voting_fft_rotated_f32': (
        c_int, [ptr, ptr, ptr, c_float, ptr, ptr, c_int, c_int, c_int, c_int, c_int, c_float, ptr, c_size, ptr, ptr]),
    'snap_stack_templates_f32': (c_int, [ptr, ptr, c_int, c_int, c_int, c_int, c_int, ptr]),
    'snap_stack_templates_rhwd_f32': (c_int, [ptr, ptr, c_int, c_int, c_int, c_int, c_int, ptr]),
    'snap_pack_stacked_templates_split_bf16': (c_int, [ptr, c_int, c_int, c_int, c_int, c_int, ptr, c_size, ptr]),
    'snap_layer_norm_f32': (c_int, [ptr, ptr, ptr, ptr, c_i64, c_int, c_float, ptr]),
    'snap_attention_bf16_f32': (c_int, [ptr, ptr, c_int, c_int, c_int, c_int, c_float, ptr]),
    'snap_layer_norm_bf16out_f32': (c_int, [ptr, ptr, ptr, ptr, c_i64, c_int, c_float, ptr]),
    'snap_attention_bf16out_f32': (c_int, [ptr, ptr, c_int, c_int, c_int, c_int, c_float, ptr]),
    'snap_attention_bf16io': (c_int, [ptr, ptr, c_int, c_int, c_int, c_int, c_float, ptr]),
    'snap_attention_lse_bf16_f32': (c_int, [ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_float, ptr]),
    'snap_attention_bwd_bf16_f32': (
        c_int, [ptr, ptr, ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_float, ptr]),
    'snap_layer_norm_bwd_workspace_bytes': (c_size, [c_i64, c_int]),
    'snap_layer_norm_bwd_f32': (
        c_int, [ptr, ptr, ptr, ptr, ptr, ptr, c_i64, c_int, c_float, ptr, c_size, ptr]),
    'snap_gelu_f32': (c_int, [ptr, ptr, c_i64, ptr]),
    'snap_gelu_bwd_f32': (c_int, [ptr, ptr, ptr, c_i64, ptr]),
    'snap_epilogue_bwd_colsum_half': (c_int, [ptr, ptr, ptr, c_i64, c_int, c_int, ptr, ptr, ptr, c_size, c_int, ptr]),
    'snap_epilogue_bwd_colsum_wsum_half': (
        c_int, [ptr, ptr, ptr, c_i64, c_int, c_int, ptr, ptr, ptr, c_size, c_int, ptr, ptr, c_i64, c_int, ptr, ptr]),
    'snap_epilogue_bwd_colsum_wsum_tail_half': (
        c_int, [ptr, ptr, ptr, c_i64, c_int, c_int, ptr, ptr, ptr, c_size, c_int, ptr, ptr, c_i64, c_int, ptr,
                ptr, ptr, c_i64, ptr]),
    'snap_conv2d_wgrad_half_f32': (
        c_int,
        [ctypes.POINTER(SnapConvDesc), ptr, ptr, ptr, ptr, ptr, ptr, c_int, ptr, c_size, ptr, ptr, ptr, c_int,
         c_int, c_int, ptr],
    ),
    'snap_adam_multi_blocks': (c_i64, [c_i64]),
    'snap_adam_multi_f32': (c_int, [ptr, c_int, c_i64, c_float, c_float, c_float, c_float, c_int, ptr, ptr]),
    'snap_conv2d_packed_weights_bytes': (c_size, [c_int, c_int, c_int]),
    'snap_conv2d_pack_weights_bf16': (c_int, [ptr, c_int, c_int, c_int, ptr, c_size, ptr]),
    'snap_conv2d_pack_weights_f16': (c_int, [ptr, c_int, c_int, c_int, ptr, c_size, ptr]),
    'snap_conv2d_packed_weights_split_bytes': (c_size, [c_int, c_int, c_int, c_int]),
    'snap_conv2d_packed_weights_split_root_bytes': (c_size, [c_int, c_int]),
    'snap_conv2d_pack_weights_split_root_bf16': (c_int, [ptr, c_int, c_int, ptr, c_size, ptr]),
    'snap_conv2d_pack_weights_split_bf16': (c_int, [ptr, c_int, c_int, c_int, c_int, ptr, c_size, ptr]),
    'snap_conv2d_pack_weights_split_blocks': (c_int, [c_int, c_int, c_int]),
    'snap_conv2d_pack_weights_split_multi_bf16': (c_int, [ptr, c_int, c_int, c_int, ptr]),
    'snap_group_norm_stats_from_partial_f32': (
        c_int, [ptr, c_int, c_int, c_int, c_int, c_float, c_int, ptr, ptr, ptr, ptr, ptr]
    ),
    'snap_compact_rows_workspace_bytes': (c_size, [c_i64]),
    'snap_compact_rows_u8': (c_int, [ptr, c_i64, ptr, ptr, ptr, c_size, ptr]),
    'snap_compact_rows_range_u8': (c_int, [ptr, c_i64, c_int, c_int, ptr, ptr, ptr, c_size, ptr]),
    'snap_mlp2_pool_max_classes_f32': (c_int, [ptr, c_i64, c_int, c_int, ptr, ptr, ptr, ptr, c_int, c_int,
                                               ptr, c_size, ptr, c_int, ptr, c_size, ptr, c_int, c_int,
                                               c_int, c_int, c_i64, ptr, ptr, ptr]),
    'snap_mlp2_pool_max_gather_f32': (c_int, [ptr, c_i64, c_int, c_int, ptr, ptr, ptr, ptr, ptr, c_i64, c_int,
                                              c_int, c_int, ptr, c_int, ptr, c_size, ptr, c_int, ptr, c_size,
                                              ptr, c_int, c_int, c_i64, ptr, ptr, ptr]),
    'snap_mlp2_pool_max_f32': (c_int, [ptr, c_i64, c_int, c_int, ptr, ptr, ptr, c_size, ptr, c_int,
                                       ptr, c_size, ptr, c_int, c_int, c_int, c_int, c_i64, ptr, ptr, ptr]),
    'snap_pad_image_f32': (c_int, [ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_int, ptr, ptr]),
    'snap_voxel_points_f32': (c_int, [ptr, c_int, ptr, c_int, c_int, c_int, ptr, ptr]),
    'snap_fill_masked_rows_f32': (c_int, [ptr, ptr, c_i64, c_int, c_float, ptr]),
    'snap_weight_standardize_f32': (c_int, [ptr, ptr, c_int, c_int, c_float, ptr]),
    'snap_weight_standardize_multi_f32': (c_int, [ptr, c_int, c_int, c_float, ptr]),
    'snap_weight_standardize_bwd_multi_f32': (c_int, [ptr, c_int, c_int, c_float, ptr]),
    'snap_group_norm_stats_workspace_bytes': (c_size, [c_int, c_int, c_int, c_int]),
    'snap_group_norm_stats_f32': (
        c_int,
        [ptr, c_int, c_int, c_int, c_int, c_int, c_float, c_int, ptr, ptr, ptr,
         ptr, ptr, c_size, ptr],
    ),
    'snap_group_norm_apply_f32': (
        c_int, [ptr, ptr, c_int, c_int, c_int, ptr, ptr, ptr, c_int, ptr]
    ),
    'snap_max_pool_3x3s2_f32': (c_int, [ptr, ptr, c_int, c_int, c_int, c_int, ptr]),
    'snap_lift_pool_f32': (
        c_int, [ctypes.POINTER(SnapLiftDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr]
    ),
    'snap_lift_pool_records_f32': (
        c_int, [ctypes.POINTER(SnapLiftDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr]
    ),
    'snap_lift_observations_f32': (c_int, [ctypes.POINTER(SnapLiftDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr]),
    'snap_lift_pool_observations_f32': (c_int, [ctypes.POINTER(SnapLiftDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr]),
    'snap_project_points_f32': (
        c_int, [c_int, c_int, c_int, c_int, ptr, ptr, ptr, ptr, ptr, ptr, ptr]
    ),
    'snap_vertical_pool_f32': (
        c_int, [ptr, ptr, ptr, ptr, c_i64, c_int, c_int, c_int, ptr]
    ),
    'snap_vertical_pool_max_arg_f32': (
        c_int, [ptr, ptr, ptr, ptr, ptr, ptr, c_i64, c_int, c_int, ptr]
    ),
    'snap_plane_fuse_match_f32': (
        c_int,
        [ptr, ptr, c_int, c_i64, c_int, c_int, ptr, ptr, ptr, ptr, c_int, c_int,
         c_float, ptr, ptr],
    ),
    'snap_sim_rowstats_bytes': (c_size, [c_int, c_int]),
    'snap_sim_softmax_f32': (
        c_int,
        [ptr, ptr, c_int, c_int, c_int, c_int, c_float, c_int, ptr, ptr, ptr,
         ptr, ptr, ptr],
    ),
    'snap_sim_softmax_weighted_f32': (
        c_int,
        [ptr, ptr, c_int, c_int, c_int, c_int, c_float, c_int, ptr, ptr, ptr, ptr,
         ptr, ptr, ptr],
    ),
    'snap_sim_split_workspace_bytes': (c_size, [c_int, c_int, c_int, c_int, c_int]),
    'snap_sim_softmax_split_f32': (c_int, [ptr, ptr, c_int, c_int, c_int, c_int, c_float, c_int, ptr, ptr,
                                           c_int, ptr, ptr, ptr, c_size, ptr]),
    'snap_masked_softmax_rows_f32': (c_int, [ptr, ptr, c_int, c_int, ptr, ptr, ptr]),
    'snap_confidence_head_f32': (c_int, [ptr, ptr, ptr, ptr, c_i64, c_int, ptr, ptr]),
    'snap_confidence_head_bwd_f32': (c_int, [ptr, ptr, ptr, ptr, ptr, c_i64, c_int, ptr, ptr, ptr, ptr]),
    'snap_sim_bwd_prepare_rows_f32': (c_int, [ptr, ptr, c_int, c_int, c_int, c_int, ptr, ptr, ptr]),
    'snap_masked_softmax_rows_bwd_f32': (c_int, [ptr, ptr, c_int, c_int, ptr, ptr]),
    'snap_ransac_sample_sim_f32': (
        c_int,
        [ptr, ptr, ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
         c_int, c_u64, ptr, ptr, ptr, c_size, ptr],
    ),
    'snap_ransac_sample_rows_f32': (
        c_int,
        [ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
         c_int, c_u64, ptr, ptr, ptr, c_size, ptr],
    ),
    'snap_ransac_sample_f32': (
        c_int,
        [ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
         c_int, c_u64, ptr, ptr, ptr],
    ),
    'snap_ransac_sample_workspace_bytes': (c_size, [c_int, c_int]),
    'snap_ransac_sample_ws_f32': (
        c_int,
        [ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
         c_int, c_u64, ptr, ptr, ptr, c_size, ptr],
    ),
    'snap_poses_from_corr_f32': (
        c_int, [ptr, ptr, c_int, c_int, c_int, c_int, c_float, ptr, ptr]
    ),
    'snap_pose_score_window_workspace_bytes': (c_size, [c_int, c_int, c_int, c_int, c_int]),
    'snap_pose_score_window_supported': (c_int, [c_int, c_int, c_int]),
    'snap_pose_score_window_f32': (c_int, [ptr, ptr, ptr, c_int, ptr, ptr, c_int, c_int, c_int, c_int, c_int,
                                           c_float, ptr, ptr, c_size, ptr]),
    'snap_pose_score_workspace_bytes': (c_size, [c_int, c_int, c_int, c_int, c_int]),
    'snap_pose_score_f32': (
        c_int,
        [ptr, ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_int, c_float,
         c_int, ptr, ptr, c_size, ptr],
    ),
    'snap_refine_lattice_f32': (c_int, [ptr, ptr, ptr, c_int, c_int, c_int, ptr, ptr]),
    'snap_argmax_rows_f32': (c_int, [ptr, c_int, c_int, c_int, ptr, ptr]),
    'snap_rotate_templates_f32': (
        c_int,
        [ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_float, ptr, ptr, ptr, ptr,
         ptr, ptr],
    ),
    'snap_pad_map_f32': (c_int, [ptr, ptr, c_int, c_int, c_int, ptr, ptr, ptr]),
    'snap_template_finalize_f32': (
        c_int, [ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_float, c_int, ptr, ptr]
    ),
    'snap_vertical_pool_conf_f32': (
        c_int, [ptr, ptr, ptr, ptr, c_i64, c_int, c_int, c_int, ptr, ptr, ptr, ptr, ptr]
    ),
    'snap_vertical_pool_conf_bwd_partial_rows': (c_size, [c_i64]),
    'snap_vertical_pool_conf_bwd_f32': (
        c_int, [ptr, ptr, ptr, ptr, ptr, ptr, c_i64, c_int, c_int, c_int, ptr, ptr, ptr]
    ),
    # ---- training path ----
    'snap_conv2d_wgrad_workspace_bytes': (c_size, [ctypes.POINTER(SnapConvDesc)]),
    'snap_conv2d_wgrad_f32': (
        c_int,
        [ctypes.POINTER(SnapConvDesc), ptr, ptr, ptr, ptr, ptr, ptr, c_int, ptr, c_size, ptr],
    ),
    'snap_conv2d_wgrad_rows_f32': (
        c_int,
        [ctypes.POINTER(SnapConvDesc), ptr, ptr, ptr, ptr, ptr, ptr, c_int, ptr, c_size, ptr, ptr,
         ptr, ptr],
    ),
    'snap_conv2d_wgrad_ex_f32': (
        c_int,
        [ctypes.POINTER(SnapConvDesc), ptr, ptr, ptr, ptr, ptr, ptr, c_int, ptr, c_size, ptr, ptr,
         ptr, c_int, ptr],
    ),
    'snap_colsum_rows_f32': (c_int, [ptr, c_i64, c_int, ptr, ptr, ptr, c_int, ptr, c_size, ptr]),
    'snap_group_norm_bwd_workspace_bytes': (c_size, [c_int, c_int, c_int, c_int]),
    'snap_group_norm_bwd_f32': (
        c_int,
        [ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, ptr, ptr, ptr, ptr, c_int, ptr, ptr,
         c_int, ptr, c_size, ptr],
    ),
    'snap_group_norm_bwd_ex_f32': (
        c_int,
        [ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, ptr, ptr, ptr, ptr, c_int, ptr, ptr,
         c_int, ptr, c_size, ptr, c_int, ptr],
    ),
    'snap_group_norm_bwd_stats_f32': (
        c_int,
        [ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, ptr, ptr, ptr, ptr, c_int, ptr, ptr,
         c_int, ptr, c_size, ptr, c_int, ptr, c_int, ptr],
    ),
    'snap_weight_standardize_bwd_f32': (c_int, [ptr, ptr, ptr, c_int, c_int, c_float, ptr]),
    'snap_max_pool_3x3s2_bwd_f32': (c_int, [ptr, ptr, ptr, c_int, c_int, c_int, c_int, ptr]),
    'snap_upsample2x_bwd_f32': (c_int, [ptr, ptr, c_int, c_int, c_int, c_int, ptr]),
    'snap_epilogue_bwd_f32': (c_int, [ptr, ptr, ptr, ptr, c_i64, c_int, c_int, ptr]),
    'snap_epilogue_bwd_colsum_f32': (c_int, [ptr, ptr, ptr, ptr, c_i64, c_int, c_int, ptr, ptr, ptr, c_size, ptr]),
    'snap_colsum_workspace_bytes': (c_size, [c_i64, c_int]),
    'snap_colsum_f32': (c_int, [ptr, c_i64, c_int, ptr, c_int, ptr, c_size, ptr]),
    'snap_lift_pool_bwd_f32': (
        c_int, [ctypes.POINTER(SnapLiftDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr]
    ),
    'snap_lift_pool_bwd_det_workspace_bytes': (c_size, [ctypes.POINTER(SnapLiftDesc)]),
    'snap_lift_pool_observations_bwd_f32': (
        c_int, [ctypes.POINTER(SnapLiftDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr]),
    'snap_lift_observations_bwd_workspace_bytes': (c_size, [ctypes.POINTER(SnapLiftDesc)]),
    'snap_lift_observations_bwd_f32': (
        c_int, [ctypes.POINTER(SnapLiftDesc), ptr, ptr, ptr, ptr, ptr, ptr, c_size, ptr]),
    'snap_lift_pool_bwd_det_f32': (
        c_int, [ctypes.POINTER(SnapLiftDesc), ptr, ptr, ptr, ptr, ptr, ptr, ptr, c_size, ptr]
    ),
    'snap_vertical_pool_bwd_f32': (
        c_int, [ptr, ptr, ptr, ptr, c_i64, c_int, c_int, c_int, ptr]
    ),
    'snap_vertical_pool_max_bwd_arg_f32': (
        c_int, [ptr, ptr, ptr, ptr, ptr, ptr, c_i64, c_int, c_int, ptr]
    ),
    'snap_plane_fuse_match_bwd_f32': (
        c_int,
        [ptr, ptr, ptr, c_int, c_i64, c_int, c_int, ptr, ptr, c_int, c_int, c_float, ptr, ptr,
         ptr, ptr],
    ),
    'snap_pose_score_bwd_workspace_bytes': (c_size, [c_int, c_int]),
    'snap_pose_score_bwd_f32': (
        c_int,
        [ptr, ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_int, c_float, c_int, ptr, ptr,
         c_size, ptr],
    ),
    'snap_pose_score_bwd_ex_f32': (
        c_int,
        [ptr, ptr, ptr, ptr, ptr, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int, ptr, ptr,
         c_size, ptr],
    ),
    'snap_sim_bwd_prepare_f32': (
        c_int, [ptr, ptr, c_int, c_i64, c_int, ptr, ptr, c_int, ptr]
    ),
}

ABI_VERSION = 22

_lib = None


def load():
  """Load libsnap_hip.so and bind every declared symbol; raises if impossible."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f'{LIB_PATH} not found: the HIP extension is not built. '
        'Run `make -C snap_amd/csrc` (needs hipcc); there is no CPU fallback.'
    )
  lib = ctypes.CDLL(LIB_PATH)
  for name, (restype, argtypes) in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError if the symbol is missing
    fn.restype = restype
    fn.argtypes = argtypes
  if lib.snap_abi_version() != ABI_VERSION:
    raise RuntimeError(
        f'libsnap_hip.so ABI {lib.snap_abi_version()} != expected {ABI_VERSION}; rebuild.'
    )
  _lib = lib
  return lib


def check(status, what):
  if status != 0:
    msg = load().snap_status_string(status).decode()
    raise RuntimeError(f'{what} failed: {msg} (status {status})')
