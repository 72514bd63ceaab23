"""ctypes binding of libgnpde_hip.so (include/gnpde.h).  There is no CPU fallback: if the library
is missing, or a tensor is not on a HIP device, the product path raises."""
import ctypes
import os
import subprocess
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libgnpde_hip.so')
_lib = None

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int32)
c_vp = ctypes.c_void_p

LONG_ROW = 512

STAGE_RHS, STAGE_EULER, STAGE_RK1, STAGE_RK2, STAGE_RK3, STAGE_RK4 = range(6)
STAGE_RK1C, STAGE_RK2C, STAGE_RK3C, STAGE_RK4C = range(6, 10)
STAGE_LINCOMB = 10
ABI_VERSION = 7      # GNPDE_ABI_VERSION of include/gnpde.h this package's struct layouts and prototypes were written for
ATT_SCALED_DOT, ATT_COSINE, ATT_PEARSON, ATT_EXP_KERNEL, ATT_GAT = range(5)
RHS_LAPLACIAN, RHS_TRANSFORMER, RHS_GAT = range(3)
METHOD_EULER, METHOD_RK4, METHOD_MIDPOINT = range(3)
ADAPTIVE_HEUN, ADAPTIVE_DOPRI5 = range(2)
TUNE_SPMM_VARIANT, TUNE_FUSED_BLOCKS_PER_CU, TUNE_ONE_PASS, TUNE_FORK, TUNE_ATT_GENERIC_ROWS, TUNE_RK4_CLASSIC = range(6)
TUNE_ROW_FUSION, TUNE_ONE_PASS_VARIANT, TUNE_LINEAR_STREAMING, TUNE_SPMM_PART, TUNE_XCD_ROWS, TUNE_HUB_FOLD = 6, 7, 8, 9, 10, 11

ATT_TYPES = {'scaled_dot': ATT_SCALED_DOT, 'cosine_sim': ATT_COSINE, 'pearson': ATT_PEARSON,
             'exp_kernel': ATT_EXP_KERNEL}


class GraphStruct(ctypes.Structure):
  _fields_ = [('n', ctypes.c_int32), ('e', ctypes.c_int32),
              ('rowptr', c_vp), ('colidx', c_vp), ('rowidx', c_vp), ('perm', c_vp),
              ('cscptr', c_vp), ('cscpos', c_vp),
              ('n_long_rows', ctypes.c_int32), ('n_long_chunks', ctypes.c_int32),
              ('long_rows', c_vp), ('long_chunk_ptr', c_vp), ('long_chunk_row', c_vp),
              ('long_chunk_begin', c_vp), ('long_chunk_end', c_vp),
              ('n_long_cols', ctypes.c_int32), ('n_bin16', ctypes.c_int32), ('n_bin64', ctypes.c_int32),
              ('max_row_len', ctypes.c_int32), ('max_col_len', ctypes.c_int32), ('row_begin', ctypes.c_int32), ('long_cols', c_vp), ('bin_rows', c_vp), ('long_chunk_first', c_vp),
              ('xcd_deal', ctypes.c_int32), ('n_bin_le64', ctypes.c_int32)]


class EpilogueStruct(ctypes.Structure):
  _fields_ = [('alpha', c_vp), ('beta', c_vp), ('x0', c_vp),
              ('alpha_sigmoid', ctypes.c_int32), ('stage', ctypes.c_int32), ('dt', ctypes.c_float),
              ('y', c_vp), ('k1', c_vp), ('k2', c_vp), ('k3', c_vp), ('out_k', c_vp), ('out_y', c_vp),
              ('n_prev', ctypes.c_int32), ('pad_', ctypes.c_int32), ('prev', c_vp * 7), ('coef', ctypes.c_float * 8),
              ('coef_scale', c_vp)]


class AttentionStruct(ctypes.Structure):
  _fields_ = [('type', ctypes.c_int32), ('heads', ctypes.c_int32), ('att_dim', ctypes.c_int32),
              ('norm_idx', ctypes.c_int32), ('square_plus', ctypes.c_int32), ('leaky_slope', ctypes.c_float),
              ('q', c_vp), ('k', c_vp), ('ldqk', ctypes.c_int32), ('n_key_rows', ctypes.c_int32),
              ('gat_a', c_vp), ('output_var', c_vp), ('lengthscale', c_vp), ('edge_w_csr', c_vp),
              ('graph_t', ctypes.POINTER(GraphStruct)), ('t_from_csr', c_vp)]


class RhsStruct(ctypes.Structure):
  _fields_ = [('kind', ctypes.c_int32), ('graph', ctypes.POINTER(GraphStruct)),
              ('d', ctypes.c_int32), ('ld', ctypes.c_int32), ('n_state_rows', ctypes.c_int32), ('proj_row_begin', ctypes.c_int32), ('proj_row_end', ctypes.c_int32),
              ('flags', ctypes.c_int32),
              ('alpha', c_vp), ('beta', c_vp), ('x0', c_vp), ('alpha_sigmoid', ctypes.c_int32),
              ('w_csr', c_vp),
              ('proj_w', c_vp), ('proj_b', c_vp), ('proj_m', ctypes.c_int32),
              ('att', AttentionStruct)]


class DecoderStruct(ctypes.Structure):
  _fields_ = [('weight', c_vp), ('bias', c_vp), ('labels', c_vp), ('split', c_vp),
              ('n_classes', ctypes.c_int32), ('d_dec', ctypes.c_int32)]


class HaloStruct(ctypes.Structure):
  _fields_ = [('world', ctypes.c_int32), ('rank', ctypes.c_int32), ('n_own', ctypes.c_int32), ('n_halo', ctypes.c_int32),
              ('send_idx', c_vp), ('send_counts', c_int_p), ('recv_counts', c_int_p)]


class GeneralStruct(ctypes.Structure):
  _fields_ = [('att_graph', ctypes.POINTER(GraphStruct)), ('spmm_graph', ctypes.POINTER(GraphStruct)), ('att', ctypes.POINTER(AttentionStruct)),
              ('qk', c_vp), ('w', c_vp), ('stats_send', c_vp), ('att_ws', c_vp), ('att_ws_bytes', ctypes.c_size_t),
              ('spmm_ws', c_vp), ('spmm_ws_bytes', ctypes.c_size_t), ('stats_buffer', ctypes.c_int32), ('in_offset', ctypes.c_int64),
              ('peer_in_offset', ctypes.POINTER(ctypes.c_int64)), ('peer_buffer_bytes', ctypes.POINTER(ctypes.c_int64)),
              ('peer_rev_row0', ctypes.POINTER(ctypes.c_int64))]


XCD_CONTIGUOUS, XCD_HASHED = 0, 1
COMM_ID_BYTES = 128
P2P_HANDLE_BYTES = 128
EARLY_STATE_INTS = 8 + 3 * 2048

# name -> (restype, argtypes); every symbol include/gnpde.h declares
PROTOTYPES = {
  'gnpde_abi_version': (ctypes.c_int, []),
  'gnpde_last_error': (ctypes.c_char_p, []),
  'gnpde_tune': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32]),
  'gnpde_graph_count_long': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int32, c_int_p, c_int_p, c_int_p]),
  'gnpde_graph_build': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int32] + [c_vp] * 15),
  'gnpde_partition_rows': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_uint64, c_vp]),
  'gnpde_partition_rows_ex': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, c_vp]),
  'gnpde_partition_refine_links': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                  c_vp, c_vp]),
  'gnpde_push_order': (ctypes.c_int, [c_int_p, ctypes.c_int32, c_int_p]),
  'gnpde_xcd_row_map': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_int_p, c_int_p, c_vp]),
  'gnpde_spmm_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(GraphStruct), ctypes.c_int32]),
  'gnpde_spmm_rhs': (ctypes.c_int, [ctypes.POINTER(GraphStruct), c_vp, c_vp, ctypes.c_int32, ctypes.c_int32,
                                    ctypes.POINTER(EpilogueStruct), c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_spmm': (ctypes.c_int, [ctypes.POINTER(GraphStruct), c_vp, c_vp, ctypes.c_int32, ctypes.c_int32,
                                c_vp, c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_sddmm': (ctypes.c_int, [ctypes.POINTER(GraphStruct), c_vp, ctypes.c_int32, c_vp, ctypes.c_int32, ctypes.c_int32,
                                 c_vp, ctypes.c_int32, c_vp, c_vp]),
  'gnpde_softmax_rows_bwd': (ctypes.c_int, [ctypes.POINTER(GraphStruct), c_vp, ctypes.c_int32, c_vp, c_vp, c_vp, ctypes.c_int32, c_vp, c_vp]),
  'gnpde_attention_rows_bwd': (ctypes.c_int, [ctypes.POINTER(GraphStruct), ctypes.POINTER(AttentionStruct), c_vp, c_vp,
                                              ctypes.c_int32, c_vp, c_vp]),
  'gnpde_head_spmm': (ctypes.c_int, [ctypes.POINTER(GraphStruct), ctypes.c_int32, c_vp, ctypes.c_int32, ctypes.c_int32, c_vp,
                                     ctypes.c_int32, ctypes.c_float, c_vp, ctypes.c_int32, c_vp]),
  'gnpde_linear': (ctypes.c_int, [c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_vp, ctypes.c_int32,
                                  ctypes.c_int32, c_vp, c_vp, ctypes.c_int32, c_vp]),
  'gnpde_relu_linear': (ctypes.c_int, [c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_vp, ctypes.c_int32,
                                       ctypes.c_int32, c_vp, c_vp, ctypes.c_int32, c_vp]),
  'gnpde_attention_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(GraphStruct), ctypes.POINTER(AttentionStruct)]),
  'gnpde_edge_attention_pass': (ctypes.c_int, [ctypes.POINTER(GraphStruct), ctypes.POINTER(AttentionStruct), ctypes.c_int32, c_vp, c_vp,
                                               ctypes.c_size_t, c_vp]),
  'gnpde_attention_workspace_regions': (ctypes.c_int, [ctypes.POINTER(GraphStruct), ctypes.POINTER(AttentionStruct), c_vp]),
  'gnpde_segment_stats_merge': (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32, c_vp, c_vp, ctypes.c_int32, c_vp]),
  'gnpde_edge_attention': (ctypes.c_int, [ctypes.POINTER(GraphStruct), ctypes.POINTER(AttentionStruct),
                                          c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_attention_bwd_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(GraphStruct), ctypes.POINTER(AttentionStruct)]),
  'gnpde_edge_attention_bwd': (ctypes.c_int, [ctypes.POINTER(GraphStruct), ctypes.POINTER(AttentionStruct), c_vp, c_vp,
                                              ctypes.c_int32, c_vp, c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_edge_attention_bwd_heads': (ctypes.c_int, [ctypes.POINTER(GraphStruct), ctypes.POINTER(AttentionStruct), c_vp,
                                                    ctypes.c_int32, c_vp, c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_attn_rhs_fused_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(GraphStruct), ctypes.c_int32, ctypes.c_int32]),
  'gnpde_attn_rhs_fused_supported': (ctypes.c_int, [ctypes.POINTER(AttentionStruct), ctypes.c_int32, ctypes.c_int32]),
  'gnpde_attn_rhs_fused': (ctypes.c_int, [ctypes.POINTER(GraphStruct), ctypes.POINTER(AttentionStruct), c_vp, c_vp, c_vp,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(EpilogueStruct), c_vp,
                                          ctypes.c_size_t, c_vp]),
  'gnpde_edge_to_csr_mean': (ctypes.c_int, [ctypes.POINTER(GraphStruct), c_vp, ctypes.c_int32, c_vp, c_vp]),
  'gnpde_solver_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(RhsStruct), ctypes.c_int32]),
  'gnpde_solver_create': (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.POINTER(RhsStruct), ctypes.c_int32,
                                         c_float_p, ctypes.c_int32, c_vp, ctypes.c_size_t]),
  'gnpde_solver_run': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, c_vp]),
  'gnpde_adjoint_grad_floats': (ctypes.c_int, [ctypes.POINTER(RhsStruct)]),
  'gnpde_adjoint_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(RhsStruct), ctypes.POINTER(GraphStruct), ctypes.c_int32]),
  'gnpde_adjoint_create': (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.POINTER(RhsStruct), ctypes.POINTER(GraphStruct), c_vp, c_vp, c_vp,
                                          ctypes.c_int32, c_float_p, ctypes.c_int32, c_vp, ctypes.c_size_t]),
  'gnpde_adjoint_run': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int32, c_vp]),
  'gnpde_linear_split_supported': (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
  'gnpde_linear_split': (ctypes.c_int, [c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_vp, ctypes.c_int32, ctypes.c_int32, c_vp, c_vp, c_vp,
                                        ctypes.c_int32, c_vp]),
  'gnpde_adjoint_set_tape': (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, c_vp, c_vp]),
  'gnpde_adjoint_tape_swapped': (ctypes.c_int, [c_vp]),
  'gnpde_solver_tape_bytes': (ctypes.c_size_t, [ctypes.POINTER(RhsStruct), ctypes.c_int32, ctypes.c_int32]),
  'gnpde_solver_set_tape': (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t]),
  'gnpde_adjoint_num_rhs_evals': (ctypes.c_int, [c_vp]),
  'gnpde_adjoint_destroy': (ctypes.c_int, [c_vp]),
  'gnpde_rhs_eval': (ctypes.c_int, [ctypes.POINTER(RhsStruct), c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_rhs_stage': (ctypes.c_int, [ctypes.POINTER(RhsStruct), c_vp, ctypes.POINTER(EpilogueStruct), c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_rk_error_ratio': (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_float_p, ctypes.c_int32, ctypes.c_float,
                                          ctypes.c_float, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_vp, c_vp, c_vp]),
  'gnpde_dopri5_interp': (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_float_p, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, c_vp, c_vp]),
  'gnpde_rhs_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(RhsStruct)]),
  'gnpde_early_stop_reset': (ctypes.c_int, [c_vp, c_vp]),
  'gnpde_early_stop_eval': (ctypes.c_int, [ctypes.POINTER(DecoderStruct), c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_int32, c_vp, c_vp, ctypes.c_int32, c_vp]),
  'gnpde_solver_set_early_stop': (ctypes.c_int, [c_vp, ctypes.POINTER(DecoderStruct), c_vp, c_vp, ctypes.c_int32]),
  'gnpde_lincomb': (ctypes.c_int, [c_vp, ctypes.POINTER(c_vp), c_float_p, ctypes.c_int32, ctypes.c_int64, c_vp, c_vp]),
  'gnpde_solver_num_rhs_evals': (ctypes.c_int, [c_vp]),
  'gnpde_solver_destroy': (ctypes.c_int, [c_vp]),
  'gnpde_gather_rows': (ctypes.c_int, [c_vp, ctypes.c_int32, c_vp, ctypes.c_int32, ctypes.c_int32, c_vp,
                                       ctypes.c_int32, c_vp]),
  'gnpde_gather_ceiling': (ctypes.c_int, [c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_vp, ctypes.c_int32, c_vp,
                                          ctypes.c_int32, ctypes.c_int32, c_vp]),
  'gnpde_stream_read': (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.c_int32, c_vp, ctypes.c_int32, c_vp]),
  'gnpde_quantile_workspace_bytes': (ctypes.c_size_t, []),
  'gnpde_quantile': (ctypes.c_int, [c_vp, ctypes.c_int64, ctypes.c_double, c_vp, c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_graph_build_device_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int32]),
  'gnpde_graph_build_device': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, ctypes.c_int32] + [c_vp] * 8 + [c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_graph_build_device_long': (ctypes.c_int, [c_vp, c_vp] + [ctypes.c_int32] * 4 + [c_vp] * 8 + [c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_dopri5_workspace_bytes': (ctypes.c_size_t, [c_vp]),
  'gnpde_dopri5_create': (ctypes.c_int, [c_vp, c_vp, ctypes.c_float, ctypes.c_float, c_vp, ctypes.c_size_t]),
  'gnpde_dopri5_sharded_workspace_bytes': (ctypes.c_size_t, [c_vp]),
  'gnpde_dopri5_create_sharded': (ctypes.c_int, [ctypes.POINTER(c_vp), c_vp, ctypes.c_float, ctypes.c_float, ctypes.c_int64, c_vp,
                                                 ctypes.c_size_t]),
  'gnpde_dopri5_run': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, ctypes.c_double, ctypes.c_double, c_vp, ctypes.c_int32,
                                      ctypes.c_int32, ctypes.c_int32, c_vp, c_vp]),
  'gnpde_dopri5_set_early_stop': (ctypes.c_int, [c_vp, ctypes.POINTER(DecoderStruct), c_vp, c_vp, ctypes.c_int32, c_vp, ctypes.c_int32,
                                                 ctypes.c_int32]),
  'gnpde_dopri5_stats': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
  'gnpde_dopri5_set_row_order': (ctypes.c_int, [c_vp, c_vp]),
  'gnpde_dopri5_set_pair': (ctypes.c_int, [c_vp, ctypes.c_int32]),
  'gnpde_dopri5_destroy': (ctypes.c_int, [c_vp]),
  'gnpde_adjoint_adaptive_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(RhsStruct), ctypes.POINTER(GraphStruct), ctypes.c_int32]),
  'gnpde_adjoint_adaptive_create': (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.POINTER(RhsStruct), ctypes.POINTER(GraphStruct), c_vp, ctypes.c_int32,
                                                   ctypes.c_float, ctypes.c_float, c_vp, ctypes.c_size_t]),
  'gnpde_adjoint_adaptive_run': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, c_vp, ctypes.c_int32, c_vp, ctypes.c_double, ctypes.c_double,
                                                ctypes.c_double, ctypes.c_int32, ctypes.c_int32, c_vp, c_vp]),
  'gnpde_adjoint_adaptive_stats': (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
  'gnpde_adjoint_adaptive_destroy': (ctypes.c_int, [c_vp]),
  'gnpde_dopri5_tape_bytes': (ctypes.c_size_t, [ctypes.POINTER(RhsStruct), ctypes.c_int32]),
  'gnpde_dopri5_set_tape': (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, ctypes.c_int32]),
  'gnpde_dopri5_tape_steps': (ctypes.c_int, [c_vp]),
  'gnpde_dopri5_tape_record': (ctypes.c_int, [c_vp, c_float_p, ctypes.c_int32, c_float_p]),
  'gnpde_dopri5_tape_backward_workspace_bytes': (ctypes.c_size_t, [c_vp, ctypes.POINTER(GraphStruct)]),
  'gnpde_dopri5_tape_backward': (ctypes.c_int, [c_vp, ctypes.POINTER(GraphStruct), c_vp, c_vp, ctypes.c_int32, c_vp, ctypes.c_int32, c_vp, c_vp, c_vp, c_vp,
                                                ctypes.c_size_t, c_vp]),
  'gnpde_two_hop_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int32]),
  'gnpde_two_hop_count': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, c_vp, c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_two_hop_fill': (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int32, c_vp, c_vp, ctypes.c_int64, c_vp, c_vp,
                                        ctypes.c_size_t, c_vp]),
  'gnpde_threshold_edges_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int32]),
  'gnpde_threshold_edges': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int64, c_vp, ctypes.c_int32, ctypes.c_int32, c_vp, c_vp, c_vp,
                                           c_vp, ctypes.c_size_t, c_vp]),
  'gnpde_comm_load_library': (ctypes.c_int, [ctypes.c_char_p]),
  'gnpde_comm_get_unique_id': (ctypes.c_int, [c_vp]),
  'gnpde_comm_create': (ctypes.c_int, [ctypes.POINTER(c_vp), c_vp, ctypes.c_int32, ctypes.c_int32]),
  'gnpde_comm_destroy': (ctypes.c_int, [c_vp]),
  'gnpde_p2p_create': (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, ctypes.c_int32]),
  'gnpde_p2p_get_handle': (ctypes.c_int, [c_vp, c_vp]),
  'gnpde_p2p_connect': (ctypes.c_int, [c_vp, c_vp]),
  'gnpde_p2p_buffer': (c_vp, [c_vp, ctypes.c_int32]),
  'gnpde_p2p_destroy': (ctypes.c_int, [c_vp]),
  'gnpde_sharded_solver_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(HaloStruct), ctypes.POINTER(RhsStruct),
                                                             ctypes.POINTER(RhsStruct), ctypes.c_int32, ctypes.c_int32]),
  'gnpde_sharded_solver_create_p2p': (ctypes.c_int, [ctypes.POINTER(c_vp), c_vp, ctypes.POINTER(HaloStruct),
                                                     ctypes.POINTER(RhsStruct), ctypes.POINTER(RhsStruct), ctypes.c_int32,
                                                     c_float_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64),
                                                     ctypes.POINTER(ctypes.c_int64), c_vp, ctypes.c_size_t]),
  'gnpde_sharded_solver_set_general': (ctypes.c_int, [c_vp, ctypes.POINTER(GeneralStruct)]),
  'gnpde_sharded_solver_status': (ctypes.c_int, [c_vp, c_int_p, ctypes.POINTER(ctypes.c_int64)]),
  'gnpde_sharded_solver_timing': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, c_vp, c_vp]),
  'gnpde_sharded_solver_set_spin_limit': (ctypes.c_int, [c_vp, ctypes.c_int64]),
  'gnpde_sharded_solver_set_boundary_chunks': (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.POINTER(RhsStruct)), ctypes.c_int32, c_int_p, c_int_p]),
  'gnpde_sharded_solver_create': (ctypes.c_int, [ctypes.POINTER(c_vp), c_vp, ctypes.POINTER(HaloStruct),
                                                 ctypes.POINTER(RhsStruct), ctypes.POINTER(RhsStruct), ctypes.c_int32,
                                                 c_float_p, ctypes.c_int32, c_vp, ctypes.c_size_t]),
  'gnpde_sharded_solver_run': (ctypes.c_int, [c_vp, c_vp, ctypes.c_int32, c_vp]),
  'gnpde_sharded_solver_num_rhs_evals': (ctypes.c_int, [c_vp]),
  'gnpde_sharded_solver_destroy': (ctypes.c_int, [c_vp]),
}


def build(verbose=False):
  """Compile csrc/*.hip for gfx950 into csrc/libgnpde_hip.so (hipcc cross-compiles without a GPU)."""
  script = os.path.join(_HERE, 'csrc', 'build.sh')
  res = subprocess.run(['bash', script], capture_output=True, text=True)
  if verbose or res.returncode != 0:
    print(res.stdout)
    print(res.stderr)
  if res.returncode != 0:
    raise RuntimeError('building libgnpde_hip.so failed')
  return LIB_PATH


def lib():
  """The loaded library; raises if it has not been built (no fallback path exists)."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError('libgnpde_hip.so is missing (%s): run `python -c "import __graft_entry__ as g; g.build()"`. '
                         'There is no CPU / PyTorch fallback for the ODE right-hand side.' % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
      fn = getattr(handle, name)
      fn.restype = res
      fn.argtypes = args
    if handle.gnpde_abi_version() != ABI_VERSION:
      raise RuntimeError('libgnpde_hip.so ABI version mismatch: the library says %d, this package was written for %d -- rebuild '
                         '(python -c "import __graft_entry__ as g; g.build()")' % (handle.gnpde_abi_version(), ABI_VERSION))
    _lib = handle
  return _lib


class GnpdeError(RuntimeError):
  pass


def check(rc):
  if rc != 0:
    msg = lib().gnpde_last_error().decode(errors='replace')
    raise GnpdeError('libgnpde_hip error %d: %s' % (rc, msg))


def ptr(t):
  """Device (or host) address of a tensor, None -> NULL."""
  if t is None:
    return None
  return ctypes.c_void_p(t.data_ptr())


def require_hip(*tensors):
  for t in tensors:
    if t is not None and not t.is_cuda:
      raise GnpdeError('the ODE right-hand side runs only on a HIP device (got a %s tensor); there is no CPU fallback'
                       % t.device.type)


def f32c(t, name='tensor'):
  if t.dtype != torch.float32:
    raise GnpdeError('%s must be float32 (got %s)' % (name, t.dtype))
  return t if t.is_contiguous() else t.contiguous()


RHS_PADDED_ROWS = 1


def f32rows(t, name='tensor'):
  """A float32 matrix whose rows may be padded (unit column stride, row stride >= width) is handed on as it is; anything
  else is made contiguous."""
  if t.dtype != torch.float32:
    raise GnpdeError('%s must be float32 (got %s)' % (name, t.dtype))
  if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
    return t
  return t if t.is_contiguous() else t.contiguous()


def alloc_state(n, d, device):
  """[n, d] float32 state buffer; when d is not a multiple of 4 its rows are padded to the next multiple (a view of an
  [n, ld] allocation), which lets the kernels use 16-byte lanes (GNPDE_RHS_PADDED_ROWS)."""
  ld = (d + 3) // 4 * 4
  out = torch.zeros(n, ld, dtype=torch.float32, device=device)[:, :d]
  if ld != d:
    out._gnpde_padded = True     # marks THIS tensor object: a column slice of somebody else's matrix never qualifies
  return out


def is_padded(t):
  """True only for buffers handed out by alloc_state (their columns [d, ld) are ours to overwrite)."""
  return bool(getattr(t, '_gnpde_padded', False))


def stream_of(t):
  return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
