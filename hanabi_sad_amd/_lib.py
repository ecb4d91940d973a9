"""ctypes binding of libhsad.so (include/hsad.h).  Fails loudly when the library is missing —
the product has no CPU or PyTorch fallback for the hot path."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libhsad.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("hsad_env.hip", "hsad_replay.hip", "hsad_r2d2.hip", "hsad_r2d2_f32.hip", "hsad_agent.hip", "hsad_comm.hip",
                                                      "hsad_actor.hip")]
_lib = None


class HsadError(RuntimeError):
    pass


class EnvConfig(C.Structure):
    _fields_ = [
        ("num_games", C.c_int32), ("players", C.c_int32), ("hand_size", C.c_int32), ("bomb", C.c_int32),
        ("seed0", C.c_int32), ("max_len", C.c_int32), ("sad", C.c_int32), ("shuffle_obs", C.c_int32),
        ("shuffle_color", C.c_int32), ("knowledge_mode", C.c_int32), ("n_eps", C.c_int32), ("device", C.c_int32),
        ("track_deck_history", C.c_int32), ("deal_mode", C.c_int32), ("games_per_workgroup", C.c_int32),
        ("eps_list", C.POINTER(C.c_float)),
    ]


class Field(C.Structure):
    _fields_ = [("width", C.c_int32), ("dtype", C.c_int32)]


# every symbol include/hsad.h declares: (restype, argtypes)
_P = C.c_void_p
class LstmFwdRec(C.Structure):
    """hsad_lstm_fwd_rec (include/hsad.h)"""
    _fields_ = [("gates", C.c_void_p), ("Whh_blocked", C.c_void_p), ("h_prev16", C.c_void_p), ("c_prev", C.c_void_p),
                ("hseq16", C.c_void_p), ("cseq", C.c_void_p), ("hT", C.c_void_p), ("xchg", C.c_void_p)]


class LstmBwdRec(C.Structure):
    """hsad_lstm_bwd_rec (include/hsad.h)"""
    _fields_ = [("gates", C.c_void_p), ("cseq", C.c_void_p), ("c_before", C.c_void_p), ("WhhT_blocked", C.c_void_p),
                ("dO", C.c_void_p), ("dG16", C.c_void_p), ("dc_io", C.c_void_p), ("has_next", C.c_int), ("xchg", C.c_void_p),
                ("saved_frag_major", C.c_int), ("tail_is_zero", C.c_int)]


class ActorConfig(C.Structure):
    """hsad_actor_config (include/hsad.h)"""
    _fields_ = [("vdn", C.c_int32), ("multi_step", C.c_int32), ("seq_len", C.c_int32), ("hand_size", C.c_int32), ("hid_dim", C.c_int32),
                ("gamma", C.c_float), ("eta", C.c_float), ("seed", C.c_uint64)]


class ActorIO(C.Structure):
    """hsad_actor_io (include/hsad.h)"""
    _fields_ = [(k, C.c_void_p) for k in ("legal_move", "own_hand", "eps", "reward", "terminal", "priv_bits", "legal_bits", "own_bits", "priv_s_bf16")]


class LstmFusedBwdRec(C.Structure):
    """hsad_lstm_fused_bwd_rec (include/hsad.h)"""
    _fields_ = [("WhhT_blocked", C.c_void_p), ("WihT_above_blocked", C.c_void_p), ("gates", C.c_void_p), ("cseq", C.c_void_p),
                ("c_before", C.c_void_p), ("dO", C.c_void_p), ("dG16", C.c_void_p), ("dc_io", C.c_void_p), ("has_next", C.c_int),
                ("xchg", C.c_void_p), ("saved_frag_major", C.c_int), ("tail_is_zero", C.c_int), ("xout", C.c_void_p), ("dO_stage", C.c_void_p), ("sink_WT", C.c_void_p), ("sink_out16", C.c_void_p), ("sink_mask16", C.c_void_p),
                ("sink_xout", C.c_void_p), ("sink_outT16", C.c_void_p), ("sink_ldT", C.c_int), ("sink_bias_grad", C.c_void_p), ("dGT16", C.c_void_p), ("ldT", C.c_int), ("bias_grad0", C.c_void_p), ("bias_grad1", C.c_void_p),
                ("bias_col_map", C.c_void_p), ("layout_steps", C.c_int), ("wide_blocks", C.c_int)]


class LstmFusedRec(C.Structure):
    """hsad_lstm_fused_rec (include/hsad.h)"""
    _fields_ = [("Wih_blocked", C.c_void_p), ("Whh_blocked", C.c_void_p), ("bias_blocked", C.c_void_p), ("x16", C.c_void_p),
                ("gates", C.c_void_p), ("cseq", C.c_void_p), ("hseq16", C.c_void_p), ("hT", C.c_void_p)]


SIGNATURES = {
    "hsad_last_error": (C.c_char_p, []),
    "hsad_version": (C.c_char_p, []),
    "hsad_env_create": (C.c_int, [C.POINTER(EnvConfig), C.POINTER(_P)]),
    "hsad_env_destroy": (None, [_P]),
    "hsad_env_feature_size": (C.c_int, [_P]),
    "hsad_env_num_action": (C.c_int, [_P]),
    "hsad_env_hand_feature_size": (C.c_int, [_P]),
    "hsad_env_num_games": (C.c_int, [_P]),
    "hsad_env_num_players": (C.c_int, [_P]),
    "hsad_env_games_per_workgroup": (C.c_int, [_P]),
    "hsad_env_threads_per_workgroup": (C.c_int, [_P]),
    "hsad_env_set_threads_per_workgroup": (C.c_int, [_P, C.c_int]),
    "hsad_env_state_bytes": (C.c_int64, [_P]),
    "hsad_env_bind_outputs": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "hsad_env_bind_packed": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int]),
    "hsad_env_reset": (C.c_int, [_P, _P]),
    "hsad_env_step": (C.c_int, [_P, _P, _P, _P]),
    "hsad_env_policy_random": (C.c_int, [_P, C.c_uint64, _P, _P, _P]),
    "hsad_env_rollout_random": (C.c_int, [_P, C.c_int, C.c_uint64, _P, _P, _P]),
    "hsad_env_set_partitions": (C.c_int, [_P, C.c_int]),
    "hsad_env_set_rollout_stagger": (C.c_int, [_P, C.c_int]),
    "hsad_env_set_rollout_chunk": (C.c_int, [_P, C.c_int]),
    "hsad_env_last_rollout_ms": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "hsad_env_query": (C.c_int, [_P, _P, _P]),
    "hsad_env_move_is_legal": (C.c_int, [_P, _P, _P, _P]),
    "hsad_env_deck_history": (C.c_int, [_P, _P, _P, _P]),
    "hsad_env_state_words": (C.c_int, [_P]),
    "hsad_env_export_state": (C.c_int, [_P, _P, _P]),
    "hsad_env_debug_timing": (C.c_int, [_P, _P]),
    "hsad_env_error_count": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hsad_aggregate_priority": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, _P, _P]),
    "hsad_replay_create": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(Field), C.c_int, C.POINTER(_P)]),
    "hsad_replay_destroy": (None, [_P]),
    "hsad_replay_bytes": (C.c_int64, [_P]),
    "hsad_replay_add": (C.c_int, [_P, C.c_int, C.POINTER(_P), _P, _P, _P, _P, _P, _P, _P]),
    "hsad_replay_sample": (C.c_int, [_P, C.c_int, C.POINTER(_P), _P, _P, _P, _P, _P, _P]),
    "hsad_replay_priority_sum": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "hsad_replay_draw_canonical": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    "hsad_replay_sample_at": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.POINTER(_P), _P, _P, _P, _P, _P, _P]),
    "hsad_replay_update_priority": (C.c_int, [_P, _P, C.c_int, _P]),
    "hsad_replay_size": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hsad_replay_get": (C.c_int, [_P, C.c_int, C.POINTER(_P), _P, _P, _P, _P, _P]),
    "hsad_replay_last_ids": (C.c_int, [_P, _P, C.c_int, _P]),
    "hsad_replay_set_outstanding": (C.c_int, [_P, C.c_int]),
    "hsad_replay_error_kinds": (C.c_int, [_P]),
    "hsad_replay_stats": (C.c_int, [_P, _P, _P]),
    "hsad_replay_wire_bytes": (C.c_int, [_P]),
    "hsad_replay_serve": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "hsad_replay_update_owned": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P]),
    "hsad_replay_assemble": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.POINTER(_P), _P, _P, _P, _P, _P, _P]),
    "hsad_replay_set_field_output": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "hsad_replay_row_bytes": (C.c_int, [_P]),
    "hsad_replay_field_bytes": (C.c_int, [_P, C.c_int]),
    "hsad_seqwriter_set_prepacked": (C.c_int, [_P, C.c_uint32]),
    "hsad_replay_error_count": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "hsad_seqwriter_create": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(Field), C.c_int,
                                        C.POINTER(_P)]),
    "hsad_seqwriter_destroy": (None, [_P]),
    "hsad_seqwriter_push_obs_action": (C.c_int, [_P, C.POINTER(_P), _P]),
    "hsad_seqwriter_push_reward_terminal": (C.c_int, [_P, _P, _P, _P]),
    "hsad_seqwriter_push_reward_terminal_rep": (C.c_int, [_P, _P, _P, C.c_int, _P]),
    "hsad_seqwriter_can_pop": (C.c_int, [_P]),
    "hsad_seqwriter_pop_transition": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), _P, _P, _P, _P]),
    "hsad_seqwriter_push_sequence": (C.c_int, [_P, _P, _P]),
    "hsad_seqwriter_step_tail": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_double, _P, _P, _P, _P]),
    "hsad_seqwriter_step_tail_ready": (C.c_int, [_P]),
    "hsad_seqwriter_flush_to_replay": (C.c_int, [_P, _P, C.c_float, _P, _P]),
    "hsad_gemm_nt_bf16": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int,
                                    C.c_int, C.c_int, _P]),
    "hsad_cast_pad_bf16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "hsad_gemm_nt_bf16_splitk": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P]),
    "hsad_gemm_nt_bf16_splitk_acc": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P]),
    "hsad_transpose_bf16_colsum": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P]),
    "hsad_prepare_weight": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int, _P]),
    "hsad_bias_sum_perm": (C.c_int, [_P, _P, _P, _P, C.c_int, _P]),
    "hsad_transpose_bf16": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "hsad_lstm_layer_forward": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "hsad_lstm_cell_fused": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hsad_lstm_cell_set_variant": (C.c_int, [C.c_int, C.c_int]),
    "hsad_gemm_set_pp": (C.c_int, [C.c_int]),
    "hsad_gemm_group_workspace_floats": (C.c_int64, [C.c_int, _P]),
    "hsad_gemm_nt_bf16_group_splitk": (C.c_int, [C.c_int, _P, _P, C.c_int64, _P]),
    "hsad_lstm_cell_fused_pair": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int] + [_P] * 17),
    "hsad_lstm_set_exchange_mode": (C.c_int, [C.c_int]),
    "hsad_lstm_debug_timing": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "hsad_lstm_debug_timing32": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "hsad_lstm_debug_enable": (C.c_int, [C.c_int]),
    "hsad_lstm_debug_trace": (C.c_int, [C.POINTER(C.c_uint64), C.c_size_t]),
    "hsad_debug_resident_kernel": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "hsad_lstm_sync_timed_out": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "hsad_q_head": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "hsad_gemm_nt_bf16_ex": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int,
                                       C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "hsad_lstm_layer_backward": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hsad_heads_backward": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P,
                                      C.c_int, _P]),
    "hsad_aux_xent": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "hsad_loss_tail": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                 C.c_float, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "hsad_colsum": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "hsad_colsum_acc": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "hsad_colsum_acc_ordered": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "hsad_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_int, _P, _P]),
    "hsad_adam_step_zero_grad": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _P,
                                          C.POINTER(_P), _P]),
    "hsad_act_select": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
    "hsad_nstep_priority": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_double, C.c_int, _P, _P]),
    "hsad_zero_rows": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "hsad_gemm_nt_bf16_pair": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, C.c_int,
                                         C.c_int, _P]),
    "hsad_gemm_timing": (C.c_int, [C.c_int]),
    "hsad_gemm_timing_read": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "hsad_lstm_cell_timing": (C.c_int, [C.c_int]),
    "hsad_lstm_cell_timing_read": (C.c_int, [_P, _P, _P]),
    "hsad_refresh_begin": (C.c_int, []),
    "hsad_refresh_add_weight": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int]),
    "hsad_refresh_add_bias": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "hsad_refresh_launch": (C.c_int, [_P]),
    "hsad_act_select_q": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P]),
    "hsad_act_select_q2": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P, _P]),
    "hsad_q_at": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P]),
    "hsad_zero_state_rows": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "hsad_lstm_forward_chunk_multi": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "hsad_lstm_backward_chunk_multi": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "hsad_lstm_forward_fused": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "hsad_lstm_backward_fused": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "hsad_lstm_fused_timing": (C.c_int, [C.c_int]),
    "hsad_lstm_fused_timing_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "hsad_lstm_fused_timing_read_kind": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "hsad_lstm_forward_chunk": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hsad_lstm_backward_chunk": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "hsad_gemm_f32": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int,
                                C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "hsad_actor_create": (C.c_int, [_P, _P, _P, _P, C.POINTER(ActorConfig), C.POINTER(ActorIO), C.POINTER(_P)]),
    "hsad_actor_destroy": (None, [_P]),
    "hsad_actor_step": (C.c_int, [_P, _P]),
    "hsad_actor_set_run_ahead": (C.c_int, [_P, C.c_int]),
    "hsad_actor_num_act": (C.c_int64, [_P]),
    "hsad_actor_num_redo": (C.c_int64, [_P]),
    "hsad_actor_n_finished_dev": (_P, [_P]),
    "hsad_actor_writer": (_P, [_P]),
    "hsad_actor_state": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "hsad_actor_last_actions": (_P, [_P, C.POINTER(_P)]),
    "hsad_actor_last_priority": (_P, [_P, C.POINTER(C.c_int32)]),
    "hsad_r2d2_net_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "hsad_r2d2_net_create_ex": (C.c_int, [C.c_int] * 9 + [C.POINTER(_P)]),
    "hsad_r2d2_net_num_params": (C.c_int, [_P]),
    "hsad_r2d2_net_param_name": (C.c_char_p, [_P, C.c_int]),
    "hsad_r2d2_net_arch": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "hsad_r2d2_net_destroy": (None, [_P]),
    "hsad_r2d2_num_params": (C.c_int, []),
    "hsad_r2d2_param_name": (C.c_char_p, [C.c_int]),
    "hsad_r2d2_net_param_count": (C.c_int64, [_P]),
    "hsad_r2d2_net_param_offset": (C.c_int64, [_P, C.c_int]),
    "hsad_r2d2_net_param_size": (C.c_int64, [_P, C.c_int]),
    "hsad_r2d2_net_params": (_P, [_P]),
    "hsad_r2d2_net_refresh": (C.c_int, [_P, _P]),
    "hsad_r2d2_net_version": (C.c_uint64, [_P]),
    "hsad_r2d2_net_in_dim_padded": (C.c_int, [_P]),
    "hsad_r2d2_target_q": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hsad_comm_unique_id": (C.c_int, [_P, C.c_int]),
    "hsad_comm_init": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "hsad_comm_destroy": (None, [_P]),
    "hsad_comm_rank": (C.c_int, [_P]),
    "hsad_comm_world": (C.c_int, [_P]),
    "hsad_comm_bcast_params": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "hsad_comm_gather_batch": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, _P, _P]),
    "hsad_comm_scatter_priority": (C.c_int, [_P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "hsad_comm_star_round": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "hsad_comm_star_open": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, _P, _P, _P, C.c_int64, _P, _P]),
    "hsad_comm_star_collect": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "hsad_comm_star_serve": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int64, _P]),
    "hsad_comm_all_stats": (_P, [_P]),
    "hsad_ipc_handle_bytes": (C.c_int, []),
    "hsad_ipc_alloc": (C.c_int, [C.c_int64, C.POINTER(_P), _P, C.c_int]),
    "hsad_ipc_free": (C.c_int, [_P]),
    "hsad_ipc_open": (C.c_int, [_P, C.POINTER(_P)]),
    "hsad_ipc_close": (C.c_int, [_P]),
    "hsad_ipc_put": (C.c_int, [_P, _P, C.c_int64, _P]),
    "hsad_r2d2_act": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hsad_r2d2_q_of": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "hsad_r2d2_compute_priority": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_double,
                                             _P, _P, _P]),
    "hsad_r2d2_learner_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_float, C.POINTER(_P)]),
    "hsad_r2d2_learner_destroy": (None, [_P]),
    "hsad_r2d2_learner_set_schedule": (C.c_int, [_P, C.c_int, C.c_int]),
    "hsad_r2d2_learner_set_fused": (C.c_int, [_P, C.c_int]),
    "hsad_r2d2_learner_grad": (_P, [_P]),
    "hsad_r2d2_learner_timed_out": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "hsad_r2d2_learner_inject_timeout": (C.c_int, [_P, C.c_int]),
    "hsad_r2d2_loss_bwd_weighted": (C.c_int, [_P, _P, _P, _P]),
    "hsad_r2d2_learner_set_optim": (C.c_int, [_P, C.c_float, C.c_float, C.c_float]),
    "hsad_r2d2_loss_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_float, _P, _P, C.c_int, _P]),
    "hsad_r2d2_loss_bwd": (C.c_int, [_P, _P]),
    "hsad_r2d2_optimizer_step": (C.c_int, [_P, C.c_float, C.c_float, C.POINTER(_P), _P]),
    "hsad_r2d2_learner_grad_norm_dev": (_P, [_P]),
    "hsad_r2d2_sync_target_with_online": (C.c_int, [_P, _P]),
    "hsad_eltwise_mul": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P]),
    "hsad_lstm_cell_f32_forward": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "hsad_lstm_cell_f32_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "hsad_heads_backward_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P,
                                          C.c_int, _P]),
    "hsad_td_loss": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, _P, _P]),
}


def build_library(verbose=False):
    """Compile every HIP source for gfx950 into hanabi_sad_amd/libhsad.so (hipcc cross-compiles
    without a GPU)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-inline-asm",
           "-I" + os.path.join(_ROOT, "include")] + SOURCES + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HsadError(
            "libhsad.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "— there is no CPU fallback for the HIP hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise HsadError("hsad call failed (%d): %s" % (rc, load_library().hsad_last_error().decode()))
