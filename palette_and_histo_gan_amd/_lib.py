"""ctypes binding of libp2pgan_hip.so (include/p2pgan.h).

The product path has NO fallback: if the HIP library is missing or a call fails this module raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P2P_LIB") or os.path.join(HERE, "libp2pgan_hip.so")      # P2P_LIB: diagnostic builds (tools/ubench)

F32, BF16 = 0, 1
OP_G, OP_P, OP_W = 0, 1, 2
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2


class Tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("img_stride", C.c_longlong), ("row_stride", C.c_int), ("ld", C.c_int)]


class GSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("kind", C.c_int), ("nslabs", C.c_int), ("slab_stride", C.c_longlong),
                ("ld", C.c_int), ("coff", C.c_int)]


_TP = C.POINTER(Tensor)
_GP = C.POINTER(GSrc)
_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> argtypes; every function returns int except the two noted below
SIGNATURES = {
    "p2p_conv_direct": [_i, _i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _vp, _vp, _vp, _vp],
    "p2p_igemm": [_i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _i, _vp, _vp, _vp],
    "p2p_igemm_norm_act": [_i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _vp, _vp, _f, _i, _f, _TP, _vp, _vp],
    "p2p_igemm_edge": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _vp, _i, _f, _vp],
    "p2p_conv_strip": [_i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _vp, _vp],
    "p2p_conv_fewin": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _vp, _i, _f, _vp],
    "p2p_conv_fewin_actbwd": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _TP, _f, _vp],
    "p2p_conv_fewout": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _vp, _i, _f, _vp],
    "p2p_wgemm_edge": [_i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _i, _vp, _vp],
    "p2p_wgrad_small": [_i, _i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _vp, _vp],
    "p2p_view_colsum": [_i, _i, _i, _i, _i, _TP, _vp, _vp, _vp],
    "p2p_act_bwd": [_i, _i, _i, _i, _i, _TP, _GP, _GP, _f, _TP, _vp],
    "p2p_weight_prep_pad": [_i, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp],
    "p2p_wgemm": [_i, _i, _i, _i, _i, _i, _TP, _TP, _vp, _i, _vp, _vp],
    "p2p_norm_act_fwd": [_i, _i, _i, _i, _i, _vp, _i, _i, _ll, _vp, _vp, _f, _i, _f, _vp, _TP, _vp, _vp, _vp, _ll, _i, _vp],
    "p2p_norm_act_fwd_tail": [_i, _i, _i, _i, _i, _vp, _i, _i, _ll, _vp, _vp, _f, _i, _f, _vp, _TP, _vp, _vp, _vp, _ll, _i, _TP, _i, _vp],
    "p2p_norm_act_bwd": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _GP, _GP, _TP, _vp, _vp, _vp, _ll, _i, _vp],
    "p2p_colsum": [_vp, _i, _i, _f, _vp, _vp],
    "p2p_colsum_batched": [_vp, _vp, _i, _i, _vp, _vp],
    "p2p_bce_logits": [_i, _i, _i, _i, _i, _TP, _f, _TP, _TP, _vp, _vp],
    "p2p_bce_logits_pad8": [_i, _i, _i, _i, _i, _TP, _f, _TP, _TP, _vp, _vp],
    "p2p_loss_partials_sum": [_vp, _i, _vp, _vp],
    "p2p_tanh_l1_fwd": [_i, _i, _i, _i, _i, _TP, _TP, _TP, _f, _vp, _vp, _vp],
    "p2p_tanh_l1_fwd_pair": [_i, _i, _i, _i, _TP, _TP, _TP, _f, _vp, _vp, _vp],
    "p2p_tanh_l1_bwd": [_i, _i, _i, _i, _i, _TP, _TP, _GP, _GP, _f, _TP, _vp],
    "p2p_tanh_l1_bwd_pad8": [_i, _i, _i, _i, _TP, _TP, _GP, _GP, _f, _TP, _vp],
    "p2p_adam_flat": [_vp, _vp, _vp, _vp, _ll, _i, _f, _f, _f, _f, _f, _vp],
    "p2p_adam_tick": [_vp, _vp, _f, _f, _f, _vp],
    "p2p_adam_flat_dev": [_vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _vp],
    "p2p_counter_add": [_vp, _ll, _vp],
    "p2p_dropout_mask_dev": [_vp, _ll, _ll, _vp, _ll, _ll, _vp],
    "p2p_weight_prep": [_i, _vp, _i, _i, _vp, _vp, _vp],
    "p2p_weight_prep_batched": [_i, _vp, _i, _ll, _vp],
    "p2p_adam_prep_batched": [_i, _ll, _vp, _i, _ll, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp],
    "p2p_pack_input": [_i, _i, _i, _i, _i, _vp, _i, _TP, _vp],
    "p2p_pack_pair": [_i, _i, _i, _i, _vp, _vp, _TP, _TP, _TP, _TP, _vp],
    "p2p_pack_pair_idx": [_i, _i, _i, _i, _vp, _vp, _TP, _TP, _TP, _TP, _vp],
    "p2p_pack_input_multi": [_i, _i, _i, _i, _i, _vp, _i, _TP, _i, _vp],
    "p2p_finish_losses": [_vp, _i, _i, _f, _f, _vp, _vp],
    "p2p_unpack": [_i, _i, _i, _i, _i, _TP, _vp, _vp],
    "p2p_dropout_mask": [_vp, _ll, _ll, _ll, _vp],
    "p2p_rgbuv_hist_fwd": [_i, _i, _i, _i, _TP, _vp, _vp],
    "p2p_hist_normalize": [_vp, _i, _vp, _vp],
    "p2p_rgbuv_hist_general": [_i, _i, _i, _i, _TP, _i, _i, _f, _vp, _vp],
    "p2p_rgbuv_hist_fwd3": [_i, _i, _i, _i, _TP, _vp, _vp, _i, _vp, _vp, _vp],
    "p2p_rgbuv_points": [_i, _i, _i, _i, _TP, _i, _vp, _vp, _vp],
    "p2p_hellinger_fwd": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "p2p_hellinger_finish": [_vp, _f, _vp, _vp],
    "p2p_rgbuv_hist_hellinger_bwd": [_i, _i, _i, _i, _TP, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp],
    "p2p_rgbuv_hist_hellinger_bwd3": [_i, _i, _i, _i, _TP, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp],
    "p2p_softmax_cce_argmax": [_i, _i, _i, _i, _i, _TP, _TP, _TP, _f, _f, _TP, _vp, _vp, _vp, _vp],
    "p2p_argmax_lastdim": [_vp, _ll, _i, _vp, _vp],
    "p2p_head_dgrad": [_i, _i, _i, _i, _i, _i, _TP, _vp, _i, _TP, _vp],
    "p2p_head_softmax_cce": [_i, _i, _i, _i, _i, _i, _TP, _vp, _vp, _TP, _TP, _f, _f, _TP, _vp, _vp, _vp, _vp],
    "p2p_comm_unique_id": [_vp],
    "p2p_comm_init": [_vp, _i, _i, C.POINTER(_vp)],
    "p2p_comm_allreduce_sum": [_vp, _vp, _ll, _vp],
    "p2p_comm_destroy": [_vp],
    "p2p_event_create": [C.POINTER(_vp)],
    "p2p_event_destroy": [_vp],
    "p2p_event_record": [_vp, _vp],
    "p2p_stream_wait_event": [_vp, _vp],
    "p2p_arm_stop_event": [_vp],
    "p2p_disarm_stop_event": [C.POINTER(C.c_int)],
    "p2p_png_unfilter": [_vp, _i, _i, _i, _vp],
    "p2p_sprites_rgba_batch": [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "p2p_gather_rows_i32": [_vp, _i, _i, _vp, _i, _vp, _vp],
    "p2p_palette_relabel_batch": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
}
SPECIAL = {"p2p_last_error": ([], C.c_char_p),
           "p2p_replay_fn_index": ([C.c_char_p], C.c_int), "p2p_replay_fn_nargs": ([_i], C.c_int),
           "p2p_replay": ([_vp, _i], C.c_int), "p2p_version": ([], C.c_int), "p2p_view_halo_pixels": ([], C.c_int),
           "p2p_igemm_stat_slots": ([_i, _i, _i, _i, _i], C.c_int),
           "p2p_brig_ok": ([_i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_igemm_norm_act_ok": ([_i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_brig_stat_slots": ([_i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_igemm_layer_stat_slots": ([_i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_conv_fewin_ok": ([_i, _i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_conv_strip_ok": ([_i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_conv_strip_stat_slots": ([_i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_conv_fewout_ok": ([_i, _i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_wgrad_small_blocks": ([_i, _i, _i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_wgemm_workspace_bytes": ([_i, _i, _i, _i, _i, _i], C.c_longlong),
           "p2p_rgbuv_hist_fwd3_workspace_bytes": ([_i], C.c_longlong),
           "p2p_head_softmax_ok": ([_i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_head_dgrad_ok": ([_i, _i, _i, _i, _i, _i, _i, _i, _i], C.c_int),
           "p2p_head_softmax_workspace_bytes": ([_i, _i], C.c_longlong),
           "p2p_view_colsum_workspace_bytes": ([_i, _i, _i, _i, _i, _TP], C.c_longlong),
           "p2p_weight_prep_task_blocks": ([_i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_longlong)}


class PrepTask(C.Structure):
    """include/p2pgan.h p2p_prep_task"""
    _fields_ = [("w", C.c_void_p), ("wn", C.c_void_p), ("wt", C.c_void_p),
                ("Cg", C.c_int), ("Cd", C.c_int), ("wn_rows", C.c_int), ("wn_cols", C.c_int),
                ("wt_rows", C.c_int), ("wt_cols", C.c_int), ("tiles_g", C.c_int), ("tiles_d", C.c_int),
                ("first_block", C.c_longlong)]

_lib = None


class P2PError(RuntimeError):
    pass


def lib():
    """Loads the shared library once; raises if it has not been built (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P2PError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so); the library must bind to THAT instance -- the
        # device pointers and streams it is handed come from it.  Loading torch first makes the dynamic linker resolve
        # the library's libamdhip64 dependency to the copy that is already mapped (loaded the other way round, a second
        # runtime is initialised and every launch fails with "no ROCm-capable device").
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        for name, (argtypes, restype) in SPECIAL.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = L
    return _lib


_fns = {}


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise P2PError(f"{name} failed (rc={rc}): {lib().p2p_last_error().decode()}")


def exported_symbols():
    return list(SIGNATURES) + list(SPECIAL)
