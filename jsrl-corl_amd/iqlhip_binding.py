"""ctypes binding of libiqlhip.so (C ABI: include/iqlhip.h).

There is deliberately NO fallback: if the shared library is missing or a call
fails, an exception is raised.  The product path never routes through the
oracle or through PyTorch arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IQLHIP_LIB", os.path.join(_HERE, "libiqlhip.so"))  # IQLHIP_LIB: diagnostic builds

IQLHIP_HIDDEN = 256
IQLHIP_ACT_ROWS = 4096      # rows per iqlhip_actor_forward call (include/iqlhip.h)
IQLHIP_MAX_WORLD = 8
IQLHIP_GRAPH_STEPS = 64
IQLHIP_UNIQUE_ID_BYTES = 128
IQLHIP_IPC_HANDLE_BYTES = 64
XCH_NONE, XCH_RCCL, XCH_P2P = 0, 1, 2
TS_CONTINUE = 1
NET_V, NET_Q1, NET_Q2, NET_PI = 0, 1, 2, 3
POLICY_GAUSSIAN, POLICY_DETERMINISTIC = 0, 1

E_INVAL, E_HIP, E_NOTBOUND, E_UNSUPPORTED, E_INDEX = -1, -2, -3, -4, -5


class Dims(C.Structure):
    _fields_ = [("state_dim", C.c_int32), ("action_dim", C.c_int32), ("hidden_dim", C.c_int32),
                ("n_hidden", C.c_int32), ("policy", C.c_int32), ("max_batch", C.c_int32)]


class NetLayout(C.Structure):
    _fields_ = [("seg_begin", C.c_int64), ("seg_end", C.c_int64),
                ("w0", C.c_int64), ("b0", C.c_int64), ("w1", C.c_int64), ("b1", C.c_int64),
                ("w2", C.c_int64), ("b2", C.c_int64), ("log_std", C.c_int64),
                ("k_in", C.c_int32), ("d_out", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [("net", NetLayout * 4), ("n_params", C.c_int64), ("n_target", C.c_int64),
                ("target_src", C.c_int64)]


class Hyper(C.Structure):
    _fields_ = [("iql_tau", C.c_float), ("beta", C.c_float), ("discount", C.c_float), ("tau", C.c_float),
                ("one_minus_tau", C.c_float), ("exp_adv_max", C.c_float), ("log_std_min", C.c_float),
                ("log_std_max", C.c_float)]


class StepScalars(C.Structure):
    _fields_ = [("step_size", C.c_float * 3), ("bc2_sqrt", C.c_float * 3), ("beta2", C.c_float),
                ("one_minus_beta1", C.c_float), ("one_minus_beta2", C.c_float), ("eps", C.c_float),
                ("grad_scale", C.c_float), ("inv_batch", C.c_float)]


class Batch(C.Structure):
    _fields_ = [("s_dev", C.c_void_p), ("a_dev", C.c_void_p), ("r_dev", C.c_void_p), ("ns_dev", C.c_void_p),
                ("d_dev", C.c_void_p),
                ("ld_s", C.c_int64), ("ld_a", C.c_int64), ("ld_r", C.c_int64), ("ld_ns", C.c_int64),
                ("ld_d", C.c_int64),
                ("idx_dev", C.c_void_p), ("rows", C.c_int32)]


# every symbol include/iqlhip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("iqlhip_version", C.c_int, []),
    ("iqlhip_last_error", C.c_char_p, []),
    ("iqlhip_arena_layout", C.c_int, [C.POINTER(Dims), C.POINTER(Layout)]),
    ("iqlhip_create", C.c_int, [C.POINTER(Dims), C.POINTER(Hyper), C.c_int, C.POINTER(C.c_void_p)]),
    ("iqlhip_destroy", C.c_int, [C.c_void_p]),
    ("iqlhip_set_hyper", C.c_int, [C.c_void_p, C.POINTER(Hyper)]),
    ("iqlhip_bind", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("iqlhip_set_precision", C.c_int, [C.c_void_p, C.c_int]),
    ("iqlhip_set_dropout", C.c_int, [C.c_void_p, C.c_float, C.c_uint64]),
    ("iqlhip_get_counters", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("iqlhip_set_counters", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("iqlhip_debug_write_masks", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ("iqlhip_step", C.c_int, [C.c_void_p, C.POINTER(Batch), C.POINTER(StepScalars), C.c_void_p]),
    ("iqlhip_step_sync", C.c_int, [C.c_void_p, C.POINTER(Batch), C.POINTER(StepScalars), C.POINTER(C.c_float), C.c_void_p]),
    ("iqlhip_step_begin", C.c_int, [C.c_void_p, C.POINTER(Batch), C.POINTER(StepScalars), C.c_void_p]),
    ("iqlhip_step_wait", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]),
    ("iqlhip_online_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.POINTER(StepScalars), C.POINTER(C.c_float), C.c_void_p, C.c_float,
                                     C.c_uint64, C.c_void_p, C.c_void_p]),
    ("iqlhip_forward_backward", C.c_int, [C.c_void_p, C.POINTER(Batch), C.POINTER(StepScalars), C.c_void_p, C.c_void_p]),
    ("iqlhip_apply_update", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(StepScalars), C.c_void_p]),
    ("iqlhip_grad_words", C.c_int64, [C.c_void_p]),
    ("iqlhip_train_steps", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                     C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p]),
    ("iqlhip_train_steps_prepare", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    ("iqlhip_comm_unique_id", C.c_int, [C.c_void_p]),
    ("iqlhip_allreduce_init", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("iqlhip_p2p_export", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    ("iqlhip_p2p_attach", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("iqlhip_xch_select", C.c_int, [C.c_void_p, C.c_int]),
    ("iqlhip_xch_status", C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    ("iqlhip_xch_clear_status", C.c_int, [C.c_void_p, C.c_void_p]),
    ("iqlhip_xch_shutdown", C.c_int, [C.c_void_p]),
    ("iqlhip_read_losses", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_void_p]),
    ("iqlhip_read_loss_ring", C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_void_p]),
    ("iqlhip_row_stride", C.c_int64, [C.c_int32, C.c_int32]),
    ("iqlhip_rows_write", C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("iqlhip_rows_fill_synth", C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_uint64,
                                         C.c_float, C.c_int32, C.c_void_p]),
    ("iqlhip_rows_gather", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("iqlhip_actor_forward", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_float,
                                       C.c_void_p, C.c_int64, C.c_void_p]),
    ("iqlhip_rows_gather_packed", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("iqlhip_actor_sample", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_uint64, C.c_float, C.c_void_p,
                                      C.c_int64, C.c_void_p]),
    ("iqlhip_stream_synchronize", C.c_int, [C.c_void_p]),
    ("iqlhip_rows_gather_packed_h", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                              C.c_void_p]),
    ("iqlhip_rows_sample_packed", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("iqlhip_cols_mean_std", C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    ("iqlhip_rows_normalize", C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    ("iqlhip_draw_indices", C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p]),
    ("iqlhip_debug_read", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int64,
                                    C.POINTER(C.c_int64), C.c_void_p]),
    ("iqlhip_debug_time_kernel", C.c_int, [C.c_void_p, C.POINTER(Batch), C.c_int, C.c_int, C.POINTER(C.c_float),
                                           C.c_void_p]),
    ("iqlhip_debug_drain_spin", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    ("iqlhip_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("iqlhip_get_timing", C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
]

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libiqlhip.so (built by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the IQL step.")
        # torch first: libiqlhip.so needs libamdhip64; loading it before torch would bind the system ROCm
        # runtime while torch brings its own copy — two HIP runtimes in one process, the first blind to the GPU.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)     # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error() -> str:
    return (lib().iqlhip_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map C status codes to the exception types the reference raises."""
    if rc == 0:
        return
    msg = f"iqlhip: {last_error()} (code {rc})"
    if rc == E_INVAL:
        raise ValueError(msg)
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == E_INDEX:
        raise IndexError(msg)
    raise RuntimeError(msg)


def arena_layout(state_dim: int, action_dim: int, gaussian: bool = True, max_batch: int = 256,
                 hidden_dim: int = IQLHIP_HIDDEN, n_hidden: int = 2) -> Layout:
    d = Dims(state_dim, action_dim, hidden_dim, n_hidden,
             POLICY_GAUSSIAN if gaussian else POLICY_DETERMINISTIC, max_batch)
    out = Layout()
    check(lib().iqlhip_arena_layout(C.byref(d), C.byref(out)))
    return out


def row_stride(state_dim: int, action_dim: int) -> int:
    return int(lib().iqlhip_row_stride(state_dim, action_dim))
