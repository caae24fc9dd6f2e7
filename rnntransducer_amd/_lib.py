"""ctypes binding of librnnt_hip.so (C ABI: include/rnnt_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.  The product path
never routes through torch ops or the CPU oracle for the kernels declared in the header.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RNNT_HIP_LIB: another build of the same library (A/B variants made by `csrc/build.py --variant`); still no fallback of any kind
LIB_PATH = os.environ.get("RNNT_HIP_LIB") or os.path.join(_HERE, "csrc", "librnnt_hip.so")
ABI_VERSION = 4   # RNNT_HIP_ABI_VERSION of include/rnnt_hip.h

GEMM_GELU_A, GEMM_GELU_B, GEMM_ACCUM, GEMM_MUL_DGELU, GEMM_EXACT_F32 = 1, 2, 4, 8, 16
CELL_LSTM, CELL_GRU, CELL_RNN_TANH, CELL_RNN_RELU = 0, 1, 2, 3

c_f32p = C.c_void_p
c_i64 = C.c_int64
c_i32 = C.c_int32


class GemmDesc(C.Structure):
    _fields_ = [("M", c_i64), ("N", c_i64), ("K", c_i64), ("A", C.c_void_p), ("a_div", c_i64), ("a_so", c_i64),
                ("a_si", c_i64), ("a_sk", c_i64), ("a_mc", c_i32), ("a_rowidx", C.c_void_p), ("B", C.c_void_p),
                ("b_sn", c_i64), ("b_sk", c_i64), ("C", C.c_void_p), ("c_div", c_i64), ("c_so", c_i64),
                ("c_si", c_i64), ("bias", C.c_void_p), ("aux", C.c_void_p), ("flags", C.c_uint32),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class LstmDesc(C.Structure):
    _fields_ = [("T", c_i32), ("B", c_i32), ("I", c_i32), ("H", c_i32), ("D", c_i32), ("cell", c_i32), ("lens", C.c_void_p),
                ("x", C.c_void_p), ("x_st", c_i64), ("x_sb", c_i64), ("w_ih", C.c_void_p * 2),
                ("w_hh", C.c_void_p * 2), ("b_ih", C.c_void_p * 2), ("b_hh", C.c_void_p * 2), ("y", C.c_void_p),
                ("y_drop", C.c_void_p), ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
                ("gates", C.c_void_p), ("cst", C.c_void_p), ("aux", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("status", C.c_void_p), ("x_abs_bound", C.c_float), ("row_idx", C.c_void_p),
                ("n_rows", c_i32)]


class LstmBwdDesc(C.Structure):
    _fields_ = [("f", LstmDesc), ("dy", C.c_void_p), ("dx", C.c_void_p), ("dw_ih", C.c_void_p * 2),
                ("dw_hh", C.c_void_p * 2), ("db", C.c_void_p * 2), ("db_hh", C.c_void_p * 2), ("accumulate", c_i32), ("phase", c_i32),
                ("beside_recurrence", c_i32)]


class HpProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("a_amax", C.c_void_p), ("B", C.c_void_p), ("b_amax", C.c_void_p), ("M", c_i64), ("N", c_i64),
                ("K", c_i64), ("C", C.c_void_p), ("ldc", c_i64), ("flags", C.c_uint32)]


DECODE_MAX_LAYERS = 8


class DecodeDesc(C.Structure):
    _fields_ = [("T", c_i32), ("B", c_i32), ("V", c_i32), ("Hp", c_i32), ("O", c_i32), ("L", c_i32), ("cell", c_i32),
                ("blank", c_i32), ("max_iters", c_i32), ("max_out", c_i32), ("A", C.c_void_p), ("t_lens", C.c_void_p), ("emb", C.c_void_p),
                ("w_ih", C.c_void_p * DECODE_MAX_LAYERS), ("w_hh", C.c_void_p * DECODE_MAX_LAYERS),
                ("b_ih", C.c_void_p * DECODE_MAX_LAYERS), ("b_hh", C.c_void_p * DECODE_MAX_LAYERS),
                ("w_o", C.c_void_p), ("b_o", C.c_void_p), ("w_d", C.c_void_p), ("ld_d", c_i64),
                ("tokens", C.c_void_p), ("ntok", C.c_void_p)]


class PrednetStepDesc(C.Structure):
    _fields_ = [("B", c_i32), ("Hp", c_i32), ("L", c_i32), ("cell", c_i32), ("tokens", C.c_void_p), ("emb", C.c_void_p),
                ("w_ih", C.c_void_p * DECODE_MAX_LAYERS), ("w_hh", C.c_void_p * DECODE_MAX_LAYERS),
                ("b_ih", C.c_void_p * DECODE_MAX_LAYERS), ("b_hh", C.c_void_p * DECODE_MAX_LAYERS),
                ("h_in", C.c_void_p), ("c_in", C.c_void_p), ("h_out", C.c_void_p), ("c_out", C.c_void_p)]


# every symbol include/rnnt_hip.h declares: (name, restype, argtypes)
SYMBOLS = {
    "rnnt_hip_version": (C.c_int, []),
    "rnnt_hip_last_error": (C.c_char_p, []),
    "rnnt_hip_device_cus": (C.c_int, []),
    "rnnt_hip_prof_enable": (C.c_int, [C.c_int]),
    "rnnt_hip_prof_collect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "rnnt_hip_gemm_workspace_bytes": (C.c_size_t, [c_i64, c_i64, c_i64]),
    "rnnt_hip_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "rnnt_hip_hp_bytes": (C.c_size_t, [c_i64, c_i64]),
    "rnnt_hip_hp_split": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, c_i32, c_i64, c_i64, C.c_void_p, C.c_void_p, c_i32, C.c_void_p]),
    "rnnt_hip_gemm_hp_workspace_bytes": (C.c_size_t, [c_i64, c_i64, c_i64]),
    "rnnt_hip_gemm_hp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p, c_i64, C.c_void_p,
                                  C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnnt_hip_hp_split_both": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rnnt_hip_gemm_hp_grouped_workspace_bytes": (C.c_size_t, [C.POINTER(HpProblem), c_i32]),
    "rnnt_hip_gemm_hp_grouped": (C.c_int, [C.POINTER(HpProblem), c_i32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnnt_hip_lstm_workspace_bytes": (C.c_size_t, [c_i32] * 5),
    "rnnt_hip_lstm_max_batch": (c_i32, [c_i32, c_i32, c_i32]),
    "rnnt_hip_lstm_free_xcds": (c_i32, [c_i32] * 5),
    "rnnt_hip_lstm_takes_row_idx": (c_i32, [c_i32] * 6),
    "rnnt_hip_lstm_fwd": (C.c_int, [C.POINTER(LstmDesc), C.c_void_p]),
    "rnnt_hip_lstm_bwd": (C.c_int, [C.POINTER(LstmBwdDesc), C.c_void_p]),
    "rnnt_hip_lstm_check": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rnnt_hip_lstm_debug_read": (C.c_int, [C.c_void_p, c_i32, c_i32, c_i32, c_i32, c_i32, C.c_void_p, c_i32, C.c_void_p]),
    "rnnt_hip_joint_loss_workspace_bytes": (C.c_size_t, [c_i32] * 4),
    "rnnt_hip_joint_loss_fwd_bwd": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, c_i32, c_i32, c_i32, c_i32, c_i32, C.c_float,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnnt_hip_joint_loss_bwd": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, c_i32, c_i32, c_i32, c_i32, c_i32, C.c_float, C.c_void_p, c_i32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnnt_hip_scaled_sum_f32": (C.c_int, [C.c_void_p, c_i32, C.c_float, C.c_void_p, C.c_void_p]),
    "rnnt_hip_joint_logits_fwd": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_void_p, c_i64, c_i64, C.c_void_p, c_i32, c_i32,
                                             c_i32, c_i32, C.c_void_p, C.c_void_p]),
    "rnnt_hip_loss_from_logits_fwd_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i32, c_i32, c_i32,
                                                     c_i32, c_i32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_size_t, C.c_void_p]),
    "rnnt_hip_loss_from_logits_fwd_bwd_ex": (C.c_int, [C.c_void_p, c_i32, C.c_void_p, C.c_void_p, C.c_void_p, c_i32, c_i32,
                                                        c_i32, c_i32, c_i32, C.c_float, C.c_void_p, C.c_void_p,
                                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnnt_hip_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, c_i64, C.c_void_p]),
    "rnnt_hip_adamw_step_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_float, c_i64, C.c_float, C.c_void_p, C.c_void_p]),
    "rnnt_hip_embedding_fwd": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i32, c_i32, C.c_void_p, C.c_void_p]),
    "rnnt_hip_colsum_workspace_bytes": (C.c_size_t, [c_i64, c_i64]),
    "rnnt_hip_colsum_f32": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnnt_hip_colsum_f32_acc": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rnnt_hip_embedding_bwd": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i32, c_i32, c_i64, C.c_void_p, C.c_void_p]),
    "rnnt_hip_embedding_bwd_acc": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i32, c_i32, c_i64, C.c_void_p, C.c_void_p]),
    "rnnt_hip_greedy_decode": (C.c_int, [C.POINTER(DecodeDesc), C.c_void_p]),
    "rnnt_hip_prednet_step": (C.c_int, [C.POINTER(PrednetStepDesc), C.c_void_p]),
    "rnnt_hip_frontend_norm_pad": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, c_i32, c_i32, c_i64, c_i32, C.c_void_p, C.c_void_p]),
    "rnnt_hip_power_mel_log1p": (C.c_int, [C.c_void_p, c_i64, c_i32, C.c_void_p, c_i32, C.c_void_p, c_i32, C.c_void_p, C.c_void_p]),
}

KERNEL_KINDS = ["gemm_f32_kernel", "lstm_fwd_kernel", "lstm_bwd_kernel", "lse_kernel", "alphabeta_kernel",
                "lattice_grad_kernel", "misc", "gemm_hp_kernel", "hp_split_kernels"]

_lib = None


class RnntHipError(RuntimeError):
    pass


def lib():
    """Loads librnnt_hip.so or raises.  No silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RnntHipError(
                f"{LIB_PATH} not found: build it with `python -m rnntransducer_amd.csrc.build` "
                "(or __graft_entry__.build()).  rnntransducer_amd has no non-HIP execution path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.rnnt_hip_version() != ABI_VERSION:   # a stale build would read the descriptors with the wrong layout
            raise RnntHipError(f"{LIB_PATH} reports ABI version {handle.rnnt_hip_version()}, this package binds version {ABI_VERSION}: "
                               "rebuild it with `python -m rnntransducer_amd.csrc.build`")
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().rnnt_hip_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise RnntHipError(f"{what}: rc={rc}: {msg}")
