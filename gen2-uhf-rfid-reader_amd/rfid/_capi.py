"""ctypes binding of librfid_mi355x.so (include/rfid_mi355x.h).

The library is the product: there is no Python or CPU implementation of the receive
path behind it.  If the shared object is missing or no gfx950 GPU is usable, loading /
context creation raises RfidError -- nothing falls back.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.path.join(PKG_ROOT, "lib", "librfid_mi355x.so")
HEADER_PATH = os.path.join(REPO_ROOT, "include", "rfid_mi355x.h")

# enums (include/rfid_mi355x.h; numeric values of gr-rfid/include/rfid/global_vars.h:31-34)
RUNNING, TERMINATED = 0, 1
(SEND_QUERY, SEND_ACK, SEND_QUERY_REP, IDLE, SEND_CW, START, SEND_QUERY_ADJUST, SEND_NAK_QR,
 SEND_NAK_Q, POWER_DOWN) = range(10)
GATE_OPEN, GATE_CLOSED, GATE_SEEK_RN16, GATE_SEEK_EPC = range(4)
DECODE_RN16, DECODE_EPC = 0, 1

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_CAPACITY, ERR_STATE = 0, -1, -2, -3, -4, -5, -6


class RfidError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"rfid_mi355x status {status}: {msg}")
        self.status = status


class Params(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("decim", C.c_int32), ("n_taps", C.c_int32),
                ("fixed_q", C.c_int32), ("max_num_queries", C.c_int32), ("number_unique_tags", C.c_int32)]


class ReaderState(C.Structure):
    _fields_ = [("status", C.c_int32), ("gen2_logic_status", C.c_int32), ("gate_status", C.c_int32),
                ("decoder_status", C.c_int32), ("n_samples_to_ungate", C.c_int32),
                ("n_queries_sent", C.c_int32), ("cur_inventory_round", C.c_int32),
                ("cur_slot_number", C.c_int32), ("max_slot_number", C.c_int32), ("n_epc_correct", C.c_int32),
                ("n_unique_tags", C.c_int32), ("tag_reads", C.c_int32 * 256)]


class SynthGen2Params(C.Structure):
    _fields_ = [("leak_re", C.c_float), ("leak_im", C.c_float), ("h_re", C.c_float * 16), ("h_im", C.c_float * 16),
                ("n_tags", C.c_int32), ("tail_us", C.c_int32)]


class LsReport(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("pieces", "units", "chunk", "avg_rounds", "avg_reruns", "fsm_rounds", "dc_rounds",
                                         "dc_reruns", "verified", "gave_up", "cuts_dropped", "windows", "dc_finished")]


class BatchTiming(C.Structure):
    _fields_ = [("mf_ms", C.c_float), ("gate_ms", C.c_float), ("decode_ms", C.c_float),
                ("stats_ms", C.c_float), ("total_ms", C.c_float), ("front_ms", C.c_float),
                ("front_chunks", C.c_int32), ("decode_launches", C.c_int32),
                ("fused_front", C.c_int32), ("reserved_", C.c_int32)]


WINDOW_DTYPE = np.dtype([("stream", "<i4"), ("seq", "<i4"), ("start", "<i4"), ("type", "<i4"),
                         ("dc_re", "<f4"), ("dc_im", "<f4")])
RESULT_DTYPE = np.dtype([("type", "<i4"), ("index", "<i4"), ("h_re", "<f4"), ("h_im", "<f4"), ("T", "<f4"),
                         ("bits", "<u4", (4,)), ("n_bits", "<i4"), ("crc_ok", "<i4"), ("tag_id", "<i4")])
SCORES_DTYPE = np.dtype([("corr", "<f4", (15,)), ("energy", "<f4", (20,)), ("pad_", "<f4")])
STATS_DTYPE = np.dtype([("n_queries_sent", "<i4"), ("cur_inventory_round", "<i4"), ("cur_slot_number", "<i4"),
                        ("n_epc_correct", "<i4"), ("n_unique_tags", "<i4"), ("n_windows", "<i4"),
                        ("n_windows_used", "<i4"), ("status", "<i4"), ("tag_reads", "<i4", (256,))])
STREAM_WINDOW_DTYPE = np.dtype([("start", "<i8"), ("type", "<i4"), ("reserved_", "<i4"), ("dc_re", "<f4"), ("dc_im", "<f4")])
assert STREAM_WINDOW_DTYPE.itemsize == 24
assert WINDOW_DTYPE.itemsize == 24 and RESULT_DTYPE.itemsize == 48
assert SCORES_DTYPE.itemsize == 144 and STATS_DTYPE.itemsize == 1056

# every symbol include/rfid_mi355x.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _ip = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_int)
SIGNATURES = {
    "rfid_params_default": (_i, [C.POINTER(Params)]),
    "rfid_ctx_create": (_i, [C.POINTER(Params), _i, C.POINTER(_vp)]),
    "rfid_ctx_destroy": (_i, [_vp]),
    "rfid_ctx_reset": (_i, [_vp]),
    "rfid_strerror": (C.c_char_p, [_i]),
    "rfid_last_error": (C.c_char_p, [_vp]),
    "rfid_version": (C.c_char_p, []),
    "rfid_selftest": (_i, [_vp, _ip]),
    "rfid_mf_work": (_i, [_vp, _vp, _i, _vp, _i, _ip]),
    "rfid_gate_work": (_i, [_vp, _vp, _i, _vp, _i, _ip, _ip]),
    "rfid_decoder_work": (_i, [_vp, _vp, _i, _vp, _i, _ip, _ip, _vp, _vp]),
    "rfid_reader_work": (_i, [_vp, _i, _ip]),
    "rfid_reader_work_tx": (_i, [_vp, _i, _vp, _i, _vp, _i, _ip, _ip]),
    "rfid_reader_tx_max": (_i, [_i]),
    "rfid_get_state": (_i, [_vp, C.POINTER(ReaderState)]),
    "rfid_print_results": (_i, [_vp, C.c_char_p, _i, _ip]),
    "rfid_synth_gen2_size": (_i, [_vp, _vp, _i64, C.POINTER(C.c_int64)]),
    "rfid_synth_gen2": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, C.c_float, C.c_uint64, _i64, C.POINTER(C.c_int64)]),
    "rfid_stream_begin": (_i, [_vp, _i64]),
    "rfid_stream_staging": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "rfid_stream_work": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _i64, C.POINTER(C.c_int64)]),
    "rfid_stream_end": (_i, [_vp]),
    "rfid_lookahead_enable": (_i, [_vp, _i64]),
    "rfid_lookahead_enable_gate": (_i, [_vp, _i64]),
    "rfid_lookahead_pending": (_i, [_vp, _ip, _ip]),
    "rfid_lookahead_set_coalesce": (_i, [_vp, _i64]),
    "rfid_lookahead_drain": (_i, [_vp]),
    "rfid_lookahead_set_late_outputs": (_i, [_vp, _i]),
    "rfid_mf_pending": (_i, [_vp, _ip]),
    "rfid_mf_must_fetch": (_i, [_vp, _ip]),
    "rfid_lookahead_set_consume_ahead": (_i, [_vp, _i]),
    "rfid_gate_forecast": (_i, [_vp, _i, _ip]),
    "rfid_lookahead_set_scheduler": (_i, [_vp, _i64]),
    "rfid_abi_version": (_i, []),
    "rfid_lookahead_flush": (_i, [_vp]),
    "rfid_host_alloc": (_vp, [C.c_size_t]),
    "rfid_host_free": (None, [_vp]),
    "rfid_gate_magn_squared": (_i, [_vp, _vp, _i, C.POINTER(_i)]),
    "rfid_batch_get_gated": (_i, [_vp, _i, _i, _vp, _i64, C.POINTER(C.c_int64)]),
    "rfid_batch_plan": (_i, [_vp, _i, _i64]),
    "rfid_batch_set_streams": (_i, [_vp, _i]),
    "rfid_batch_set_long_stream": (_i, [_vp, _i]),
    "rfid_batch_ls_report": (_i, [_vp, _vp]),
    "rfid_ctx_set_knob": (_i, [_vp, C.c_char_p, _i]),
    "rfid_ctx_get_knob": (_i, [_vp, C.c_char_p, _ip]),
    "rfid_batch_mf": (_i, [_vp, _vp, _i64, _i64, _vp]),
    "rfid_batch_gate": (_i, [_vp]),
    "rfid_batch_decode": (_i, [_vp, _i]),
    "rfid_batch_stats": (_i, [_vp]),
    "rfid_batch_process": (_i, [_vp, _vp, _i64, _i64, _vp, _i]),
    "rfid_batch_sync": (_i, [_vp]),
    "rfid_batch_timing_get": (_i, [_vp, C.POINTER(BatchTiming)]),
    "rfid_batch_get_stats": (_i, [_vp, _vp, _i]),
    "rfid_batch_get_windows": (_i, [_vp, _vp, _vp, _vp, _i64, C.POINTER(_i64)]),
    "rfid_batch_device_ptrs": (_i, [_vp, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_vp)]),
    "rfid_batch_get_mf": (_i, [_vp, _i, _vp, _i64, C.POINTER(_i64)]),
    "rfid_ctx_stream": (_vp, [_vp]),
    "rfid_synth_replicas": (_i, [_vp, _vp, _i64, _vp, _i64, _i, C.c_float, C.c_uint64, _i64]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load librfid_mi355x.so and bind every declared symbol.  Raises RfidError when the
    library has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RfidError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: the HIP library has not been built; "
                                       "there is no CPU fallback")
    # Callers own HBM through PyTorch, whose wheel bundles its own HIP runtime (same soname as /opt/rocm's):
    # whichever copy is loaded first serves the whole process, and torch finds no device when it comes second
    # (measured on the MI355X box).  So let torch load and initialise its runtime first when it is installed.
    try:
        import torch
        torch.cuda.is_available()
    except ImportError:
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise RfidError(ERR_NO_DEVICE, f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, ctx: Optional[int] = None) -> None:
    if status != OK:
        lib = load()
        msg = lib.rfid_strerror(status).decode()
        if ctx:
            detail = lib.rfid_last_error(ctx).decode()
            if detail:
                msg += f" ({detail})"
        raise RfidError(status, msg)


def default_params(**overrides) -> Params:
    p = Params()
    check(load().rfid_params_default(C.byref(p)))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown rfid_params field {k!r}")
        setattr(p, k, int(v))
    return p
