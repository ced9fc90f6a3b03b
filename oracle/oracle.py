"""ctypes binding of the CPU oracle (oracle/librfid_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
Parity status: unpinned (see oracle/rfid_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librfid_oracle.so")

ORC_MAX_ROUNDS_LOG = 4096
ORC_MAX_MAGN = 4096
DECODE_RN16, DECODE_EPC = 0, 1


class Config(C.Structure):
    _fields_ = [("fixed_q", C.c_int), ("max_num_queries", C.c_int), ("number_unique_tags", C.c_int)]


class ReaderState(C.Structure):
    _fields_ = [
        ("status", C.c_int), ("gen2_logic_status", C.c_int), ("gate_status", C.c_int),
        ("decoder_status", C.c_int),
        ("n_queries_sent", C.c_int), ("cur_inventory_round", C.c_int), ("cur_slot_number", C.c_int),
        ("max_slot_number", C.c_int), ("n_epc_correct", C.c_int),
        ("tag_reads", C.c_int * 256), ("n_unique_tags", C.c_int), ("n_rounds_logged", C.c_int),
        ("unique_tags_round", C.c_int * ORC_MAX_ROUNDS_LOG),
        ("magn_squared", C.c_float * ORC_MAX_MAGN), ("n_magn", C.c_int),
        ("n_samples_to_ungate", C.c_int), ("cfg", Config),
    ]


class ReaderTx(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_data0", "n_data1", "n_pw", "n_cw", "n_delim", "n_trcal", "n_cwquery",
                                       "n_cwack", "n_pdown", "fixed_q", "n_rtcal", "n_rtcal_hi", "n_trcal_hi")] + \
               [("query_bits", C.c_float * 22)]


class Cf(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float)]


class DecodeDump(C.Structure):
    _fields_ = [
        ("type", C.c_int), ("index", C.c_int), ("corr", C.c_float * 15), ("h_est", Cf),
        ("energy", C.c_float * 20), ("T", C.c_float), ("n_bits", C.c_int),
        ("bits", C.c_ubyte * 128), ("crc_ok", C.c_int), ("tag_id", C.c_int),
    ]


DUMP_DTYPE = np.dtype([
    ("type", "<i4"), ("index", "<i4"), ("corr", "<f4", (15,)), ("h_est", "<f4", (2,)),
    ("energy", "<f4", (20,)), ("T", "<f4"), ("n_bits", "<i4"), ("bits", "u1", (128,)),
    ("crc_ok", "<i4"), ("tag_id", "<i4")], align=True)
assert DUMP_DTYPE.itemsize == C.sizeof(DecodeDump), (DUMP_DTYPE.itemsize, C.sizeof(DecodeDump))


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rfid_oracle.c")
    hdr = os.path.join(_HERE, "rfid_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "librfid_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        L.orc_default_config.argtypes = [C.POINTER(Config)]
        L.orc_fir_boxcar25_decim5.argtypes = [vp, C.c_long, vp]
        L.orc_fir_boxcar25_decim5.restype = C.c_long
        L.orc_run_trace.argtypes = [C.POINTER(Config), vp, C.c_long, C.c_int, C.POINTER(ReaderState),
                                    vp, vp, vp, C.c_long]
        L.orc_run_trace.restype = C.c_long
        L.orc_run_decimated.argtypes = L.orc_run_trace.argtypes
        L.orc_run_decimated.restype = C.c_long
        L.orc_stream_new.argtypes = [C.POINTER(Config), C.c_int]
        L.orc_stream_new.restype = vp
        L.orc_stream_free.argtypes = [vp]
        L.orc_stream_state.argtypes = [vp, C.POINTER(ReaderState)]
        L.orc_stream_windows.argtypes = [vp]
        L.orc_stream_windows.restype = C.c_long
        for fn in (L.orc_stream_feed, L.orc_stream_feed_raw):
            fn.argtypes = [vp, vp, C.c_long, vp, vp, vp, C.c_long]
            fn.restype = C.c_long
        L.orc_time_trace.argtypes = [C.POINTER(Config), vp, C.c_long, C.c_int, C.POINTER(C.c_double * 3),
                                     C.POINTER(ReaderState)]
        L.orc_time_trace.restype = C.c_long
        L.orc_time_trace_mt.argtypes = [C.POINTER(Config), vp, C.c_long, C.c_int, C.c_int, C.POINTER(C.c_double),
                                        C.POINTER(C.c_int)]
        L.orc_time_trace_mt.restype = C.c_long
        L.orc_reader_tx_init.argtypes = [C.POINTER(ReaderTx), C.c_int, C.c_int]
        L.orc_reader_work_tx.argtypes = [C.POINTER(ReaderTx), C.POINTER(ReaderState), vp, C.c_int, vp]
        L.orc_reader_work_tx.restype = C.c_int
        L.orc_initialize_reader_state.argtypes = [C.POINTER(ReaderState), C.POINTER(Config)]
        L.orc_print_results.argtypes = [C.POINTER(ReaderState), C.c_char_p, C.c_int]
        L.orc_print_results.restype = C.c_int
        L.orc_check_crc.argtypes = [C.c_char_p, C.c_int]
        L.orc_check_crc.restype = C.c_int
        L.orc_crc16_bytes.argtypes = [C.c_char_p, C.c_int]
        L.orc_crc16_bytes.restype = C.c_uint
        _lib = L
    return _lib


def config(fixed_q: int = 0, max_num_queries: int = 1000, number_unique_tags: int = 100) -> Config:
    return Config(fixed_q, max_num_queries, number_unique_tags)


def fir(raw: np.ndarray) -> np.ndarray:
    raw = np.ascontiguousarray(raw, dtype=np.complex64)
    y = np.empty(len(raw) // 5, dtype=np.complex64)
    lib().orc_fir_boxcar25_decim5(raw.ctypes.data, len(raw), y.ctypes.data)
    return y


class Result:
    def __init__(self, state: ReaderState, dumps: np.ndarray, open_idx: np.ndarray, dc: np.ndarray,
                 n_windows: int):
        self.state = state
        self.dumps = dumps
        self.open_idx = open_idx
        self.dc = dc
        self.n_windows = n_windows

    def stats(self) -> dict:
        s = self.state
        return dict(n_queries_sent=s.n_queries_sent, cur_inventory_round=s.cur_inventory_round,
                    cur_slot_number=s.cur_slot_number, n_epc_correct=s.n_epc_correct,
                    n_unique_tags=s.n_unique_tags, status=s.status,
                    tag_reads={i: s.tag_reads[i] for i in range(256) if s.tag_reads[i]})

    def print_results(self) -> str:
        buf = C.create_string_buffer(1 << 15)
        n = lib().orc_print_results(C.byref(self.state), buf, len(buf))
        return buf.raw[:n].decode()


def _run(fn, cfg: Config, x: np.ndarray, chunk: int, max_dumps: Optional[int]) -> Result:
    x = np.ascontiguousarray(x, dtype=np.complex64)
    if max_dumps is None:
        max_dumps = max(16, len(x) // 200)
    st = ReaderState()
    dumps = np.zeros(max_dumps, dtype=DUMP_DTYPE)
    open_idx = np.zeros(max_dumps, dtype=np.int64)
    dc = np.zeros(max_dumps, dtype=np.complex64)
    n = fn(C.byref(cfg), x.ctypes.data, len(x), chunk, C.byref(st), dumps.ctypes.data,
           open_idx.ctypes.data, dc.ctypes.data, max_dumps)
    k = min(n, max_dumps)
    return Result(st, dumps[:k], open_idx[:k], dc[:k], n)


def run_trace(raw: np.ndarray, cfg: Optional[Config] = None, chunk: int = 4096,
              max_dumps: Optional[int] = None) -> Result:
    """Full chain FIR -> gate -> decoder (+reader transitions) on a raw 2 Msps trace."""
    return _run(lib().orc_run_trace, cfg or config(), raw, chunk, max_dumps)


def run_decimated(y: np.ndarray, cfg: Optional[Config] = None, chunk: int = 4096,
                  max_dumps: Optional[int] = None) -> Result:
    return _run(lib().orc_run_decimated, cfg or config(), y, chunk, max_dumps)


class Stream:
    """The harness as a resumable object: feed a trace in pieces (raw 2 Msps or decimated samples); windows,
    dumps and the reader state accumulate exactly as in one run_trace() call over the concatenation."""

    def __init__(self, cfg: Optional[Config] = None, chunk: int = 4096):
        self.cfg = cfg or config()
        self._h = lib().orc_stream_new(C.byref(self.cfg), chunk)
        self._dumps, self._open, self._dc = [], [], []

    def _feed(self, fn, x: np.ndarray, keep: bool):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        cap = max(16, len(x) // 40)
        dumps = np.zeros(cap, dtype=DUMP_DTYPE)
        open_idx = np.zeros(cap, dtype=np.int64)
        dc = np.zeros(cap, dtype=np.complex64)
        n = fn(self._h, x.ctypes.data, len(x), dumps.ctypes.data, open_idx.ctypes.data, dc.ctypes.data, cap)
        assert n <= cap
        if keep:
            self._dumps.append(dumps[:n]); self._open.append(open_idx[:n]); self._dc.append(dc[:n])
        return n

    def feed_raw(self, x: np.ndarray, keep: bool = True) -> int:
        return self._feed(lib().orc_stream_feed_raw, x, keep)

    def feed_decimated(self, y: np.ndarray, keep: bool = True) -> int:
        return self._feed(lib().orc_stream_feed, y, keep)

    def result(self) -> Result:
        st = ReaderState()
        lib().orc_stream_state(self._h, C.byref(st))
        cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
        return Result(st, cat(self._dumps, DUMP_DTYPE), cat(self._open, np.int64), cat(self._dc, np.complex64),
                      int(lib().orc_stream_windows(self._h)))

    def close(self) -> None:
        if self._h:
            lib().orc_stream_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def time_trace(raw: np.ndarray, reps: int = 1, cfg: Optional[Config] = None):
    raw = np.ascontiguousarray(raw, dtype=np.complex64)
    st = ReaderState()
    secs = (C.c_double * 3)()
    cfg = cfg or config()
    nw = lib().orc_time_trace(C.byref(cfg), raw.ctypes.data, len(raw), reps, C.byref(secs), C.byref(st))
    return dict(fir_s=secs[0], gate_decoder_s=secs[1], total_s=secs[2], windows=nw,
                n_epc_correct=st.n_epc_correct)


def time_trace_mt(raw: np.ndarray, reps: int, nthreads: int, cfg: Optional[Config] = None):
    """`nthreads` independent single-thread runs of time_trace() at once -> wall time."""
    raw = np.ascontiguousarray(raw, dtype=np.complex64)
    wall = C.c_double(0.0)
    nepc = C.c_int(0)
    cfg = cfg or config()
    nw = lib().orc_time_trace_mt(C.byref(cfg), raw.ctypes.data, len(raw), reps, nthreads, C.byref(wall), C.byref(nepc))
    return dict(wall_s=wall.value, windows=nw, n_epc_correct=nepc.value, threads=nthreads, reps=reps)


class ReaderTxSim:
    """reader_impl::general_work incl. the TX waveform (orc_reader_work_tx), stepping a READER_STATE."""

    def __init__(self, dac_rate: int = 1000000, cfg: Optional[Config] = None):
        self.cfg = cfg or config()
        self.tx = ReaderTx()
        lib().orc_reader_tx_init(C.byref(self.tx), int(dac_rate), int(self.cfg.fixed_q))
        self.state = ReaderState()
        lib().orc_initialize_reader_state(C.byref(self.state), C.byref(self.cfg))

    def work(self, in_bits=None) -> np.ndarray:
        bits = np.ascontiguousarray(in_bits if in_bits is not None else [], dtype=np.float32)
        out = np.zeros(16384, dtype=np.float32)
        n = lib().orc_reader_work_tx(C.byref(self.tx), C.byref(self.state), bits.ctypes.data if len(bits) else None,
                                     len(bits), out.ctypes.data)
        return out[:n].copy()
