"""rfid.Context -- one RX stream / one batch workspace on one MI355X.

Thin object wrapper over the C-ABI (include/rfid_mi355x.h).  All sample arithmetic runs in
the HIP kernels behind it.  Device buffers are passed as raw pointers; torch (or any other
allocator) is only used by callers to own HBM.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _capi as capi


def _c64(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a


class Context:
    def __init__(self, device: int = 0, **params):
        self._lib = capi.load()
        self.params = capi.default_params(**params)
        h = C.c_void_p()
        capi.check(self._lib.rfid_ctx_create(C.byref(self.params), int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self._planned: Optional[Tuple[int, int]] = None
        self._active = 0

    # -- lifetime ------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.rfid_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, status: int) -> None:
        capi.check(status, self._h)

    def reset(self) -> None:
        self._chk(self._lib.rfid_ctx_reset(self._h))

    def selftest(self) -> int:
        n = C.c_int(0)
        self._chk(self._lib.rfid_selftest(self._h, C.byref(n)))
        return n.value

    @property
    def stream_handle(self) -> int:
        return int(self._lib.rfid_ctx_stream(self._h) or 0)

    # -- READER_STATE ----------------------------------------------------------------------
    def state(self) -> capi.ReaderState:
        st = capi.ReaderState()
        self._chk(self._lib.rfid_get_state(self._h, C.byref(st)))
        return st

    def stats(self) -> dict:
        s = self.state()
        return dict(n_queries_sent=s.n_queries_sent, cur_inventory_round=s.cur_inventory_round,
                    cur_slot_number=s.cur_slot_number, n_epc_correct=s.n_epc_correct,
                    n_unique_tags=s.n_unique_tags, status=s.status,
                    tag_reads={i: s.tag_reads[i] for i in range(256) if s.tag_reads[i]})

    def print_results(self) -> str:
        buf = C.create_string_buffer(1 << 15)
        n = C.c_int(0)
        self._chk(self._lib.rfid_print_results(self._h, buf, len(buf), C.byref(n)))
        return buf.raw[: n.value].decode()

    # -- (1) streaming, host buffers ---------------------------------------------------------
    def mf_work(self, x, out_cap: Optional[int] = None) -> np.ndarray:
        """rfid_mf_work.  With late outputs (lookahead_set_late_outputs) the call returns what the call before it made;
        out_cap: the room the caller's output buffer has (default: this call's outputs, or everything held back)."""
        x = _c64(x)
        if out_cap is None:
            out_cap = max(len(x) // 5 + 2, self.mf_pending())
        out = np.empty(max(int(out_cap), 1), dtype=np.complex64)
        n = C.c_int(0)
        self._chk(self._lib.rfid_mf_work(self._h, x.ctypes.data if len(x) else None, len(x), out.ctypes.data, int(out_cap), C.byref(n)))
        return out[: n.value].copy()

    def lookahead_set_late_outputs(self, on: bool = True) -> None:
        """mf_work returns the filter outputs of the call before it (no call waits for the device); mf_work(empty) fetches
        what is held back at the end of the input."""
        self._chk(self._lib.rfid_lookahead_set_late_outputs(self._h, 1 if on else 0))

    def mf_pending(self) -> int:
        n = C.c_int(0)
        self._chk(self._lib.rfid_mf_pending(self._h, C.byref(n)))
        return n.value

    def gate_work(self, x, out_cap: Optional[int] = None) -> Tuple[int, np.ndarray]:
        """-> (consumed, gated samples), as gate_impl::general_work's consume_each()/output.  out_cap: the room of the caller's
        output buffer (default: as many items as the call is shown; a consume-ahead gate may be called without input)."""
        x = _c64(x)
        if out_cap is None:
            out_cap = max(len(x), 1)
        out = np.empty(max(int(out_cap), 1), dtype=np.complex64)
        cons, wr = C.c_int(0), C.c_int(0)
        self._chk(self._lib.rfid_gate_work(self._h, x.ctypes.data if len(x) else None, len(x), out.ctypes.data, int(out_cap),
                                           C.byref(cons), C.byref(wr)))
        return cons.value, out[: wr.value].copy()

    def lookahead_set_consume_ahead(self, on: bool = True) -> None:
        """gate_work consumes everything it is shown; the windows follow when the passes have found them (gate_forecast says
        when a call without input has something to hand out; lookahead_flush at the end of the input)."""
        self._chk(self._lib.rfid_lookahead_set_consume_ahead(self._h, 1 if on else 0))

    def gate_forecast(self, upstream_done: bool = False) -> bool:
        """-> True when the gate needs input to do anything (the reference's forecast), False when a call without input can."""
        n = C.c_int(1)
        self._chk(self._lib.rfid_gate_forecast(self._h, 1 if upstream_done else 0, C.byref(n)))
        return n.value != 0

    def decoder_work(self, x):
        """-> (consumed, port-0 floats, result record or None, scores record or None)."""
        x = _c64(x)
        bits = np.zeros(16, dtype=np.float32)
        cons, prod = C.c_int(0), C.c_int(0)
        res = np.zeros(1, dtype=capi.RESULT_DTYPE)
        sc = np.zeros(1, dtype=capi.SCORES_DTYPE)
        self._chk(self._lib.rfid_decoder_work(self._h, x.ctypes.data, len(x), bits.ctypes.data, len(bits),
                                              C.byref(cons), C.byref(prod), res.ctypes.data, sc.ctypes.data))
        if cons.value == 0:
            return 0, bits[:0], None, None
        return cons.value, bits[: prod.value].copy(), res[0], sc[0]

    def reader_work(self, n_in: int) -> int:
        cons = C.c_int(0)
        self._chk(self._lib.rfid_reader_work(self._h, int(n_in), C.byref(cons)))
        return cons.value

    def reader_work_tx(self, in_bits=None, dac_rate: int = 1000000):
        """reader_impl::general_work incl. the TX waveform -> (consumed, float32 samples written)."""
        bits = np.ascontiguousarray(in_bits if in_bits is not None else [], dtype=np.float32)
        out = np.zeros(self._lib.rfid_reader_tx_max(int(dac_rate)), dtype=np.float32)
        cons, wr = C.c_int(0), C.c_int(0)
        self._chk(self._lib.rfid_reader_work_tx(self._h, int(dac_rate), bits.ctypes.data if len(bits) else None, len(bits),
                                                out.ctypes.data, len(out), C.byref(cons), C.byref(wr)))
        return cons.value, out[: wr.value].copy()

    # -- (1b) whole-chain streaming ------------------------------------------------------------------
    def lookahead_enable(self, max_chunk_raw: int) -> None:
        """The per-block calls (mf_work / gate_work / decoder_work) answered from one whole-chain pass per mf_work call."""
        self._chk(self._lib.rfid_lookahead_enable(self._h, int(max_chunk_raw)))
        self._planned = (1, int(max_chunk_raw))   # (the look-ahead runs on a one-trace plan of its own: any batch plan is gone)
        self._active = 1

    def lookahead_enable_gate(self, max_items: int) -> None:
        """The same keyed on the gate's input (a flowgraph whose matched filter is not this library's): gate_work uploads
        what is new in its input and runs gate -> tag_decoder over it in one submission."""
        self._chk(self._lib.rfid_lookahead_enable_gate(self._h, int(max_items)))
        self._planned = (1, 5 * int(max_items))   # (the look-ahead runs on a one-trace plan of its own: any batch plan is gone)
        self._active = 1

    def lookahead_pending(self):
        """-> (windows found and not yet handed out by gate_work, windows handed out and waiting for decoder_work)."""
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self._lib.rfid_lookahead_pending(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def lookahead_flush(self) -> None:
        """End of the input: what the look-ahead still holds back is decided now (the gate / decoder calls hand it out)."""
        self._chk(self._lib.rfid_lookahead_flush(self._h))

    def stream_begin(self, max_chunk_raw: int) -> None:
        self._chk(self._lib.rfid_stream_begin(self._h, int(max_chunk_raw)))
        self._stream_cap = 4096
        self._planned = (1, int(max_chunk_raw))   # (the stream runs on a one-trace plan of its own)
        self._active = 1

    def stream_staging(self, idx: int) -> np.ndarray:
        """One of the two pinned staging buffers as a complex64 array (fill it, then pass a slice that starts at
        element 0 to stream_work: the upload is then a true asynchronous DMA)."""
        p, cap = C.c_void_p(), C.c_int64(0)
        self._chk(self._lib.rfid_stream_staging(self._h, int(idx), C.byref(p), C.byref(cap)))
        buf = (C.c_float * (2 * cap.value)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.complex64)

    def stream_work(self, raw=None, flush: bool = False):
        """-> (windows, results) completed by this call (rfid_stream_work)."""
        if raw is None:
            raw = np.zeros(0, dtype=np.complex64)
        raw = np.ascontiguousarray(raw, dtype=np.complex64)
        n = C.c_int64(0)
        ptr = raw.ctypes.data if len(raw) else None
        n_in = len(raw)
        while True:
            w = np.zeros(self._stream_cap, dtype=capi.STREAM_WINDOW_DTYPE)
            r = np.zeros(self._stream_cap, dtype=capi.RESULT_DTYPE)
            st = self._lib.rfid_stream_work(self._h, ptr, n_in, int(bool(flush)), w.ctypes.data, r.ctypes.data,
                                            self._stream_cap, C.byref(n))
            if st == capi.ERR_CAPACITY and n.value > self._stream_cap:   # arrays too small: nothing lost, ask again
                self._stream_cap = int(n.value) * 2
                ptr, n_in = None, 0
                continue
            self._chk(st)
            return w[: n.value].copy(), r[: n.value].copy()

    def stream_end(self) -> None:
        self._chk(self._lib.rfid_stream_end(self._h))

    # -- (2) batched offline, device buffers ---------------------------------------------------
    def batch_plan(self, n_streams: int, max_raw: int) -> None:
        self._chk(self._lib.rfid_batch_plan(self._h, int(n_streams), int(max_raw)))
        self._planned = (int(n_streams), int(max_raw))
        self._active = int(n_streams)

    def batch_set_streams(self, n_streams: int) -> None:
        """Process only the first n_streams (<= planned) rows in the following passes."""
        self._chk(self._lib.rfid_batch_set_streams(self._h, int(n_streams)))
        self._active = int(n_streams)

    def batch_process_ptr(self, d_raw: int, raw_stride: int, n_raw: int, d_lens: int = 0,
                          want_scores: bool = False) -> None:
        """Asynchronous mf->gate->decode->stats over [n_streams][raw_stride] complex64 in HBM."""
        self._chk(self._lib.rfid_batch_process(self._h, C.c_void_p(d_raw), int(raw_stride), int(n_raw),
                                               C.c_void_p(d_lens) if d_lens else None, int(bool(want_scores))))

    def batch_stage(self, which: str, *args) -> None:
        fn = {"mf": self._lib.rfid_batch_mf, "gate": self._lib.rfid_batch_gate,
              "decode": self._lib.rfid_batch_decode, "stats": self._lib.rfid_batch_stats}[which]
        if which == "mf":
            d_raw, raw_stride, n_raw, d_lens = args
            self._chk(fn(self._h, C.c_void_p(d_raw), int(raw_stride), int(n_raw),
                         C.c_void_p(d_lens) if d_lens else None))
        elif which == "decode":
            self._chk(fn(self._h, int(bool(args[0])) if args else 0))
        else:
            self._chk(fn(self._h))

    def batch_set_long_stream(self, mode: int) -> None:
        """0: never cut traces along time, 1: automatic (few long traces), 2: whenever possible."""
        self._chk(self._lib.rfid_batch_set_long_stream(self._h, int(mode)))

    def set_knob(self, name: str, value: int) -> None:
        """One of the switches the environment sets at context creation, by its lower-case name
        without the RFID_ prefix: ctx.set_knob("overlap", 0)."""
        self._chk(self._lib.rfid_ctx_set_knob(self._h, name.encode(), int(value)))

    def get_knob(self, name: str) -> int:
        v = C.c_int(0)
        self._chk(self._lib.rfid_ctx_get_knob(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def batch_ls_report(self) -> dict:
        r = capi.LsReport()
        self._chk(self._lib.rfid_batch_ls_report(self._h, C.byref(r)))
        return {n: int(getattr(r, n)) for n, _ in capi.LsReport._fields_}

    def batch_sync(self) -> None:
        self._chk(self._lib.rfid_batch_sync(self._h))

    def batch_timing(self) -> dict:
        t = capi.BatchTiming()
        self._chk(self._lib.rfid_batch_timing_get(self._h, C.byref(t)))
        return dict(mf_ms=t.mf_ms, gate_ms=t.gate_ms, decode_ms=t.decode_ms, stats_ms=t.stats_ms,
                    total_ms=t.total_ms, front_ms=t.front_ms, front_chunks=t.front_chunks,
                    decode_launches=t.decode_launches, fused_front=t.fused_front)

    def batch_stats(self) -> np.ndarray:
        n = self._active
        out = np.zeros(n, dtype=capi.STATS_DTYPE)
        self._chk(self._lib.rfid_batch_get_stats(self._h, out.ctypes.data, n))
        return out

    def batch_windows(self, want_scores: bool = False):
        """-> (windows, results, scores|None) ordered by (stream, seq)."""
        n = C.c_int64(0)
        self._chk(self._lib.rfid_batch_get_windows(self._h, None, None, None, 0, C.byref(n)))
        k = n.value
        w = np.zeros(k, dtype=capi.WINDOW_DTYPE)
        r = np.zeros(k, dtype=capi.RESULT_DTYPE)
        s = np.zeros(k, dtype=capi.SCORES_DTYPE) if want_scores else None
        if k:
            self._chk(self._lib.rfid_batch_get_windows(self._h, w.ctypes.data, r.ctypes.data,
                                                       s.ctypes.data if s is not None else None, k, C.byref(n)))
        return w, r, s

    def batch_mf_output(self, stream: int) -> np.ndarray:
        cap = self._planned[1] // 5 + 1
        out = np.empty(cap, dtype=np.complex64)
        n = C.c_int64(0)
        self._chk(self._lib.rfid_batch_get_mf(self._h, int(stream), out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy()

    def batch_gated_output(self, stream: int, seq: int) -> np.ndarray:
        """Gated, DC-removed samples of window `seq` of trace `stream` (the gate block's output; debug tap)."""
        out = np.empty(1370, dtype=np.complex64)
        n = C.c_int64(0)
        self._chk(self._lib.rfid_batch_get_gated(self._h, int(stream), int(seq), out.ctypes.data, len(out), C.byref(n)))
        return out[: n.value].copy()

    def synth_replicas_ptr(self, d_base: int, n_raw: int, d_out: int, out_stride: int, n_streams: int, sigma: float,
                           seed: int, first_replica: int = 0) -> None:
        """Asynchronous: d_out[s] = d_base + sigma * complex Gaussian noise (replica first_replica + s)."""
        self._chk(self._lib.rfid_synth_replicas(self._h, C.c_void_p(d_base), int(n_raw), C.c_void_p(d_out), int(out_stride),
                                                int(n_streams), C.c_float(sigma), C.c_uint64(seed), int(first_replica)))

    def _gen2_params(self, plan) -> "capi.SynthGen2Params":
        p = capi.SynthGen2Params()
        lk = np.complex64(plan.leak)
        p.leak_re, p.leak_im = float(lk.real), float(lk.imag)
        for k, h in enumerate(plan.hs):
            hk = np.complex64(h)
            p.h_re[k], p.h_im[k] = float(hk.real), float(hk.imag)
        p.n_tags = len(plan.hs)
        p.tail_us = int(plan.tail_us)
        return p

    def synth_gen2_size(self, plan) -> int:
        """Samples of the trace rfid_synth_gen2 builds from `plan` (rfid.synth.TracePlan)."""
        slots = np.ascontiguousarray(plan.slots)
        n = C.c_int64(0)
        p = self._gen2_params(plan)
        self._chk(self._lib.rfid_synth_gen2_size(C.byref(p), slots.ctypes.data, len(slots), C.byref(n)))
        return n.value

    def synth_gen2_ptr(self, plan, d_out: int, out_cap: int, sigma: float = 0.0, seed: int = 0, replica: int = 0) -> int:
        """Asynchronous: the Gen2 receive trace of `plan` generated in HBM at d_out; returns its sample count."""
        slots = np.ascontiguousarray(plan.slots)
        n = C.c_int64(0)
        p = self._gen2_params(plan)
        self._chk(self._lib.rfid_synth_gen2(self._h, C.byref(p), slots.ctypes.data, len(slots), C.c_void_p(d_out),
                                            int(out_cap), C.c_float(sigma), C.c_uint64(seed), int(replica), C.byref(n)))
        return n.value

    def batch_device_ptrs(self) -> dict:
        y, st, fc = C.c_void_p(), C.c_void_p(), C.c_void_p()
        stride = C.c_int64(0)
        self._chk(self._lib.rfid_batch_device_ptrs(self._h, C.byref(y), C.byref(stride), C.byref(st), C.byref(fc)))
        return dict(mf_out=y.value, mf_stride=stride.value, stats=st.value, flat_count=fc.value)


def unpack_bits(words: np.ndarray, n_bits: int) -> np.ndarray:
    """rfid_decode_result.bits -> array of n_bits 0/1 values (frame order)."""
    words = np.asarray(words, dtype=np.uint32)
    j = np.arange(n_bits)
    return ((words[j >> 5] >> (j & 31)) & 1).astype(np.uint8)
