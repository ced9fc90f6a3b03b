"""ctypes wrapper of tests/wave_emu/librfid_wave_emu.so -- TEST INFRASTRUCTURE ONLY.

Runs the product's kernel source on the lock-step host emulator so kernel logic can be
checked against the oracle without a GPU.  Never imported by the product package."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "gen2-uhf-rfid-reader_amd"))
sys.path.insert(0, HERE)
from rfid import _capi as capi  # noqa: E402  (struct dtypes only)
import build as _build  # noqa: E402

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
    return _lib


def batch_process(raw: np.ndarray, lens=None, fixed_q=0, max_num_queries=1000, number_unique_tags=100,
                  want_y=False, gate_chunk=0, unaligned=False):
    """raw: [B][L] complex64.  -> dict(windows, results, scores, stats, y)"""
    raw = np.ascontiguousarray(raw, dtype=np.complex64)
    if raw.ndim == 1:
        raw = raw[None, :]
    B, L = raw.shape
    # 16-byte aligned copy with even stride so the float4 path is exercised
    stride = (L + 1) & ~1
    buf = np.zeros(B * stride + 2, dtype=np.complex64)
    off = (16 - buf.ctypes.data % 16) % 16 // 8
    if unaligned:          # rows 8-byte aligned only: the kernels' float2 load path
        off ^= 1
    view = buf[off:off + B * stride].reshape(B, stride)
    view[:, :L] = raw
    cap = B * (L // 5 // 347 + 2)
    windows = np.zeros(cap, dtype=capi.WINDOW_DTYPE)
    results = np.zeros(cap, dtype=capi.RESULT_DTYPE)
    scores = np.zeros(cap, dtype=capi.SCORES_DTYPE)
    stats = np.zeros(B, dtype=capi.STATS_DTYPE)
    n = C.c_long(0)
    y = np.zeros((B, L // 5), dtype=np.complex64) if want_y else None
    lens_arr = None
    if lens is not None:
        lens_arr = np.ascontiguousarray(lens, dtype=np.int64)
    rc = lib().emu_batch_process(
        C.c_void_p(view.ctypes.data), B, C.c_long(stride), C.c_long(L),
        C.c_void_p(lens_arr.ctypes.data) if lens_arr is not None else None,
        fixed_q, max_num_queries, number_unique_tags,
        C.c_void_p(windows.ctypes.data), C.c_void_p(results.ctypes.data), C.c_void_p(scores.ctypes.data),
        C.c_long(cap), C.byref(n), C.c_void_p(stats.ctypes.data),
        C.c_void_p(y.ctypes.data) if y is not None else None, C.c_long(gate_chunk))
    assert rc == 0
    k = n.value
    return dict(windows=windows[:k], results=results[:k], scores=scores[:k], stats=stats, y=y)


LS2_CTL_FIELDS = (["fail", "ok", "n_pieces", "n_heads"] + [f"avg_count{r}" for r in range(12)] + [f"avg_list{r}" for r in range(12)] + [f"fsm_count{r}" for r in range(12)] +
                  ["avg_reruns", "fsm_reruns", "dc_reruns", "avg_rounds", "fsm_rounds", "dc_rounds", "n_units", "n_windows",
                   "wb_clash", "n_dc_pieces", "dc_open_alloc", "dc_finished"] + [f"dc_count{r}" for r in range(65)])


def ls2_process(raw: np.ndarray, lens=None, fixed_q=0, max_num_queries=1000, number_unique_tags=100, min_piece=512,
                target=131072, state=None, hold_last=False, cuts=None, y_skip=0, chain_slots=64, generous=True, dc_rounds=-1, fsm_lanes=False, dc_two_levels=False, dc_bias=0,
                fused=False):
    """batch_process() with the long-stream front end (rfid_ls2.hpp) in place of the sequential gate scan.
    -> dict(windows, results, scores, stats, ctl, ok[, consumed])"""
    raw = np.ascontiguousarray(raw, dtype=np.complex64)
    if raw.ndim == 1:
        raw = raw[None, :]
    B, L = raw.shape
    stride = (L + 1) & ~1
    buf = np.zeros(B * stride + 2, dtype=np.complex64)
    off = (16 - buf.ctypes.data % 16) % 16 // 8
    view = buf[off:off + B * stride].reshape(B, stride)
    view[:, :L] = raw
    cap = B * (L // 5 // 347 + 4)
    windows = np.zeros(cap, dtype=capi.WINDOW_DTYPE)
    results = np.zeros(cap, dtype=capi.RESULT_DTYPE)
    scores = np.zeros(cap, dtype=capi.SCORES_DTYPE)
    stats = np.zeros(B, dtype=capi.STATS_DTYPE)
    n = C.c_long(0)
    lens_arr = None
    if lens is not None:
        lens_arr = np.ascontiguousarray(lens, dtype=np.int64)
    lib().emu_ls2_fsm_lanes_min(0 if fsm_lanes else 1 << 30)
    lib().emu_ls2_dcb_bias(int(dc_bias))   # (ulps added to the first round's centres: what the rounding drift of a long trace does to the ring means)
    lib().emu_ls2_dcb_top_min(0 if dc_two_levels else 64)   # (the dc_est chain's second level, as on traces of more than 4 096 idle-grid slots)
    lib().emu_ls2_chain_slots(int(chain_slots))   # (several workgroups per trace in the chain launches, as on long traces)
    nw = lib().emu_ls2_ctl_words()
    ctl = np.zeros(nw, dtype=np.int32)
    consumed = np.zeros(1, dtype=np.int32)
    pcs = np.full((4096, 8), -1, dtype=np.int32)
    cuts_arr = None if cuts is None else np.ascontiguousarray(cuts, dtype=np.int32)
    ok = lib().emu_ls2_process(
        C.c_void_p(view.ctypes.data), B, C.c_long(stride), C.c_long(L),
        C.c_void_p(lens_arr.ctypes.data) if lens_arr is not None else None,
        fixed_q, max_num_queries, number_unique_tags,
        C.c_void_p(windows.ctypes.data), C.c_void_p(results.ctypes.data), C.c_void_p(scores.ctypes.data),
        C.c_long(cap), C.byref(n), C.c_void_p(stats.ctypes.data), int(min_piece), int(target),
        C.c_void_p(ctl.ctypes.data), nw,
        C.c_void_p(state.ctypes.data) if state is not None else None, 1 if hold_last else 0, C.c_void_p(consumed.ctypes.data),
        C.c_void_p(pcs.ctypes.data), len(pcs) - 1,
        C.c_void_p(cuts_arr.ctypes.data) if cuts_arr is not None else None, 0 if cuts_arr is None else len(cuts_arr), int(y_skip), 1 if generous else 0, int(dc_rounds),
        1 if fused else 0)
    assert ok >= 0
    k = n.value
    npc = int(np.argmax(pcs[:, 0] < 0)) if (pcs[:, 0] < 0).any() else len(pcs)
    return dict(windows=windows[:k], results=results[:k], scores=scores[:k], stats=stats, ok=int(ok),
                ctl=dict(zip(LS2_CTL_FIELDS, ctl.tolist())), consumed=int(consumed[0]), pieces=pcs[:npc].copy())


class GateStream:
    def __init__(self):
        self.state = np.zeros(lib().emu_gate_state_size(), dtype=np.uint8)

    def work(self, x: np.ndarray, seek_type: int = -1):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.zeros(max(len(x), 1), dtype=np.complex64)
        c, w, o = C.c_int(0), C.c_int(0), C.c_int(0)
        lib().emu_gate_stream(C.c_void_p(self.state.ctypes.data), C.c_void_p(x.ctypes.data), len(x), seek_type,
                              C.c_void_p(out.ctypes.data), C.byref(c), C.byref(w), C.byref(o))
        return c.value, out[:w.value].copy(), o.value


def decode_one(win: np.ndarray, type_: int):
    win = np.ascontiguousarray(win, dtype=np.complex64)
    res = np.zeros(1, dtype=capi.RESULT_DTYPE)
    sc = np.zeros(1, dtype=capi.SCORES_DTYPE)
    lib().emu_decode_one(C.c_void_p(win.ctypes.data), type_, C.c_void_p(res.ctypes.data), C.c_void_p(sc.ctypes.data))
    return res[0], sc[0]


def chain_scan2(x, ca, cb):
    """-> (sums from carry ca, sums from carry cb, scanned): chain_add_auto2 on one 64-sample step"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    oa = np.zeros(64, dtype=np.float32); ob = np.zeros(64, dtype=np.float32)
    sc = C.c_int(0)
    lib().emu_chain_scan2(C.c_void_p(x.ctypes.data), C.c_float(ca), C.c_float(cb), C.c_void_p(oa.ctypes.data),
                          C.c_void_p(ob.ctypes.data), C.byref(sc))
    return oa, ob, bool(sc.value)


def mf_stream(staging: np.ndarray, in_off: int, n_out: int) -> np.ndarray:
    buf = np.zeros(len(staging) + 2, dtype=np.complex64)
    off = (16 - buf.ctypes.data % 16) % 16 // 8
    if unaligned:          # rows 8-byte aligned only: the kernels' float2 load path
        off ^= 1
    st = buf[off:off + len(staging)]
    st[:] = staging
    out = np.zeros(max(n_out, 1), dtype=np.complex64)
    lib().emu_mf_stream(C.c_void_p(st.ctypes.data), len(st), in_off, n_out, C.c_void_p(out.ctypes.data))
    return out[:n_out]


def selftest(x, num, den, carry):
    x, num, den = (np.ascontiguousarray(a, dtype=np.float32) for a in (x, num, den))
    outs = [np.zeros(64, dtype=np.float32) for _ in range(4)]
    lib().emu_selftest(C.c_void_p(x.ctypes.data), C.c_void_p(num.ctypes.data), C.c_void_p(den.ctypes.data),
                       C.c_float(carry), *[C.c_void_p(o.ctypes.data) for o in outs])
    return outs


def chain_scan(x, carry):
    """-> (chain, scan, clean): the exact 63-deep chain, chain_add_scan (or its fallback), and whether the scan applied."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    chain = np.zeros(64, dtype=np.float32)
    scan = np.zeros(65, dtype=np.float32)
    lib().emu_chain_scan(C.c_void_p(x.ctypes.data), C.c_float(carry), C.c_void_p(chain.ctypes.data), C.c_void_p(scan.ctypes.data))
    return chain, scan[:64], bool(scan[64])


def synth_replicas(base: np.ndarray, n_streams: int, sigma: float, seed: int, first_replica: int = 0) -> np.ndarray:
    base = np.ascontiguousarray(base, dtype=np.complex64)
    out = np.zeros((n_streams, len(base)), dtype=np.complex64)
    lib().emu_synth_replicas(C.c_void_p(base.ctypes.data), C.c_long(len(base)), C.c_void_p(out.ctypes.data),
                             C.c_long(len(base)), n_streams, C.c_float(sigma), C.c_ulonglong(seed), C.c_long(first_replica))
    return out


def synth_gen2(plan, sigma: float = 0.0, seed: int = 0, replica: int = 0) -> np.ndarray:
    """synth_gen2_kernel on the emulator: the trace of an rfid.synth.TracePlan."""
    p = capi.SynthGen2Params()
    lk = np.complex64(plan.leak)
    p.leak_re, p.leak_im = float(lk.real), float(lk.imag)
    for k, h in enumerate(plan.hs):
        hk = np.complex64(h)
        p.h_re[k], p.h_im[k] = float(hk.real), float(hk.imag)
    p.n_tags, p.tail_us = len(plan.hs), int(plan.tail_us)
    slots = np.ascontiguousarray(plan.slots)
    buf = np.zeros(plan.n_raw + 2, dtype=np.complex64)
    off = (16 - buf.ctypes.data % 16) % 16 // 8
    out = buf[off:off + plan.n_raw]
    fn = lib().emu_synth_gen2
    fn.restype = C.c_long
    n = fn(C.byref(p), C.c_void_p(slots.ctypes.data), C.c_long(len(slots)), C.c_void_p(out.ctypes.data),
           C.c_long(plan.n_raw), C.c_float(sigma), C.c_ulonglong(seed), C.c_long(replica))
    assert n == plan.n_raw, (n, plan.n_raw)
    return out.copy()


def philox4x32_10(ctr, key) -> np.ndarray:
    c = np.asarray(ctr, dtype=np.uint32); k = np.asarray(key, dtype=np.uint32); o = np.zeros(4, dtype=np.uint32)
    lib().emu_philox4x32_10(C.c_void_p(c.ctypes.data), C.c_void_p(k.ctypes.data), C.c_void_p(o.ctypes.data))
    return o
