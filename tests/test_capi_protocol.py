"""The HOST side of the C-ABI library on the CPU -- csrc/rfid_capi.hip, unmodified: contexts, plans, the per-call path, the whole-chain
stream and the look-ahead protocols with their two keyings, late filter outputs, consume-ahead, the end of the input -- ~3 000 lines
that until round 6 only ever ran on a GPU box.  tests/fake_hip builds that source with g++ against a stand-in HIP runtime whose
"launches" run the UNMODIFIED kernel source on the wave emulator (tests/wave_emu); the library is bound with ctypes HERE, in the test
process only (the product's loader knows nothing of it and fails without the HIP build: tests/test_capi_load.py).

The cases are the GPU suite's own protocol tests (tests/test_gpu_*.py: the ones that talk to the library through host buffers), called
with traces of a few inventory rounds; each compares with the oracle exactly as it does on the device.  A second set runs with
FAKE_HIP_LAG > 0: work enqueued on a stream becomes runnable only some runtime calls later, so that the host meets passes that are
still "running" -- the orderings a real device produces."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "fake_hip"))


class SmallSynth:
    """rfid.synth with make_trace's n_rounds capped: the GPU suite's cases at a size the emulator runs in seconds"""

    def __init__(self, mod, cap):
        self._mod, self._cap = mod, cap

    def __getattr__(self, name):
        return getattr(self._mod, name)

    def make_trace(self, *a, **kw):
        if "n_rounds" in kw:
            kw["n_rounds"] = min(int(kw["n_rounds"]), self._cap)
        kw["corrupt_rounds"] = tuple(r for r in kw.get("corrupt_rounds", ()) if r < kw.get("n_rounds", 1 << 30)) or ((1,) if kw.get("corrupt_rounds") else ())
        return self._mod.make_trace(*a, **kw)


@pytest.fixture(scope="module", autouse=True)
def emulated_library():
    """librfid_capi_emu.so in place of librfid_mi355x.so -- for this module's tests, in this process, and put back afterwards"""
    import build_capi_emu as fake_build
    import rfid
    from rfid import _capi
    lib = C.CDLL(fake_build.build())
    for name, (res, args) in _capi.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    saved = _capi._lib
    _capi._lib = lib
    yield lib
    _capi._lib = saved


def _small(synth_mod, cap=4):
    return SmallSynth(synth_mod, cap)


def test_context_on_the_stand_in_device_and_knobs(synth_mod):
    import rfid
    import test_gpu_round5 as g
    ctx = rfid.Context(device=0)
    assert ctx.get_knob("long_stream") == 1
    ctx.close()
    with pytest.raises(rfid.capi.RfidError):
        rfid.Context(device=1)          # (one device here)
    g.test_knobs_are_read_once_and_settable(_small(synth_mod))


def test_per_call_blocks_match_the_oracle(oracle_mod, synth_mod):
    import test_gpu_parity as g
    g.test_streaming_blocks_match_oracle(oracle_mod, _small(synth_mod, 3))


def test_termination_after_max_queries_stream_and_blocks(oracle_mod, synth_mod):
    """MAX_NUM_QUERIES reached inside the trace (gate_impl.cc:101-109,125): the whole-chain stream and the per-block calls stop
    delivering windows where the oracle does; the stream refuses calls once it is closed."""
    import rfid
    t = synth_mod.make_trace(n_rounds=7, seed=5150, sigma=0.01).samples
    cfg = oracle_mod.config(max_num_queries=3)
    o = oracle_mod.run_trace(t, cfg)
    assert o.state.status == 1 and 0 < o.n_windows < 14
    ctx = rfid.Context(device=0, max_num_queries=3)
    try:
        ctx.stream_begin(40_000)
        n_win = 0
        for pos in range(0, len(t), 40_000):
            w, r = ctx.stream_work(t[pos:pos + 40_000])
            n_win += len(w)
        w, r = ctx.stream_work(flush=True)
        n_win += len(w)
        assert n_win == o.n_windows
        assert ctx.stats() == o.stats() and ctx.print_results() == o.print_results()
        ctx.stream_end()
        with pytest.raises(rfid.capi.RfidError):
            ctx.stream_work(t[:1000])                 # the stream is closed
    finally:
        ctx.close()
    for la in (False, True):
        tb = rfid.reader_top_block(samples=t, chunk=9000, lookahead=la, max_num_queries=3)
        try:
            tb.run()
            assert tb.ctx.stats() == o.stats() and tb.ctx.print_results() == o.print_results(), la
        finally:
            tb.ctx.close()


def test_capacity_and_state_errors_of_the_look_ahead(oracle_mod, synth_mod):
    """A filter call larger than the stream was enabled for (goes through in pieces); late outputs: held back, counted by
    rfid_mf_pending, fetched by a call without input, not to be switched off while something is held (RFID_ERR_STATE); new samples
    behind the announced end of the input (RFID_ERR_STATE)."""
    import rfid
    t = synth_mod.make_trace(n_rounds=3, seed=9, sigma=0.01).samples
    ctx = rfid.Context(device=0)
    try:
        ctx.lookahead_enable(20000)
        ref = oracle_mod.fir(t[:45005])
        y0 = ctx.mf_work(t[:25005])                       # larger than the stream was enabled for: goes through in pieces
        assert len(y0) == 5001 and np.array_equal(y0.view(np.uint32), ref[:5001].view(np.uint32))
        ctx.lookahead_set_late_outputs(True)
        y1 = ctx.mf_work(t[25005:35005])                  # (nothing to hand out yet: the call's own outputs are held back)
        assert len(y1) + ctx.mf_pending() == 2000         # (FAKE_HIP_LAG > 0: the device is not through with them yet, they are held back)
        if ctx.mf_pending():
            with pytest.raises(rfid.capi.RfidError) as e:
                ctx.lookahead_set_late_outputs(False)     # not while outputs are held
            assert e.value.status == rfid.capi.ERR_STATE
        y2 = ctx.mf_work(t[35005:45005])
        y3 = ctx.mf_work(t[:0])                           # a call without input fetches what is held
        y4 = ctx.mf_work(t[:0])
        got = np.concatenate([y0, y1, y2, y3, y4])
        assert len(got) == 9001 and np.array_equal(got.view(np.uint32), ref[:9001].view(np.uint32))
        assert ctx.mf_pending() == 0
        ctx.lookahead_flush()
        with pytest.raises(rfid.capi.RfidError) as e:
            ctx.mf_work(t[45005:46005])                   # the stream has ended
        assert e.value.status == rfid.capi.ERR_STATE
    finally:
        ctx.close()


def test_reader_tx_waveform(oracle_mod, synth_mod):
    import test_gpu_parity as g
    g.test_reader_tx_waveform_matches_oracle(oracle_mod, _small(synth_mod, 3))


def test_python_blocks_bind_per_flowgraph(oracle_mod, synth_mod):
    import test_gpu_round2 as g
    g.test_python_blocks_bind_per_flowgraph(oracle_mod, _small(synth_mod, 3))


@pytest.mark.parametrize("sizes", [(7000, 23000, 5000), (200000,)])
def test_whole_chain_stream_calls(oracle_mod, synth_mod, sizes):
    import test_gpu_round2 as g
    g.test_whole_chain_streaming_equals_batch_and_oracle(oracle_mod, _small(synth_mod, 4), sizes)


def test_stream_with_a_stretch_that_cannot_be_cut(oracle_mod, synth_mod):
    import test_gpu_round3 as g
    g.test_stream_with_a_stretch_that_cannot_be_cut(oracle_mod, _small(synth_mod, 4))


def test_python_flowgraphs_with_look_ahead_both_keyings(oracle_mod, synth_mod):
    import test_gpu_round3 as g3
    import test_gpu_round4 as g4
    g3.test_python_flowgraph_with_look_ahead(oracle_mod, _small(synth_mod, 4))
    g4.test_python_flowgraph_with_a_foreign_filter(oracle_mod, _small(synth_mod, 4))


@pytest.mark.parametrize("seed", [1, 2])
def test_look_ahead_with_ragged_scheduler_calls(oracle_mod, synth_mod, seed):
    import test_gpu_round3 as g
    g.test_look_ahead_with_ragged_scheduler_calls(oracle_mod, _small(synth_mod, 5), seed)


@pytest.mark.parametrize("early_flush", [False, True])
def test_gate_keyed_look_ahead(oracle_mod, synth_mod, early_flush):
    import test_gpu_round4 as g
    g.test_gate_keyed_look_ahead_through_the_c_abi(oracle_mod, _small(synth_mod, 5), early_flush)


@pytest.mark.parametrize("seed", [1, 2])
def test_late_filter_outputs(oracle_mod, synth_mod, seed):
    import test_gpu_round5 as g
    g.test_late_filter_outputs_through_the_c_abi(oracle_mod, _small(synth_mod, 5), seed)


@pytest.mark.parametrize("keyed_on", ["filter", "gate"])
def test_gate_consumes_ahead(oracle_mod, synth_mod, keyed_on):
    import test_gpu_round5 as g
    g.test_gate_consumes_ahead_through_the_c_abi(oracle_mod, _small(synth_mod, 5), keyed_on, 1)


@pytest.mark.parametrize("lag", [3, 11])
def test_the_same_protocols_on_a_device_that_is_behind(lag):
    """The look-ahead cases again in a process whose stand-in runtime finishes work only `lag` runtime calls after it was enqueued: the
    host meets passes that are still running, filter outputs that are not there yet (held back, up to three sets), flag words not
    written -- the orderings of a real device.  Same results."""
    env = dict(os.environ, FAKE_HIP_LAG=str(lag))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "late_filter or consumes_ahead or ragged_scheduler or gate_keyed or capacity_and_state or whole_chain or termination"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("mode", [0, 2], ids=["fused-front-end", "long-stream"])
def test_batch_pass_through_the_library(oracle_mod, synth_mod, mode):
    """rfid_batch_plan / rfid_batch_process / the getters over a ragged batch of two traces ("device" pointers are host pointers here):
    the fused front end of many traces, and the long-stream front end with its fused first pass, dc_est chain and finishing walk
    enqueued by the library's own launch list -- windows, results, scores and statistics equal the oracle's."""
    import parity
    import rfid
    ts = [synth_mod.make_trace(n_rounds=3 + k, seed=70 + k, sigma=0.02, t1_jitter_raw=3).samples for k in range(2)]
    L = max(map(len, ts))
    stride = (L + 1) & ~1
    host = np.zeros((2, stride), dtype=np.complex64)
    lens = np.array([len(t) for t in ts], dtype=np.int64)
    lens[0] -= 777
    for i, t in enumerate(ts):
        host[i, : len(t)] = t
    refs = [oracle_mod.run_trace(host[b, : lens[b]]) for b in range(2)]
    ctx = rfid.Context(device=0)
    try:
        ctx.batch_set_long_stream(mode)
        ctx.batch_plan(2, L)
        for rep in range(2):
            ctx.batch_process_ptr(host.ctypes.data, stride, L, lens.ctypes.data, want_scores=True)
            ctx.batch_sync()
            w, r, s = ctx.batch_windows(want_scores=True)
            st = ctx.batch_stats()
            for b, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, 2)):
                parity.compare_trace(wb, rb, sb, st[b], refs[b])
        rep = ctx.batch_ls_report()
        assert (rep["pieces"] > 0 and rep["verified"] == 1) if mode == 2 else rep["pieces"] == 0, rep
    finally:
        ctx.close()
