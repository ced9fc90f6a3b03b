"""GPU tests added in round 3: the launcher-free multi-rank bench, a stream with a stretch the long-stream front end
cannot cut, non-finite samples."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.gpu
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the script re-executes itself once per rank and rank 0 prints the
    one JSON line (n_gpus = 2, one time per rank, whole-job value).  On a one-GPU box both ranks share the device and the
    control plane runs over gloo (RCCL refuses two ranks per device); the data path is the same."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(RFID_BENCH_SHARE_DEVICES="1", RFID_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "bench.py"), "--gpus", "2", "--streams", "48", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-stream-leg"], cwd=ROOT_DIR, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["ms_per_step_by_rank"]) == 2 and d["scaling"] == "weak"
    assert d["parity_check"].startswith("ok") and d["value"] > 0
    assert abs(d["ms_per_step"] - max(d["ms_per_step_by_rank"])) < 1e-3


def _stream(ctx, x, chunk):
    ws, rs = [], []
    for pos in range(0, len(x), chunk):
        w, r = ctx.stream_work(x[pos:pos + chunk])
        ws.append(w); rs.append(r)
    w, r = ctx.stream_work(flush=True)
    ws.append(w); rs.append(r)
    return np.concatenate(ws), np.concatenate(rs)


def test_stream_with_a_stretch_that_cannot_be_cut(oracle_mod, synth_mod):
    """rfid_stream_work over a stream whose middle is 0.35 s of noise without a carrier (no idle point of the gate for
    far longer than a call holds back) and whose end is a carrier with readers' commands but hardly any quiet time: the
    calls that cannot be cut go through the sequential scan from the carried state instead of failing, and the stream
    stays usable.  Windows, dc_est, decoded fields and statistics equal the oracle's over the whole stream."""
    import rfid
    a = synth_mod.make_trace(n_rounds=12, seed=31, sigma=0.01).samples
    b = synth_mod.make_trace(n_rounds=12, seed=32, sigma=0.01).samples
    rng = np.random.default_rng(5)
    gap = (0.02 * (rng.standard_normal(700_000) + 1j * rng.standard_normal(700_000))).astype(np.complex64)
    x = np.concatenate([a, gap, b]).astype(np.complex64)
    o = oracle_mod.run_trace(x)
    ctx = rfid.Context(device=0)
    try:
        ctx.stream_begin(250_000)
        w, r = _stream(ctx, x, 200_000)
        assert len(w) == o.n_windows, (len(w), o.n_windows)
        assert np.array_equal(w["start"], o.open_idx) and np.array_equal(w["type"], o.dumps["type"])
        assert np.array_equal(w["dc_re"].view(np.uint32), o.dc.real.view(np.uint32))
        assert np.array_equal(w["dc_im"].view(np.uint32), o.dc.imag.view(np.uint32))
        fake = np.zeros(len(w), dtype=rfid.capi.WINDOW_DTYPE)
        fake["start"], fake["type"], fake["dc_re"], fake["dc_im"] = w["start"], w["type"], w["dc_re"], w["dc_im"]
        parity.compare_trace_fast(fake, r, None, o)
        assert ctx.stats() == o.stats()
        ctx.stream_end()
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", [0, 2, 3], ids=["fused", "long-stream", "long-stream-fsm-on-lanes"])
def test_non_finite_samples_propagate_as_in_the_reference(oracle_mod, synth_mod, mode, monkeypatch):
    """The contract for samples that are not finite: nothing is rejected or sanitised -- they go through the matched
    filter and the gate's recurrences exactly as through the reference's (gate_impl.cc:130-162: a NaN amplitude turns
    avg_ampl into NaN for good, every threshold test then fails and the gate never opens again; an infinity becomes a
    NaN when it leaves the 100-sample ring).  Window positions, types and dc_est are bit-identical to the oracle's, and
    so is every decoded field of the windows that hold finite samples only; what the decoder makes of a window that
    itself contains a non-finite sample is unspecified (the reference's own answer hangs on how std::max_element
    orders NaNs)."""
    import rfid
    import torch
    t = synth_mod.make_trace(n_rounds=6, sigma=0.01, seed=5).samples
    clean = oracle_mod.run_trace(t)
    cases = [("inf in the carrier", int(clean.open_idx[6]) * 5 - 2000, complex(np.inf, 0.0)),
             ("nan inside a window", int(clean.open_idx[7]) * 5 + 300, complex(np.nan, 1.0)),
             ("-inf inside a window", int(clean.open_idx[3]) * 5 + 100, complex(0.0, -np.inf))]
    if mode == 3:      # (the state machine's one-lane-per-unit form, which the library takes on long passes only)
        monkeypatch.setenv("RFID_LS2_FSM_LANES_MIN", "0")
        mode = 2
    ctx = rfid.Context(device=0)
    try:
        ctx.batch_set_long_stream(mode)
        for name, pos, val in cases:
            x = t.copy()
            x[pos] = np.complex64(val)
            o = oracle_mod.run_trace(x)
            L = len(x)
            stride = (L + 1) & ~1
            host = np.zeros((1, stride), dtype=np.complex64)
            host[0, :L] = x
            dev = torch.from_numpy(host.view(np.float32)).to("cuda:0")
            ctx.batch_plan(1, L)
            ctx.batch_process_ptr(dev.data_ptr(), stride, L, 0, want_scores=False)
            ctx.batch_sync()
            w, r, _ = ctx.batch_windows()
            assert o.n_windows < clean.n_windows, name                  # (the gate does go blind behind the bad sample)
            assert len(w) == o.n_windows, (name, len(w), o.n_windows)
            assert np.array_equal(w["start"], o.open_idx) and np.array_equal(w["type"], o.dumps["type"]), name
            assert np.array_equal(w["dc_re"].view(np.uint32), o.dc.real.view(np.uint32)), name
            assert np.array_equal(w["dc_im"].view(np.uint32), o.dc.imag.view(np.uint32)), name
            y = oracle_mod.fir(x)
            assert np.array_equal(ctx.batch_mf_output(0).view(np.uint32), y.view(np.uint32)), name
            for i in range(o.n_windows):
                wlen = 1370 if o.dumps["type"][i] else 250
                if not np.isfinite(y[o.open_idx[i]: o.open_idx[i] + wlen].view(np.float32)).all():
                    continue
                d = o.dumps[i]
                assert r["index"][i] == d["index"] and r["n_bits"][i] == d["n_bits"], (name, i)
                assert np.array_equal(rfid.unpack_bits(r["bits"][i], int(d["n_bits"])), d["bits"][: d["n_bits"]]), (name, i)
                assert r["crc_ok"][i] == d["crc_ok"], (name, i)
    finally:
        ctx.close()


def test_block_by_block_calls_with_look_ahead_give_the_same_bytes(tmp_path, oracle_mod, synth_mod):
    """rfid_reader_offline block by block (one C-ABI call per general_work) with the library's look-ahead on (default)
    and off: the same report, the same reader output (--tx-out), matched-filter output and gated samples, byte for
    byte, for a scheduler buffer of 65 536 and of 1 000 items -- and the oracle's report.  The look-ahead must also be
    faster (it exists because the per-call path is slower than one CPU core)."""
    import rfid
    exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
    t = synth_mod.make_trace(n_rounds=120, fixed_q=1, tag_ids=(0x27, 0x3C), seed=77, sigma=0.01, t1_jitter_raw=4, corrupt_rounds=(5,))
    path = tmp_path / "t.bin"
    rfid.batch.write_trace_file(str(path), t.samples)
    o = oracle_mod.run_trace(t.samples, oracle_mod.config(fixed_q=1))
    rates = {}
    for chunk in ("65536", "1000"):
        outs = {}
        for la in ("1", "0"):
            files = [tmp_path / f"{k}_{chunk}_{la}.bin" for k in ("tx", "mf", "gate")]
            env = dict(os.environ, RFID_LOOKAHEAD=la)
            out = subprocess.run([exe, str(path), "--fixed-q", "1", "--chunk", chunk, "--time", "--tx-out", str(files[0]),
                                  "--mf-out", str(files[1]), "--gate-out", str(files[2])], capture_output=True, text=True,
                                 timeout=900, env=env)
            assert out.returncode == 0, out.stderr
            assert out.stdout.startswith(o.print_results()), (chunk, la)
            outs[la] = [open(f, "rb").read() for f in files]
            rates[(chunk, la)] = float(out.stderr.split(" ms = ")[1].split(" Msamples/s")[0])
        assert outs["1"][0] == outs["0"][0], "reader output differs"
        assert outs["1"][1] == outs["0"][1], "matched-filter output differs"
        assert outs["1"][2] == outs["0"][2], "gated samples differ"
        y = np.frombuffer(outs["1"][1], dtype=np.complex64)
        assert np.array_equal(y.view(np.uint32), oracle_mod.fir(t.samples).view(np.uint32))
    print("block-by-block rates, Msamples/s:", rates)
    assert rates[("65536", "1")] > rates[("65536", "0")]      # (1.8 M samples: ~10 ms of one-time staging allocation included;
                                                              #  on the 30 M-sample trace the ratio is 8-9: profiles/r03/drop_in_path.txt)


def test_python_flowgraph_with_look_ahead(oracle_mod, synth_mod):
    """rfid.reader_top_block (the apps/reader.py topology on the ctypes binding) with the look-ahead: the oracle's report."""
    import rfid
    t = synth_mod.make_trace(n_rounds=20, seed=9, sigma=0.01).samples
    o = oracle_mod.run_trace(t)
    tb = rfid.reader_top_block(samples=t, chunk=20000, lookahead=True)
    try:
        tb.run()
        assert tb.ctx.stats() == o.stats()
        assert tb.ctx.print_results() == o.print_results()
        assert len(tb.decoded) == o.n_windows
    finally:
        tb.ctx.close()


def test_short_passes_start_with_few_rounds_and_escalate(oracle_mod, synth_mod):
    """A short pass enqueues few re-run rounds (a round without work is two empty launches, which only a short pass
    notices).  A trace that needs more -- 8 % noise: dc_est passes close to powers of two all the time -- runs out of
    them: the front end gives up, the sequential scan behind it gives the (identical) result, and from the next pass on
    the library enqueues the full number of rounds: then the front end settles."""
    import rfid
    import torch
    t = synth_mod.make_trace(n_rounds=20, sigma=0.08, seed=9).samples
    o = oracle_mod.run_trace(t)
    L = len(t)
    stride = (L + 1) & ~1
    host = np.zeros((1, stride), dtype=np.complex64)
    host[0, :L] = t
    dev = torch.from_numpy(host.view(np.float32)).to("cuda:0")
    ctx = rfid.Context(device=0)
    try:
        ctx.batch_set_long_stream(2)
        ctx.batch_plan(1, L)
        reps = []
        for _ in range(3):
            ctx.batch_process_ptr(dev.data_ptr(), stride, L, 0, want_scores=True)
            ctx.batch_sync()
            reps.append(ctx.batch_ls_report())
            w, r, s = ctx.batch_windows(want_scores=True)
            parity.compare_trace(w, r, s, ctx.batch_stats()[0], o)
        print([(x["verified"], x["gave_up"], x["avg_rounds"], x["dc_rounds"]) for x in reps])
        assert reps[-1]["verified"] == 1, str(reps)
        if reps[0]["verified"] == 0:
            assert reps[0]["gave_up"] in (2, 3, 4), reps[0]
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_look_ahead_with_ragged_scheduler_calls(oracle_mod, synth_mod, seed):
    """The per-block calls with look-ahead, driven like a scheduler whose buffers never have the same fill twice: filter
    calls of 1 000 .. 150 000 raw samples, gate calls handed 50 .. 30 000 items of what is available, the gate asked again
    at random although it just said it can decide nothing (which makes it collect the pending pass and, the third time,
    put the held-back samples through the sequential scan -- never end the stream); at the end of the input the library
    is told so (rfid_lookahead_flush).  Every decoded window, the statistics and the report equal the oracle's."""
    import rfid
    rng = np.random.default_rng(seed)
    t = synth_mod.make_trace(n_rounds=40, seed=20 + seed, sigma=0.01, fixed_q=1, tag_ids=(0x11, 0x2A), t1_jitter_raw=4,
                             corrupt_rounds=(7,)).samples
    o = oracle_mod.run_trace(t, oracle_mod.config(fixed_q=1))
    tb = rfid.reader_top_block(samples=t, chunk=30000, lookahead=True, fixed_q=1)
    try:
        tb._reader_until_idle(0)
        gq = np.zeros(0, dtype=np.complex64)
        dq = np.zeros(0, dtype=np.complex64)
        pos, n, idle, flushed = 0, len(t), 0, False
        while pos < n or len(gq):
            if pos < n:
                blk = t[pos:pos + int(rng.integers(1000, 150001))]
                pos += len(blk)
                y = tb.matched_filter.work(blk)
                gq = np.concatenate([gq, y]) if len(gq) else y
            while len(gq):
                take = gq[: int(rng.integers(50, 30001))]
                consumed, out = tb.gate.general_work(take)
                gq = gq[consumed:]
                if len(out):
                    dq = np.concatenate([dq, out]) if len(dq) else out
                while True:
                    dcons, bits, res, sc = tb.tag_decoder.general_work(dq)
                    if dcons == 0:
                        break
                    tb.decoded.append((res, sc))
                    dq = dq[dcons:]
                    tb._reader_until_idle(len(bits))
                if consumed == 0 and len(out) == 0:
                    if pos < n:
                        if rng.random() < 0.6:
                            break                   # back to the source, as a scheduler does
                        idle += 1                   # ... or ask again all the same: that must not end the stream
                        if idle > 6:
                            idle = 0
                            break
                    elif not flushed:
                        tb.ctx.lookahead_flush()    # the source has run dry
                        flushed = True
                    else:
                        gq = gq[:0]                 # (what is left lies behind the last window: the gate swallowed it)
                else:
                    idle = 0
        assert tb.ctx.stats() == o.stats()
        assert tb.ctx.print_results() == o.print_results()
        assert len(tb.decoded) == o.n_windows
        for (res, sc), d in zip(tb.decoded, o.dumps):
            assert res["n_bits"] == d["n_bits"] and res["crc_ok"] == d["crc_ok"] and res["index"] == d["index"]
            assert np.array_equal(rfid.unpack_bits(res["bits"], int(d["n_bits"])), d["bits"][: d["n_bits"]])
    finally:
        tb.ctx.close()
