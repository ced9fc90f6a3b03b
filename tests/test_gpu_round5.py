"""Round 5, on the MI355X: the drop-in under a scheduler that behaves like GNU Radio's -- bounded buffers, nobody announces the
end of the input -- and the gathering of calls into big passes; the switches read once per context."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _exe():
    import rfid
    return os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")


def _run(exe, path, tmp_path, tag, extra, env=None):
    files = [tmp_path / f"{k}_{tag}.bin" for k in ("tx", "mf", "gate")]
    out = subprocess.run([exe, str(path), "--time", "--tx-out", str(files[0]), "--mf-out", str(files[1]), "--gate-out", str(files[2])] + extra,
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, (tag, out.stderr[-2000:])
    return out.stdout, [open(f, "rb").read() for f in files]


@pytest.mark.parametrize("cut_behind_last_rn16", [False, True], ids=["whole-trace", "ends-behind-an-RN16-window"])
def test_bounded_scheduler_small_buffers_both_keyings(tmp_path, oracle_mod, synth_mod, cut_behind_last_rn16):
    """mi355x::bounded_flowgraph: GNU Radio's scheduling rules on one thread -- buffers of 8 192 items and no more, forecast, a
    block that could do nothing is left alone until new input arrives or its neighbour is done, stop() at the end and nothing
    else (nobody tells the blocks that the input has ended; the gate's adaptor sees its upstream neighbour finish).  The gate
    consumes ahead (rfid_lookahead_set_consume_ahead: the default of the adaptors under bounded buffers) or decides at once.  Both keyings of the look-ahead (the
    library's matched_filter block in front of the gate; somebody else's filter, as apps/reader.py:75 has it): the report, the
    reader's output and the gated samples are byte for byte those of the per-call path and the oracle's -- the last window
    included, also when the trace ends a few samples behind a complete RN16 window (inside the stretch that the look-ahead's
    passes leave undecided: the exact per-call scan takes it)."""
    import rfid
    exe = _exe()
    t = synth_mod.make_trace(n_rounds=40, fixed_q=1, tag_ids=(0x27, 0x3C), seed=78, sigma=0.01, t1_jitter_raw=4, corrupt_rounds=(7,))
    x = t.samples
    cfg = oracle_mod.config(fixed_q=1)
    if cut_behind_last_rn16:
        o_full = oracle_mod.run_trace(x, cfg)
        rn = [int(s) for s, ty in zip(o_full.open_idx, o_full.dumps["type"]) if ty == 0]
        x = x[: 5 * (rn[-1] + 250 + 9)]          # the last RN16 window is complete, 9 decimated samples follow
    o = oracle_mod.run_trace(x, cfg)
    assert o.n_windows > 40 and (not cut_behind_last_rn16 or o.dumps["type"][-1] == 0)
    path = tmp_path / "t.bin"
    rfid.batch.write_trace_file(str(path), x)
    outs = {}
    for name, extra, env in (("percall", ["--chunk", "8192"], {"RFID_LOOKAHEAD": "0"}),
                             ("bounded_mf", ["--scheduler", "bounded", "--buffer", "8192"], None),
                             ("bounded_hostfir", ["--scheduler", "bounded", "--buffer", "8192", "--host-fir"], None),
                             ("bounded_mf_4096", ["--scheduler", "bounded", "--buffer", "4096"], None),
                             ("bounded_hostfir_4096", ["--scheduler", "bounded", "--buffer", "4096", "--host-fir"], None),
                             # (the gate deciding at once instead of consuming ahead: rfid_lookahead_set_scheduler's bounded mode)
                             ("bounded_mf_decide_at_once", ["--scheduler", "bounded", "--buffer", "8192"], {"RFID_GATE_CONSUME_AHEAD": "0"}),
                             ("bounded_hostfir_decide_at_once", ["--scheduler", "bounded", "--buffer", "8192", "--host-fir"], {"RFID_GATE_CONSUME_AHEAD": "0"}),
                             ("sts_mf_8192", ["--chunk", "8192"], None),
                             ("sts_mf_8192_own_outputs", ["--chunk", "8192"], {"RFID_MF_LATE_OUTPUTS": "0"}),
                             ("bounded_mf_own_outputs", ["--scheduler", "bounded", "--buffer", "8192"], {"RFID_MF_LATE_OUTPUTS": "0"}),
                             ("sts_hostfir_8192", ["--chunk", "8192", "--host-fir"], None)):
        stdout, files = _run(exe, path, tmp_path, name, ["--fixed-q", "1"] + extra, env)
        assert stdout.startswith(o.print_results()), (name, stdout[-800:])
        outs[name] = files
    ref = outs["percall"]
    for key, got in outs.items():
        assert got[0] == ref[0], ("reader output differs", key)
        assert got[1] == ref[1], ("filter output differs", key)
        assert got[2] == ref[2], ("gated samples differ", key)
    # the gated samples ARE the oracle's windows, the last one included
    g = np.frombuffer(ref[2], dtype=np.complex64)
    want = sum(1370 if ty else 250 for ty in o.dumps["type"])
    assert len(g) >= want


def test_calls_gather_into_big_passes(tmp_path, oracle_mod, synth_mod):
    """The single-threaded scheduler reading 8 192 items per turn (GNU Radio's default buffer): since round 5 the calls only
    upload (and filter) their samples, a whole-chain pass runs once 65 536 decimated samples have gathered.  Same bytes as with
    big buffers; the rate is in profiles/r05/drop_in_path.txt."""
    import rfid
    exe = _exe()
    t = synth_mod.make_trace(n_rounds=300, seed=11, sigma=0.004, corrupt_rounds=(36,))
    o = oracle_mod.run_trace(t.samples, oracle_mod.config(max_num_queries=1 << 30))
    path = tmp_path / "t.bin"
    rfid.batch.write_trace_file(str(path), t.samples)
    outs = {}
    for name, extra in (("8192", ["--chunk", "8192"]), ("65536", ["--chunk", "65536"]), ("hostfir_8192", ["--chunk", "8192", "--host-fir"]),
                        ("1000", ["--chunk", "1000"])):
        stdout, files = _run(exe, path, tmp_path, name, ["--max-queries", "100000000"] + extra)
        assert stdout.startswith(o.print_results()), (name, stdout[-800:])
        outs[name] = files
    for key, got in outs.items():
        assert got == outs["65536"], key


def test_knobs_are_read_once_and_settable(synth_mod):
    """The RFID_* switches are read when a context is created; rfid_ctx_set_knob changes one of a living context."""
    import rfid
    os.environ["RFID_OVERLAP"] = "0"
    try:
        ctx = rfid.Context(device=0)
    finally:
        del os.environ["RFID_OVERLAP"]
    try:
        assert ctx.get_knob("overlap") == 0 and ctx.get_knob("long_stream") == 1 and ctx.get_knob("front_chunks") == 1
        os.environ["RFID_OVERLAP"] = "1"          # (changing the environment now changes nothing)
        try:
            assert ctx.get_knob("overlap") == 0
        finally:
            del os.environ["RFID_OVERLAP"]
        ctx.set_knob("overlap", 1)
        assert ctx.get_knob("overlap") == 1
        for bad in (("front_chunks", 99), ("overlap", -1), ("no_such_knob", 1)):
            with pytest.raises(rfid.capi.RfidError):
                ctx.set_knob(*bad)
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_late_filter_outputs_through_the_c_abi(oracle_mod, synth_mod, seed):
    """rfid_lookahead_set_late_outputs: a rfid_mf_work call returns filter outputs of the calls before it -- whatever the device
    has finished, oldest first -- and holds its own back (no call waits for the device; up to three sets are held).  Driven
    like a scheduler with ragged buffers and, every now and then, an output buffer too small for what is held back (the
    outputs then come in parts, through calls that bring no new samples);
    nobody announces the end of the input.  The filter outputs concatenated are those of calls that return their own, and
    every decoded window, the statistics and the report equal the oracle's."""
    import rfid
    rng = np.random.default_rng(100 + seed)
    t = synth_mod.make_trace(n_rounds=40, seed=40 + seed, sigma=0.01, fixed_q=1, tag_ids=(0x11, 0x2A), t1_jitter_raw=4,
                             corrupt_rounds=(9,)).samples
    o = oracle_mod.run_trace(t, oracle_mod.config(fixed_q=1))
    ref = rfid.Context(device=0, fixed_q=1)
    try:
        y_ref = ref.mf_work(t)                                   # (no look-ahead: every call returns its own outputs)
    finally:
        ref.close()
    tb = rfid.reader_top_block(samples=t, chunk=30000, lookahead=True, fixed_q=1)
    try:
        ctx = tb.ctx
        ctx.lookahead_set_late_outputs(True)
        tb._reader_until_idle(0)
        ys = []
        gq = np.zeros(0, dtype=np.complex64)
        dq = np.zeros(0, dtype=np.complex64)
        pos, n, asked_dry = 0, len(t), 0
        # up to three sets of outputs are held back; a fourth call with new samples must have room for the oldest: refused
        # otherwise, nothing consumed
        for k in range(3):
            blk = t[pos:pos + 5000]
            pos += len(blk)
            y = ctx.mf_work(blk, out_cap=1)
            assert len(y) <= 1 and ctx.mf_pending() == (k + 1) * 1000 - sum(len(v) for v in ys) - len(y)
            ys.append(y)
            gq = np.concatenate([gq, y]) if len(gq) else y
        held = ctx.mf_pending()
        with pytest.raises(rfid.capi.RfidError):
            ctx.mf_work(t[pos:pos + 5000], out_cap=10)
        assert ctx.mf_pending() == held
        while pos < n or len(gq) or ctx.mf_pending():
            if pos < n or ctx.mf_pending():
                held = ctx.mf_pending()
                room = held if rng.random() < 0.7 else int(rng.integers(1, held + 2))
                if held > room or pos >= n:
                    blk = t[:0]                                  # what is held back first (in parts when the room is short)
                else:
                    blk = t[pos:pos + int(rng.integers(1000, 150001))]
                pos += len(blk)
                y = ctx.mf_work(blk, out_cap=max(room, 1))
                new_out = len(t[:pos]) // 5 - len(t[:pos - len(blk)]) // 5
                assert len(y) <= min(held + new_out, max(room, 1))   # what the device has finished (its own only when it is through already)
                assert len(blk) or len(y) or not held            # a call without new samples hands out something
                assert ctx.mf_pending() == held - len(y) + new_out
                ys.append(y)
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
                    if pos < n or ctx.mf_pending():
                        break                                    # back to the source / the filter
                    asked_dry += 1                               # the input has ended: asked again, the gate decides what is left
                    if asked_dry > 6:
                        gq = gq[:0]                              # (what is left lies behind the last window)
                else:
                    asked_dry = 0
        got = np.concatenate(ys)
        assert got.tobytes() == y_ref[: len(got)].tobytes() and len(got) == len(t) // 5
        assert ctx.stats() == o.stats()
        assert ctx.print_results() == o.print_results()
        assert len(tb.decoded) == o.n_windows
        for (res, sc), d in zip(tb.decoded, o.dumps):
            assert res["n_bits"] == d["n_bits"] and res["crc_ok"] == d["crc_ok"] and res["index"] == d["index"]
            assert np.array_equal(rfid.unpack_bits(res["bits"], int(d["n_bits"])), d["bits"][: d["n_bits"]])
    finally:
        tb.ctx.close()


@pytest.mark.parametrize("keyed_on", ["filter", "gate"])
@pytest.mark.parametrize("seed", [1, 2])
def test_gate_consumes_ahead_through_the_c_abi(oracle_mod, synth_mod, keyed_on, seed):
    """rfid_lookahead_set_consume_ahead: the gate takes everything it is shown and hands out the windows when the passes have found
    them -- one window per call at most, in parts when the output buffer is short, the next one only after the decoder / reader
    calls have armed the gate.  Driven like a scheduler with ragged, small buffers that honours rfid_gate_forecast and tells the
    library the end of the input when it sees it (as the C++ adaptor does from detail()->input(0)->done()).  Every decoded
    window, the statistics and the report equal the oracle's, the gated samples are the oracle's windows minus their dc estimates."""
    import rfid
    rng = np.random.default_rng(200 + seed)
    t = synth_mod.make_trace(n_rounds=60, seed=60 + seed, sigma=0.01, fixed_q=1, tag_ids=(0x11, 0x2A), t1_jitter_raw=4,
                             corrupt_rounds=(9,)).samples
    o = oracle_mod.run_trace(t, oracle_mod.config(fixed_q=1))
    from rfid.flowgraph import fir_filter_ccc_ones
    y_all = fir_filter_ccc_ones(t)
    # what the gate writes: in[i] - dc_est over every window (gate_impl.cc:171-186), from the oracle's openings and dc estimates
    parts = []
    for s0, ty, dc in zip(o.open_idx, o.dumps["type"], o.dc):
        w = y_all[int(s0): int(s0) + (1370 if ty else 250)]
        parts.append((w.real - np.float32(dc.real)).astype(np.float32) + 1j * (w.imag - np.float32(dc.imag)).astype(np.float32))
    gated_ref = np.concatenate(parts).astype(np.complex64)
    ext = keyed_on == "gate"
    tb = rfid.reader_top_block(samples=t, chunk=8192, lookahead=True, external_filter=ext, fixed_q=1)
    try:
        ctx = tb.ctx
        ctx.lookahead_set_consume_ahead(True)
        tb._reader_until_idle(0)
        gq = np.zeros(0, dtype=np.complex64)
        dq = np.zeros(0, dtype=np.complex64)
        gated = []
        pos, n = 0, (len(y_all) if ext else len(t))
        told_end = False
        spins = 0
        while True:
            if pos < n:
                step = int(rng.integers(200, 9000)) * (1 if ext else 5)
                blk = (y_all if ext else t)[pos:pos + step]
                pos += len(blk)
                y = blk if ext else tb.matched_filter.work(blk)
                gq = np.concatenate([gq, y]) if len(gq) else y
            src_done = pos >= n
            # the gate: called when it has input, or when it says it can do something without
            if len(gq) or not ctx.gate_forecast(upstream_done=src_done):
                if not len(gq) and src_done and not told_end:
                    ctx.lookahead_flush()
                    told_end = True
                take = gq[: int(rng.integers(50, 8193))]
                consumed, out = ctx.gate_work(take, out_cap=int(rng.integers(100, 3000)))
                assert consumed == len(take)                     # everything it is shown
                gq = gq[consumed:]
                if len(out):
                    gated.append(out)
                    dq = np.concatenate([dq, out]) if len(dq) else out
                spins = spins + 1 if (consumed == 0 and len(out) == 0) else 0
                assert spins < 1000
            while True:
                dcons, bits, res, sc = tb.tag_decoder.general_work(dq)
                if dcons == 0:
                    break
                tb.decoded.append((res, sc))
                dq = dq[dcons:]
                tb._reader_until_idle(len(bits))
            if src_done and not len(gq) and ctx.gate_forecast(upstream_done=True):   # (nothing held undecided, no window waiting)
                break
        assert ctx.stats() == o.stats()
        assert ctx.print_results() == o.print_results()
        assert len(tb.decoded) == o.n_windows
        for (res, sc), d in zip(tb.decoded, o.dumps):
            assert res["n_bits"] == d["n_bits"] and res["crc_ok"] == d["crc_ok"] and res["index"] == d["index"]
            assert np.array_equal(rfid.unpack_bits(res["bits"], int(d["n_bits"])), d["bits"][: d["n_bits"]])
        g = np.concatenate(gated)
        assert len(g) == sum(1370 if ty else 250 for ty in o.dumps["type"])
        assert g.tobytes() == gated_ref.tobytes()
    finally:
        tb.ctx.close()


@pytest.mark.parametrize("limit", [["--max-queries", "25"], ["--unique-tags", "1"]], ids=["max-queries", "unique-tags"])
def test_reader_terminates_alike_under_every_scheduler(tmp_path, oracle_mod, synth_mod, limit):
    """MAX_NUM_QUERIES / NUMBER_UNIQUE_TAGS reached in the middle of the trace (gate_impl.cc:101-109: the gate swallows the rest):
    the report, the reader's output and the gated samples of the per-call path, of the single-threaded scheduler and of GNU
    Radio's rules with bounded buffers -- the gate consuming ahead or deciding at once, both keyings -- are the same bytes,
    and the report is the oracle's."""
    import rfid
    exe = _exe()
    t = synth_mod.make_trace(n_rounds=60, fixed_q=1, tag_ids=(0x27, 0x3C), seed=91, sigma=0.01, t1_jitter_raw=4)
    kw = dict(max_num_queries=25) if limit[0] == "--max-queries" else dict(number_unique_tags=1)
    o = oracle_mod.run_trace(t.samples, oracle_mod.config(fixed_q=1, **kw))
    full = oracle_mod.run_trace(t.samples, oracle_mod.config(fixed_q=1, max_num_queries=1 << 30))
    assert 0 < o.n_windows < full.n_windows                      # (the limit bites)
    path = tmp_path / "t.bin"
    rfid.batch.write_trace_file(str(path), t.samples)
    outs = {}
    for name, extra, env in (("percall", ["--chunk", "8192"], {"RFID_LOOKAHEAD": "0"}),
                             ("sts_mf", ["--chunk", "8192"], None),
                             ("sts_hostfir", ["--chunk", "8192", "--host-fir"], None),
                             ("bounded_mf", ["--scheduler", "bounded", "--buffer", "8192"], None),
                             ("bounded_hostfir", ["--scheduler", "bounded", "--buffer", "8192", "--host-fir"], None),
                             ("bounded_hostfir_decide_at_once", ["--scheduler", "bounded", "--buffer", "8192", "--host-fir"], {"RFID_GATE_CONSUME_AHEAD": "0"})):
        stdout, files = _run(exe, path, tmp_path, name, ["--fixed-q", "1"] + limit + extra, env)
        assert stdout.startswith(o.print_results()), (name, stdout[-800:])
        outs[name] = files
    for key, got in outs.items():
        assert got[0] == outs["percall"][0], ("reader output differs", key)
        assert got[2] == outs["percall"][2], ("gated samples differ", key)
