"""GPU tests added in round 4: the bench's control plane on the real RCCL backend, its other_configs legs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT_DIR, "bench.py")] + args, cwd=ROOT_DIR, env=env,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_control_plane_on_rccl_single_rank():
    """`bench.py --gpus 1` with RFID_BENCH_FORCE_DIST=1: one rank goes through init_process_group / barrier / all_gather /
    all_gather_object on the `nccl` (= RCCL) backend -- the calls the driver's 8-GPU run meets for the first time."""
    d = _bench(["--gpus", "1", "--streams", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-stream-leg",
                "--no-other-configs"],
               dict(RFID_BENCH_FORCE_DIST="1", RFID_BENCH_BACKEND="nccl", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                    MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert d["n_gpus"] == 1 and len(d["ms_per_step_by_rank"]) == 1 and len(d["devices_by_rank"]) == 1
    assert "nccl" in d["control_plane"]
    assert d["parity_check"].startswith("ok") and d["value"] > 0
    assert d["roofline"]["frac_of_achievable"] > d["roofline"]["frac"] > 0


def test_bench_other_configs_leg():
    """The default line carries short measurements of configs[3] (always) and configs[2] (when HBM allows), each with its
    own result check.  Here at reduced sizes of the headline (64 traces) so that the test stays short."""
    d = _bench(["--streams", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-stream-leg"], {})
    oc = d["other_configs"]
    c3 = oc["configs[3] per GPU"]
    assert c3["parity_check"].startswith("ok") and c3["ms_per_step"] > 0 and "FAILED" not in c3
    assert any(k in c3["roofline_by_kernel"] for k in ("front_long_stream", "gate_long_stream", "front_end_fused"))
    c2 = oc["configs[2]"]
    if "skipped" not in c2:
        assert c2["parity_check"].startswith("ok") and c2["windows_per_step"] == 320000 and "FAILED" not in c2


def test_gate_keyed_look_ahead_offline_binary(tmp_path, oracle_mod, synth_mod):
    """apps/reader.py as it stands builds GNU Radio's own fir_filter_ccc (apps/reader.py:75): the first buffer this
    library sees is the gate's.  `rfid_reader_offline --host-fir` is that topology -- the FIR a plain host loop, only
    gate / tag_decoder / reader made -- and the gate switches the look-ahead on keyed on its own input
    (rfid_lookahead_enable_gate): the report, the reader output and the gated samples are byte for byte those of the
    flowgraph with the library's matched_filter block and of the per-call path (RFID_LOOKAHEAD=0), and the oracle's."""
    import rfid
    exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
    t = synth_mod.make_trace(n_rounds=60, fixed_q=1, tag_ids=(0x27, 0x3C), seed=78, sigma=0.01, t1_jitter_raw=4, corrupt_rounds=(7,))
    path = tmp_path / "t.bin"
    rfid.batch.write_trace_file(str(path), t.samples)
    o = oracle_mod.run_trace(t.samples, oracle_mod.config(fixed_q=1))
    outs = {}
    for name, extra, la in (("mf", [], "1"), ("hostfir", ["--host-fir"], "1"), ("hostfir_percall", ["--host-fir"], "0")):
        for chunk in ("65536", "3000"):
            files = [tmp_path / f"{k}_{name}_{chunk}.bin" for k in ("tx", "mf", "gate")]
            out = subprocess.run([exe, str(path), "--fixed-q", "1", "--chunk", chunk, "--time", "--tx-out", str(files[0]),
                                  "--mf-out", str(files[1]), "--gate-out", str(files[2])] + extra, capture_output=True, text=True,
                                 timeout=900, env=dict(os.environ, RFID_LOOKAHEAD=la))
            assert out.returncode == 0, (name, chunk, out.stderr[-2000:])
            assert out.stdout.startswith(o.print_results()), (name, chunk, out.stdout[-800:])
            outs[(name, chunk)] = [open(f, "rb").read() for f in files]
    ref = outs[("mf", "65536")]
    for key, got in outs.items():
        assert got[0] == ref[0], ("reader output differs", key)
        assert got[1] == ref[1], ("filter output differs", key)      # (the host loop sums in the library's tap order)
        assert got[2] == ref[2], ("gated samples differ", key)


@pytest.mark.parametrize("early_flush", [False, True])
def test_gate_keyed_look_ahead_through_the_c_abi(oracle_mod, synth_mod, early_flush):
    """(early_flush: the end of the input is announced while the gate has not been shown all of it yet -- found by
    profiles/tools/fuzz_lookahead.py: the library knows only the samples the gate was shown, so the flush is carried out
    once gate calls keep showing nothing new.)  The same through the ctypes binding, with a scheduler that shows the gate ragged views of its input (unconsumed
    samples again, new ones behind them) and decoder calls that ask for scores: windows, bits, statistics equal the
    oracle's and the queue of windows waiting for their decoder call stays bounded (every decoder call retires its window)."""
    import rfid
    t = synth_mod.make_trace(n_rounds=30, seed=12, sigma=0.01).samples
    o = oracle_mod.run_trace(t)
    y = oracle_mod.fir(t)
    ctx = rfid.Context(device=0)
    try:
        ctx.lookahead_enable_gate(20000)
        ctx.reader_work(0); ctx.reader_work(0)               # START -> SEND_QUERY -> IDLE
        rng = np.random.default_rng(3)
        shown, pos, dq, n_dec, max_dq = 0, 0, np.zeros(0, np.complex64), 0, 0
        flushed = False
        idle = 0
        while True:
            if shown < len(y):
                shown = min(len(y), shown + int(rng.integers(500, 9000)))
            view = y[pos:min(shown, pos + (7000 if early_flush else 30000))]
            if early_flush and shown >= len(y) and not flushed:
                ctx.lookahead_flush(); flushed = True            # (some 20000 samples before the last one the gate has been shown)
            cons, out = ctx.gate_work(view) if len(view) else (0, np.zeros(0, np.complex64))
            pos += cons
            dq = np.concatenate([dq, out])
            while True:
                c, bits, res, sc = ctx.decoder_work(dq)
                if c == 0:
                    break
                d = o.dumps[n_dec]
                assert np.array_equal(rfid.unpack_bits(res["bits"], int(d["n_bits"])), d["bits"][: d["n_bits"]]), n_dec
                assert np.array_equal(sc["corr"].view(np.uint32), d["corr"].view(np.uint32)), n_dec
                n_dec += 1
                dq = dq[c:]
                for _ in range(4):
                    if ctx.state().gen2_logic_status == rfid.capi.IDLE:
                        break
                    ctx.reader_work(len(bits)); bits = bits[:0]
                max_dq = max(max_dq, ctx.lookahead_pending()[1])
            if cons == 0 and len(out) == 0:
                if shown >= len(y):
                    if not flushed:
                        ctx.lookahead_flush(); flushed = True
                    else:
                        idle += 1
                        if idle > 4:
                            break
            else:
                idle = 0
        assert n_dec == o.n_windows
        assert ctx.stats() == o.stats()
        assert max_dq <= 2, max_dq
    finally:
        ctx.close()


def test_passes_enqueued_back_to_back_keep_their_results_apart(oracle_mod, synth_mod):
    """With a second set of result tables (RFID_OVERLAP=2), the decoder + statistics of pass k run (on a second stream) beside
    the front end of pass k + 1.  Passes over two DIFFERENT batches enqueued back to back without a sync in between: what
    the getters return afterwards is the last pass's, bit for bit what a context without the second set returns."""
    import torch
    import rfid
    B = 96
    batches = []
    for kind in range(2):
        tr = [synth_mod.make_trace(n_rounds=3 + kind, seed=100 * kind + i, sigma=0.01, tag_ids=(0x20 + kind,)).samples for i in range(4)]
        L = min(len(t) for t in tr)
        batches.append(np.stack([tr[i % 4][:L] for i in range(B)]))
    L = min(b.shape[1] for b in batches)
    stride = (L + 1) & ~1
    dev = []
    for b in batches:
        host = np.zeros((B, stride), dtype=np.complex64)
        host[:, :L] = b[:, :L]
        dev.append(torch.from_numpy(host.view(np.float32)).to("cuda:0"))
    want = []
    for b in batches:
        o = [oracle_mod.run_trace(b[i, :L]) for i in range(4)]
        want.append(o)
    got = {}
    for overlap in ("2", "0"):
        os.environ["RFID_OVERLAP"] = overlap
        try:
            ctx = rfid.Context(device=0)
            ctx.batch_plan(B, L)
            seq = [0, 1, 0, 0, 1, 0, 1]          # the last one is batch 1
            for k in seq:
                ctx.batch_process_ptr(dev[k].data_ptr(), stride, L, 0, want_scores=False)
            ctx.batch_sync()
            st1 = ctx.batch_stats().copy()
            w1, r1, _ = ctx.batch_windows()
            ctx.batch_process_ptr(dev[0].data_ptr(), stride, L, 0, want_scores=False)    # and batch 0 once more
            ctx.batch_sync()
            st0 = ctx.batch_stats().copy()
            w0, r0, _ = ctx.batch_windows()
            got[overlap] = (st0, w0, r0, st1, w1, r1)
            ctx.close()
        finally:
            os.environ.pop("RFID_OVERLAP", None)
    for a, b in zip(got["2"], got["0"]):
        assert a.tobytes() == b.tobytes()
    st0, _, _, st1, _, _ = got["2"]
    for i in range(B):
        assert st0[i]["n_epc_correct"] == want[0][i % 4].state.n_epc_correct and st0[i]["n_windows"] == want[0][i % 4].n_windows
        assert st1[i]["n_epc_correct"] == want[1][i % 4].state.n_epc_correct and st1[i]["n_windows"] == want[1][i % 4].n_windows
        assert st1[i]["tag_reads"][0x21] == want[1][i % 4].state.n_epc_correct


def test_python_flowgraph_with_a_foreign_filter(oracle_mod, synth_mod):
    """rfid.reader_top_block(external_filter=True): the filter is numpy on the host (as GNU Radio's own block would be in
    apps/reader.py), the library sees gate / tag_decoder / reader only; with and without the gate-keyed look-ahead the
    oracle's report -- and the host filter IS the oracle's FIR, bit for bit."""
    import rfid
    from rfid.flowgraph import fir_filter_ccc_ones
    t = synth_mod.make_trace(n_rounds=12, seed=19, sigma=0.01).samples
    assert np.array_equal(fir_filter_ccc_ones(t).view(np.uint32), oracle_mod.fir(t).view(np.uint32))
    o = oracle_mod.run_trace(t)
    for la in (True, False):
        tb = rfid.reader_top_block(samples=t, chunk=6000, lookahead=la, external_filter=True)
        try:
            tb.run()
            assert tb.ctx.stats() == o.stats(), la
            assert tb.ctx.print_results() == o.print_results()
            assert len(tb.decoded) == o.n_windows
        finally:
            tb.ctx.close()


@pytest.mark.parametrize("kw", [dict(max_num_queries=300), dict(number_unique_tags=1), dict()])
def test_statistics_of_a_long_trace_from_the_decoders_summaries(oracle_mod, synth_mod, kw):
    """A plan whose traces can hold more than 2 048 windows: the decoder leaves a one-word summary of every result and the
    statistics kernel (one workgroup per trace) reads those, 16 bytes per lane, instead of the 48-byte results.  Counts, tag
    reads and the TERMINATED cut-off (queries limit inside the trace, distinct-tag limit, none) equal the oracle's."""
    import torch
    import rfid
    t = synth_mod.make_trace(n_rounds=260, sigma=0.01, seed=97, fixed_q=1, tag_ids=(0x31, 0x52), t1_jitter_raw=3).samples
    L = len(t)
    assert L // 5 // 347 > 2048                      # (the plan's window capacity per trace)
    o = oracle_mod.run_trace(t, oracle_mod.config(fixed_q=1, **kw))
    stride = (L + 1) & ~1
    host = np.zeros((1, stride), dtype=np.complex64)
    host[0, :L] = t
    dev = torch.from_numpy(host.view(np.float32)).to("cuda:0")
    ctx = rfid.Context(device=0, fixed_q=1, **kw)
    try:
        ctx.batch_plan(1, L)
        for _ in range(2):
            ctx.batch_process_ptr(dev.data_ptr(), stride, L, 0, want_scores=False)
        ctx.batch_sync()
        st = ctx.batch_stats()[0]
        assert st["n_windows_used"] == o.n_windows and st["n_windows"] >= o.n_windows
        assert st["status"] == o.state.status
        for k in ("n_queries_sent", "cur_inventory_round", "cur_slot_number", "n_epc_correct", "n_unique_tags"):
            assert st[k] == getattr(o.state, k), k
        assert np.array_equal(st["tag_reads"], np.array(o.state.tag_reads[:], dtype=np.int32))
    finally:
        ctx.close()


@pytest.mark.parametrize("overlap", ["1", "0"], ids=["second-stream", "one-stream"])
def test_long_stream_passes_back_to_back_with_the_filter_in_parts(oracle_mod, synth_mod, overlap, monkeypatch):
    """Long-stream passes enqueued one behind the other: the first launches of pass k + 1 -- the fused first pass, which filters
    the raw samples itself -- run on the second stream beside the rest of pass k, into the second matched-filter output buffer and
    the second work space (RFID_OVERLAP=0: everything on one stream, one buffer).  Two different traces alternate without a wait
    in between; what the getters return behind the last pass (and behind one more) is that trace's own result, window for
    window the oracle's.  (Round 4's form -- the matched filter of the next pass in three launches behind events of the pass
    before -- went with its knobs in round 6.)"""
    import torch
    import rfid
    monkeypatch.setenv("RFID_OVERLAP", overlap)      # (read when the context is created)
    ts = [synth_mod.make_trace(n_rounds=600, sigma=0.01, seed=300 + k, fixed_q=1, tag_ids=(0x21 + k, 0x44), t1_jitter_raw=2).samples for k in range(2)]
    L = min(len(t) for t in ts)
    assert L // 5 * 8 >= (16 << 20) and (L // 5 + 511) // 512 >= 3 * 1024      # a second filter buffer is made; the filter runs in parts
    refs = [oracle_mod.run_trace(t[:L], oracle_mod.config(fixed_q=1, max_num_queries=1 << 30)) for t in ts]
    stride = (L + 1) & ~1
    devs = []
    for t in ts:
        host = np.zeros((1, stride), dtype=np.complex64)
        host[0, :L] = t[:L]
        devs.append(torch.from_numpy(host.view(np.float32)).to("cuda:0"))
    ctx = rfid.Context(device=0, fixed_q=1, max_num_queries=1 << 30)
    try:
        ctx.batch_set_long_stream(2)
        ctx.batch_plan(1, L)
        for k in (0, 1, 0, 0, 1, 0, 1):
            ctx.batch_process_ptr(devs[k].data_ptr(), stride, L, 0, want_scores=False)
        ctx.batch_sync()
        for last in (1, 0):
            if last == 0:
                ctx.batch_process_ptr(devs[0].data_ptr(), stride, L, 0, want_scores=False)
                ctx.batch_sync()
            assert ctx.batch_ls_report()["verified"] == 1
            w, r, _ = ctx.batch_windows()
            o = refs[last]
            assert len(w) == o.n_windows
            assert np.array_equal(w["start"], o.open_idx) and np.array_equal(w["type"], o.dumps["type"])
            assert np.array_equal(w["dc_re"].view(np.uint32), o.dc.real.view(np.uint32))
            assert np.array_equal(w["dc_im"].view(np.uint32), o.dc.imag.view(np.uint32))
            assert np.array_equal(r["crc_ok"], o.dumps["crc_ok"]) and np.array_equal(r["index"], o.dumps["index"])
            assert ctx.batch_stats()[0]["n_epc_correct"] == o.state.n_epc_correct
    finally:
        ctx.close()
