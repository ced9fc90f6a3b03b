"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs.  Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest

import parity

pytestmark = pytest.mark.gpu


def _to_device(raw2d):
    import torch
    B, L = raw2d.shape
    stride = (L + 1) & ~1
    host = np.zeros((B, stride), dtype=np.complex64)
    host[:, :L] = raw2d
    return torch.from_numpy(host.view(np.float32)).to("cuda:0"), stride


def _run_batch(ctx, raw2d, lens=None, scores=True):
    import torch
    dev, stride = _to_device(raw2d)
    B, L = raw2d.shape
    ctx.batch_plan(B, L)
    d_lens = 0
    keep = None
    if lens is not None:
        keep = torch.tensor(np.asarray(lens, dtype=np.int64)).to("cuda:0")
        d_lens = keep.data_ptr()
    ctx.batch_process_ptr(dev.data_ptr(), stride, L, d_lens, want_scores=scores)
    ctx.batch_sync()
    w, r, s = ctx.batch_windows(want_scores=scores)
    return w, r, s, ctx.batch_stats()


def test_selftest_primitives(gpu_ctx):
    """DPP wave-shift chain, IEEE division, glibc-hypotf formula, wave shift: bit-exact vs host."""
    assert gpu_ctx.selftest() == 0


def test_library_is_loaded_natively(gpu_ctx):
    import rfid
    with open("/proc/self/maps") as f:
        assert "librfid_mi355x.so" in f.read()
    assert rfid.capi.load().rfid_version().decode().startswith("rfid_mi355x")


@pytest.mark.parametrize("sigma,seed", [(0.002, 1), (0.03, 2), (0.06, 3)])
def test_batch_matches_oracle(gpu_ctx, oracle_mod, synth_mod, sigma, seed):
    traces = [synth_mod.make_trace(n_rounds=4, sigma=sigma, seed=seed * 10 + i, t1_jitter_raw=6).samples
              for i in range(3)]
    L = min(map(len, traces))
    raw = np.stack([t[:L] for t in traces])
    w, r, s, st = _run_batch(gpu_ctx, raw)
    for b, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, len(raw))):
        parity.compare_trace(wb, rb, sb, st[b], oracle_mod.run_trace(raw[b]))


def test_matched_filter_bit_exact(gpu_ctx, oracle_mod, synth_mod):
    t = synth_mod.make_trace(n_rounds=1, sigma=0.05, seed=5).samples
    for L in (len(t), len(t) - 3, 2561, 25, 7, 5):
        raw = t[:L][None, :]
        _run_batch(gpu_ctx, raw, scores=False)
        y = gpu_ctx.batch_mf_output(0)
        yo = oracle_mod.fir(raw[0])
        assert np.array_equal(y.view(np.uint32), yo.view(np.uint32)), L


def test_ragged_and_empty_traces(gpu_ctx, oracle_mod, synth_mod):
    a = synth_mod.make_trace(n_rounds=3, seed=21).samples
    b = synth_mod.make_trace(n_rounds=2, seed=22).samples
    L = len(a)
    raw = np.zeros((4, L), dtype=np.complex64)
    raw[0] = a
    raw[1, : len(b)] = b
    raw[2, : len(b) // 2] = b[: len(b) // 2]     # cut inside a round: last window incomplete
    lens = [L, len(b), len(b) // 2, 0]           # one empty trace
    w, r, s, st = _run_batch(gpu_ctx, raw, lens=lens)
    for i, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, 4)):
        parity.compare_trace(wb, rb, sb, st[i], oracle_mod.run_trace(raw[i, : lens[i]]))
    assert st[3]["n_windows"] == 0


def test_fixed_q4_collisions_and_empty_slots(gpu_ctx, oracle_mod, synth_mod):
    import rfid
    t = synth_mod.make_trace(n_rounds=2, fixed_q=2, tag_ids=(0x11, 0x22, 0x33, 0x44, 0x55), seed=31,
                             sigma=0.01).samples
    ctx = rfid.Context(device=0, fixed_q=2)
    try:
        w, r, s, st = _run_batch(ctx, t[None, :])
        parity.compare_trace(w, r, s, st[0], oracle_mod.run_trace(t, oracle_mod.config(fixed_q=2)))
    finally:
        ctx.close()


def test_termination_after_max_queries(oracle_mod, synth_mod):
    import rfid
    t = synth_mod.make_trace(n_rounds=8, seed=41).samples
    ctx = rfid.Context(device=0, max_num_queries=5)
    try:
        w, r, s, st = _run_batch(ctx, t[None, :])
        o = oracle_mod.run_trace(t, oracle_mod.config(max_num_queries=5))
        assert st[0]["status"] == 1 == o.state.status
        assert st[0]["n_windows_used"] == o.n_windows
        for k in ("n_queries_sent", "cur_inventory_round", "n_epc_correct"):
            assert st[0][k] == getattr(o.state, k)
    finally:
        ctx.close()


def test_streaming_blocks_match_oracle(oracle_mod, synth_mod):
    """The per-block C-ABI calls (rfid_mf_work / rfid_gate_work / rfid_decoder_work /
    rfid_reader_work) driven like the reference's offline flowgraph."""
    import rfid
    t = synth_mod.make_trace(n_rounds=3, seed=51, sigma=0.01)
    tb = rfid.reader_top_block(samples=t.samples, device=0, chunk=1777)
    try:
        tb.run()
        o = oracle_mod.run_trace(t.samples)
        assert tb.ctx.stats() == o.stats()
        assert tb.ctx.print_results() == o.print_results()
        assert len(tb.decoded) == o.n_windows
        for (res, sc), d in zip(tb.decoded, o.dumps):
            assert res["index"] == d["index"]
            assert np.array_equal(rfid.unpack_bits(res["bits"], int(d["n_bits"])), d["bits"][: d["n_bits"]])
            assert np.array_equal(sc["corr"].view(np.uint32), d["corr"].view(np.uint32))
    finally:
        tb.ctx.close()


def test_cxx_offline_flowgraph_binary(tmp_path, oracle_mod, synth_mod):
    """bin/rfid_reader_offline = apps/reader.py's DEBUG topology in C++ on the reference's own block API
    (gr::rfid::gate::make(int) etc., cxx/include/rfid/*.h; one C-ABI call per general_work): its print_results text equals the
    oracle's for the same trace file, FIXED_Q = 0 and 2, odd chunk size included."""
    import os
    import subprocess
    import rfid
    exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    for q, chunk in ((0, 8192), (2, 1531)):
        t = synth_mod.make_trace(n_rounds=3, fixed_q=q, tag_ids=(0x27, 0x42) if q else (0x27,), seed=77 + q, sigma=0.01)
        path = tmp_path / f"trace_q{q}.bin"
        rfid.batch.write_trace_file(str(path), t.samples)
        out = subprocess.run([exe, str(path), "--fixed-q", str(q), "--chunk", str(chunk)], capture_output=True,
                             text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        o = oracle_mod.run_trace(t.samples, oracle_mod.config(fixed_q=q))
        assert out.stdout == o.print_results()


def test_fst_like_known_answer(gpu_ctx, synth_mod):
    """README.md:48-53 shape on the stand-in trace: 71 queries / round 72 / 70 EPC / 1 tag 0x27 x70."""
    t = synth_mod.fst_like_trace()
    w, r, s, st = _run_batch(gpu_ctx, t.samples[None, :], scores=False)
    assert st[0]["n_queries_sent"] - 1 == 71 and st[0]["cur_inventory_round"] == 72
    assert st[0]["n_epc_correct"] == 70 and st[0]["n_unique_tags"] == 1 and st[0]["tag_reads"][0x27] == 70
    # every decoded frame equals the generator's ground truth
    epc = r[r["type"] == 1]
    import rfid
    for res, slot in zip(epc, t.slots):
        assert list(rfid.unpack_bits(res["bits"], 128)) == slot.epc
        assert bool(res["crc_ok"]) == slot.epc_valid


def test_multi_tag_q4_long_trace_matches_oracle(oracle_mod, synth_mod):
    """BASELINE.json configs[2] shape at a size the oracle finishes in a second: FIXED_Q=4
    (16 slots per round), 8 tags, empty and collided slots, 12 rounds (192 slots, ~2.7 M raw
    samples), cut into ragged copies.  Everything bit-exact, incl. the slot/round counters."""
    import rfid
    t = synth_mod.make_trace(n_rounds=12, fixed_q=4, tag_ids=(3, 17, 39, 77, 120, 200, 201, 255), seed=77,
                             sigma=0.01, t1_jitter_raw=4).samples
    L = len(t)
    raw = np.zeros((5, L), dtype=np.complex64)     # 5 traces: the gate workgroup holds 4 -> partial group
    lens = [L, L - 12345, L // 2, L // 3, 17]
    for i, n in enumerate(lens):
        raw[i, :n] = t[:n]
    ctx = rfid.Context(device=0, fixed_q=4)
    try:
        w, r, s, st = _run_batch(ctx, raw, lens=lens)
        cfg = oracle_mod.config(fixed_q=4)
        for i, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, 5)):
            parity.compare_trace(wb, rb, sb, st[i], oracle_mod.run_trace(raw[i, : lens[i]], cfg))
        assert st[0]["cur_inventory_round"] == 13 and st[0]["n_queries_sent"] == 193
    finally:
        ctx.close()


def test_front_end_time_chunking_matches_single_launch(gpu_ctx, oracle_mod, synth_mod, monkeypatch):
    """rfid_batch_process with RFID_FRONT_CHUNKS: matched filter and gate scan overlapped on two
    streams, gate state carried from chunk to chunk -- identical results."""
    t = synth_mod.make_trace(n_rounds=9, seed=88, sigma=0.02).samples
    raw = np.stack([t, np.roll(t, 7)])
    gpu_ctx.set_knob("front_chunks", 4)      # (what RFID_FRONT_CHUNKS=4 sets when a context is created)
    try:
        w, r, s, st = _run_batch(gpu_ctx, raw)
    finally:
        gpu_ctx.set_knob("front_chunks", 1)
    assert gpu_ctx.batch_timing()["front_chunks"] == 4
    for b, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, 2)):
        parity.compare_trace(wb, rb, sb, st[b], oracle_mod.run_trace(raw[b]))


def test_fused_front_end_equals_stage_kernels(gpu_ctx, oracle_mod, synth_mod):
    """rfid_batch_process() runs the fused front end (matched filter inside the gate launch);
    rfid_batch_mf + rfid_batch_gate + rfid_batch_decode + rfid_batch_stats run the stage
    kernels one by one.  Same windows, results, scores, stats and matched-filter output, for
    16-byte aligned rows and for rows that are only 8-byte aligned (odd stride)."""
    import torch
    t = synth_mod.make_trace(n_rounds=5, seed=314, sigma=0.03, t1_jitter_raw=9).samples
    raw = np.stack([t, np.roll(t, 11), t * np.float32(0.25)])
    B, L = raw.shape
    for stride in ((L + 1) & ~1, L | 1):
        host = np.zeros((B, stride), dtype=np.complex64)
        host[:, :L] = raw
        dev = torch.from_numpy(host.view(np.float32)).to("cuda:0")
        gpu_ctx.batch_plan(B, L)
        gpu_ctx.batch_set_long_stream(0)       # (3 short traces would otherwise be cut along time: see below)
        gpu_ctx.batch_process_ptr(dev.data_ptr(), stride, L, 0, want_scores=True)
        gpu_ctx.batch_sync()
        assert gpu_ctx.batch_timing()["fused_front"] == 1
        w1, r1, s1 = gpu_ctx.batch_windows(want_scores=True)
        st1 = gpu_ctx.batch_stats().copy()
        y1 = [gpu_ctx.batch_mf_output(b) for b in range(B)]
        gpu_ctx.batch_stage("mf", dev.data_ptr(), stride, L, 0)
        gpu_ctx.batch_stage("gate")
        gpu_ctx.batch_stage("decode", True)
        gpu_ctx.batch_stage("stats")
        gpu_ctx.batch_sync()
        assert gpu_ctx.batch_timing()["fused_front"] == 0
        w2, r2, s2 = gpu_ctx.batch_windows(want_scores=True)
        st2 = gpu_ctx.batch_stats()
        assert w1.tobytes() == w2.tobytes() and r1.tobytes() == r2.tobytes() and s1.tobytes() == s2.tobytes()
        assert st1.tobytes() == st2.tobytes()
        # third front end: the long-stream one (traces cut along time, all pieces processed at once, accepted only when
        # every piece's run is exact or provably covers its true start value)
        gpu_ctx.batch_set_long_stream(2)
        gpu_ctx.batch_process_ptr(dev.data_ptr(), stride, L, 0, want_scores=True)
        gpu_ctx.batch_sync()
        gpu_ctx.batch_set_long_stream(1)
        rep = gpu_ctx.batch_ls_report()
        assert rep["verified"] == 1 and rep["pieces"] > B, rep
        w3, r3, s3 = gpu_ctx.batch_windows(want_scores=True)
        assert w1.tobytes() == w3.tobytes() and r1.tobytes() == r3.tobytes() and s1.tobytes() == s3.tobytes()
        assert st1.tobytes() == gpu_ctx.batch_stats().tobytes()
        for b in range(B):
            assert np.array_equal(y1[b].view(np.uint32), gpu_ctx.batch_mf_output(b).view(np.uint32))
            assert np.array_equal(y1[b].view(np.uint32), oracle_mod.fir(raw[b]).view(np.uint32))
        for b, (wb, rb, sb) in enumerate(parity.split_by_stream(w1, r1, s1, B)):
            parity.compare_trace(wb, rb, sb, st1[b], oracle_mod.run_trace(raw[b]))


def test_full_size_properties_idempotence_and_scaling(gpu_ctx, synth_mod):
    """Size-independent properties on full-length traces (the 71-round stand-in, 1.08 M raw samples each, 48 noise
    replicas): (1) idempotence -- a second pass over the same batch returns byte-identical windows, results, scores
    and stats although the window lists are filled through atomics in arbitrary order; (2) homogeneity -- scaling
    every sample by 4 (exact in binary32) leaves every decision, index and bit unchanged and scales dc_est and
    h_est by exactly 4, the correlation / energy scores by exactly 16; (3) every replica decodes 70 of 71 EPCs."""
    import torch
    base = synth_mod.fst_like_trace().samples
    B, L = 48, len(base)
    rng = np.random.default_rng(5)
    noise = (rng.standard_normal((B, L, 2)).astype(np.float32) * np.float32(0.004)).view(np.complex64)[..., 0]
    raw = (base[None, :] + noise).astype(np.complex64)
    w1, r1, s1, st1 = _run_batch(gpu_ctx, raw)
    st1 = st1.copy()
    w2, r2, s2, st2 = _run_batch(gpu_ctx, raw)
    assert w1.tobytes() == w2.tobytes() and r1.tobytes() == r2.tobytes() and s1.tobytes() == s2.tobytes()
    assert st1.tobytes() == st2.tobytes()
    assert (st1["n_epc_correct"] == 70).all() and (st1["tag_reads"][:, 0x27] == 70).all()
    w4, r4, s4, st4 = _run_batch(gpu_ctx, raw * np.float32(4.0))
    assert st4.tobytes() == st1.tobytes()
    for f in ("stream", "seq", "start", "type"):
        assert np.array_equal(w4[f], w1[f])
    assert np.array_equal(w4["dc_re"], w1["dc_re"] * np.float32(4)) and np.array_equal(w4["dc_im"], w1["dc_im"] * np.float32(4))
    for f in ("type", "index", "bits", "n_bits", "crc_ok", "tag_id", "T"):
        assert np.array_equal(r4[f], r1[f]), f
    assert np.array_equal(r4["h_re"], r1["h_re"] * np.float32(4)) and np.array_equal(r4["h_im"], r1["h_im"] * np.float32(4))
    assert np.array_equal(s4["corr"], s1["corr"] * np.float32(16)) and np.array_equal(s4["energy"], s1["energy"] * np.float32(16))


def test_randomized_batches_match_oracle(oracle_mod, synth_mod):
    """Randomised sweep: 3 x 64 traces with random tag sets (FIXED_Q 0, 1, 3: empty and collided slots), round
    counts, noise levels up to the decode limit, T1 jitter and truncation points, decoded as ragged batches.
    Every window, decision, score and counter bit-identical to the oracle."""
    import torch
    import rfid
    rng = np.random.default_rng(2026)
    total = 0
    for q in (0, 1, 3):
        ctx = rfid.Context(device=0, fixed_q=q)
        try:
            traces = []
            for _ in range(64):
                ntags = 1 if q == 0 else int(rng.integers(1, 5))
                ids = tuple(int(x) for x in rng.choice(256, size=ntags, replace=False))
                t = synth_mod.make_trace(n_rounds=int(rng.integers(1, 4)), fixed_q=q, tag_ids=ids,
                                         seed=int(rng.integers(1 << 30)), sigma=float(rng.uniform(0, 0.08)),
                                         t1_jitter_raw=int(rng.integers(0, 12))).samples
                cut = int(rng.integers(0, 4000)) if rng.random() < 0.3 else 0
                traces.append(t[: len(t) - cut])
            L = max(map(len, traces))
            raw = np.zeros((len(traces), L), dtype=np.complex64)
            for i, t in enumerate(traces):
                raw[i, : len(t)] = t
            w, r, s, st = _run_batch(ctx, raw, lens=[len(t) for t in traces])
            for b, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, len(traces))):
                parity.compare_trace(wb, rb, sb, st[b], oracle_mod.run_trace(traces[b], oracle_mod.config(fixed_q=q)))
                total += len(wb)
        finally:
            ctx.close()
    assert total > 2000


def test_reader_tx_waveform_matches_oracle(oracle_mod, synth_mod):
    """rfid_reader_work_tx (the complete reader block: transitions + transmit waveform, reader_impl.cc:43-380)
    against the oracle's restatement, state by state, for several Q values and both DAC rates."""
    import rfid
    rn16 = np.array([1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 0, 1, 0, 1], dtype=np.float32)
    for q, dac in ((0, 1000000), (4, 1000000), (15, 2000000), (3, 800000)):   # 800 kHz: non-integer sample_d
        ctx = rfid.Context(device=0, fixed_q=q)
        sim = oracle_mod.ReaderTxSim(dac_rate=dac, cfg=oracle_mod.config(fixed_q=q))
        try:
            def both(bits=None):
                cons, w = ctx.reader_work_tx(bits, dac_rate=dac)
                wo = sim.work(bits)
                assert np.array_equal(w, wo)
                st = ctx.state()
                assert (st.gen2_logic_status, st.n_queries_sent, st.gate_status, st.decoder_status) == \
                    (sim.state.gen2_logic_status, sim.state.n_queries_sent, sim.state.gate_status, sim.state.decoder_status)
                assert cons == (0 if bits is None else len(bits))
                return w
            assert len(both()) == 4575 * dac // 1000000          # START: carrier
            w = both()                                            # Query
            if dac % 1000000 == 0:
                assert np.array_equal(w[:: dac // 1000000][: len(synth_mod.query_cmd(q))], synth_mod.query_cmd(q))
            assert len(both()) == 0                               # IDLE
        finally:
            ctx.close()
    # ACK / carrier / QueryRep through the real flow: the streaming flowgraph leaves the states to the blocks
    tb = rfid.reader_top_block(samples=synth_mod.make_trace(n_rounds=2, seed=9, sigma=0.005).samples, device=0)
    try:
        tb.run()
        assert tb.ctx.stats()["n_epc_correct"] == 2
    finally:
        tb.ctx.close()


def test_reader_tx_capacity_error_leaves_state_untouched(gpu_ctx):
    import ctypes as C
    import rfid
    lib = rfid.capi.load()
    assert lib.rfid_reader_tx_max(1000000) >= 4575 and lib.rfid_reader_tx_max(2000000) >= 9150
    gpu_ctx.reset()
    before = gpu_ctx.state().gen2_logic_status
    out = np.zeros(100, dtype=np.float32)
    cons, wr = C.c_int(0), C.c_int(0)
    st = lib.rfid_reader_work_tx(gpu_ctx._h, 1000000, None, 0, out.ctypes.data, len(out), C.byref(cons), C.byref(wr))
    assert st == rfid.capi.ERR_CAPACITY and gpu_ctx.state().gen2_logic_status == before
    cons2, w = gpu_ctx.reader_work_tx()
    assert len(w) == 4575 and gpu_ctx.state().gen2_logic_status == rfid.capi.SEND_QUERY


def test_cxx_offline_binary_writes_the_reader_tx_stream(tmp_path, oracle_mod, synth_mod):
    """rfid_reader_offline --tx-out: the reader block's float output over a whole run (what apps/reader.py's
    DEBUG file sink records) = START carrier, then per slot Query|QueryRep + 1295 us CW, ACK + 4575 us CW,
    with the ACK carrying the RN16 the decoder read."""
    import os
    import subprocess
    import rfid
    exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
    t = synth_mod.make_trace(n_rounds=3, seed=12, sigma=0.005)
    path, txp = tmp_path / "t.bin", tmp_path / "tx.f32"
    rfid.batch.write_trace_file(str(path), t.samples)
    out = subprocess.run([exe, str(path), "--tx-out", str(txp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    tx = np.fromfile(str(txp), dtype=np.float32)
    want = [np.ones(synth_mod.CW_ACK, np.float32)]
    for r, slot in enumerate(t.slots):
        want.append(np.concatenate([synth_mod.query_cmd(0), np.ones(synth_mod.CW_QUERY, np.float32)]))
        want.append(synth_mod.ack_cmd(slot.rn16))
        want.append(np.ones(synth_mod.CW_ACK, np.float32))
    want.append(np.concatenate([synth_mod.query_cmd(0), np.ones(synth_mod.CW_QUERY, np.float32)]))   # the next Query is already out
    assert np.array_equal(tx, np.concatenate(want))


def test_extreme_amplitudes_match_oracle(gpu_ctx, oracle_mod, synth_mod):
    """The same trace scaled into the binary32 corners: products and increments in the denormal range
    (3e-39 .. 1e-30: the constant-division fast path hands over to the generic division), energies
    overflowing to +inf (1e18), and an all-zero trace.  Everything stays bit-identical to the oracle."""
    t = synth_mod.make_trace(n_rounds=3, seed=5, sigma=0.01).samples
    for scale in (1e-20, 1e-30, 1e-36, 3e-39, 1e15, 1e18, 0.0):
        x = (t * np.float32(scale)).astype(np.complex64)
        w, r, s, st = _run_batch(gpu_ctx, x[None, :])
        parity.compare_trace(w, r, s, st[0], oracle_mod.run_trace(x))


def test_device_replica_generator(gpu_ctx, synth_mod):
    """rfid_synth_replicas: deterministic, piecewise-consistent, N(0, sigma^2) per component, and the generated
    batch decodes like the base trace (70 of 71 EPCs in every replica of the stand-in)."""
    import torch
    base = synth_mod.fst_like_trace().samples
    L, B = len(base), 6
    stride = (L + 1) & ~1
    d_base = torch.from_numpy(base.view(np.float32).copy()).to("cuda:0")
    out = torch.zeros((B, 2 * stride), dtype=torch.float32, device="cuda:0")
    out2 = torch.zeros_like(out)
    torch.cuda.synchronize()     # the library runs on its own stream: torch's fills must be over
    gpu_ctx.synth_replicas_ptr(d_base.data_ptr(), L, out.data_ptr(), stride, B, 0.004, seed=1234)
    gpu_ctx.batch_plan(B, L)
    gpu_ctx.batch_sync()
    a = out.cpu().numpy().view(np.complex64)[:, :L].copy()
    gpu_ctx.synth_replicas_ptr(d_base.data_ptr(), L, out2.data_ptr(), stride, 2, 0.004, seed=1234, first_replica=0)
    gpu_ctx.synth_replicas_ptr(d_base.data_ptr(), L, out2.data_ptr() + 2 * 8 * stride, stride, 4, 0.004, seed=1234, first_replica=2)
    gpu_ctx.batch_sync()
    assert torch.equal(out, out2)
    nz = (a - base[None, :]) / np.float32(0.004)
    assert abs(nz.real.mean()) < 0.01 and abs(nz.real.std() - 1) < 0.01 and abs(nz.imag.std() - 1) < 0.01
    assert abs(np.corrcoef(nz[0].real, nz[1].real)[0, 1]) < 0.01
    gpu_ctx.batch_process_ptr(out.data_ptr(), stride, L, 0, want_scores=False)
    gpu_ctx.batch_sync()
    st = gpu_ctx.batch_stats()
    assert (st["n_epc_correct"] == 70).all() and (st["tag_reads"][:, 0x27] == 70).all()


def test_file_ingest_batch_decoder(tmp_path, oracle_mod, synth_mod):
    """Trace files in the reference's format (interleaved float32 I,Q, apps/reader.py:102) ->
    pinned staging -> HBM -> one batched pass; ragged lengths."""
    import rfid
    traces = [synth_mod.make_trace(n_rounds=r, seed=300 + r, sigma=0.01).samples for r in (1, 3, 2)]
    paths = []
    for i, t in enumerate(traces):
        p = tmp_path / f"trace{i}.cf32"
        rfid.batch.write_trace_file(str(p), t)
        paths.append(str(p))
    assert np.array_equal(rfid.batch.read_trace_file(paths[1]), traces[1])
    dec = rfid.batch.BatchDecoder(device=0)
    try:
        timing = {}
        stats, w, r, s = dec.decode_files(paths, want_scores=True, timing=timing)
        for b, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, 3)):
            parity.compare_trace(wb, rb, sb, stats[b], oracle_mod.run_trace(traces[b]))
        summ = rfid.batch.summarize(stats)
        assert [x["n_epc_correct"] for x in summ] == [1, 3, 2] and summ[1]["tag_reads"] == {0x27: 3}
        assert timing["raw_samples"] == sum(map(len, traces))
        for b in range(3):   # the per-trace report text equals reader_impl::print_results as the oracle writes it
            assert rfid.batch.format_results(stats[b]) == oracle_mod.run_trace(traces[b]).print_results()
    finally:
        dec.close()
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert rfid.batch.main(paths) == 0          # python -m rfid.batch FILE...
    assert buf.getvalue().count("Correctly decoded EPC") == 3
