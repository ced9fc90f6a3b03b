"""CPU-only check of the KERNEL SOURCE (rfid_kernels.hpp) against the oracle: the kernels are
compiled for the host against tests/wave_emu (a lock-step 64-lane emulator, test
infrastructure only) and must reproduce the oracle bit-for-bit.  The product itself has no
CPU path; the real parity gate is tests/test_gpu_parity.py on the MI355X."""
import numpy as np
import pytest

import parity


def _check(emu_mod, oracle_mod, raw2d, lens=None, cfg_kw=None):
    cfg_kw = cfg_kw or {}
    r = emu_mod.batch_process(raw2d, lens=lens, **cfg_kw)
    B = raw2d.shape[0]
    for b, (wb, rb, sb) in enumerate(parity.split_by_stream(r["windows"], r["results"], r["scores"], B)):
        n = raw2d.shape[1] if lens is None else lens[b]
        o = oracle_mod.run_trace(raw2d[b, :n], oracle_mod.config(**cfg_kw))
        parity.compare_trace(wb, rb, sb, r["stats"][b], o)
    return r


@pytest.mark.parametrize("sigma,seed", [(0.002, 1), (0.06, 3)])
def test_emulated_kernels_match_oracle(emu_mod, oracle_mod, synth_mod, sigma, seed):
    traces = [synth_mod.make_trace(n_rounds=2, sigma=sigma, seed=seed * 10 + i, t1_jitter_raw=6).samples
              for i in range(2)]
    L = min(map(len, traces))
    _check(emu_mod, oracle_mod, np.stack([t[:L] for t in traces]))


def test_emulated_matched_filter_edges(emu_mod, oracle_mod, synth_mod):
    t = synth_mod.make_trace(n_rounds=1, sigma=0.05, seed=5).samples
    for L in (2600, 2561, 25, 7, 5, 4):
        r = emu_mod.batch_process(t[:L][None, :], want_y=True)
        yo = oracle_mod.fir(t[:L])
        assert np.array_equal(r["y"][0].view(np.uint32), yo.view(np.uint32)), L


def test_emulated_ragged_and_empty(emu_mod, oracle_mod, synth_mod):
    a = synth_mod.make_trace(n_rounds=2, seed=21).samples
    b = synth_mod.make_trace(n_rounds=1, seed=22).samples
    L = len(a)
    raw = np.zeros((4, L), dtype=np.complex64)
    raw[0] = a
    raw[1, : len(b)] = b
    raw[2, : len(a) * 3 // 4] = a[: len(a) * 3 // 4]   # cut inside the second round
    lens = [L, len(b), len(a) * 3 // 4, 0]
    r = _check(emu_mod, oracle_mod, raw, lens=lens)
    assert r["stats"][3]["n_windows"] == 0


def test_emulated_fixed_q2_collisions(emu_mod, oracle_mod, synth_mod):
    t = synth_mod.make_trace(n_rounds=1, fixed_q=2, tag_ids=(0x11, 0x22, 0x33), seed=33, sigma=0.01).samples
    _check(emu_mod, oracle_mod, t[None, :], cfg_kw=dict(fixed_q=2))


def test_emulated_termination(emu_mod, oracle_mod, synth_mod):
    t = synth_mod.make_trace(n_rounds=4, seed=41).samples
    r = emu_mod.batch_process(t[None, :], max_num_queries=2)
    o = oracle_mod.run_trace(t, oracle_mod.config(max_num_queries=2))
    st = r["stats"][0]
    assert st["status"] == 1 == o.state.status
    assert st["n_windows_used"] == o.n_windows
    for k in ("n_queries_sent", "cur_inventory_round", "n_epc_correct"):
        assert st[k] == getattr(o.state, k)


def test_emulated_streaming_gate_chunks(emu_mod, oracle_mod, synth_mod):
    """gate_scan_kernel in streaming mode with odd chunk sizes: consumed / written / samples
    equal the oracle's gate block call by call."""
    import ctypes as C
    t = synth_mod.make_trace(n_rounds=2, seed=61, sigma=0.01).samples
    y = oracle_mod.fir(t)
    L = oracle_mod.lib()
    g = (C.c_char * 8192)()
    rs = oracle_mod.ReaderState()
    cfg = oracle_mod.config()
    L.orc_gate_init(C.byref(g), 400000)
    L.orc_initialize_reader_state(C.byref(rs), C.byref(cfg))
    rs.gate_status = 2   # SEEK_RN16
    gs = emu_mod.GateStream()
    pos, seek, typ = 0, 0, 0
    chunk_sizes = [777, 64, 1, 63, 1500, 129]
    k = 0
    n_closed = 0
    while pos < len(y):
        n = min(chunk_sizes[k % len(chunk_sizes)], len(y) - pos)
        k += 1
        blk = np.ascontiguousarray(y[pos:pos + n])
        out_o = np.zeros(n, dtype=np.complex64)
        cons_o = C.c_int(0)
        wr_o = L.orc_gate_work(C.byref(g), C.byref(rs), C.c_void_p(blk.ctypes.data), n,
                               C.c_void_p(out_o.ctypes.data), C.byref(cons_o))
        cons_e, out_e, open_e = gs.work(blk, seek_type=seek)
        seek = -1
        assert cons_e == cons_o.value and len(out_e) == wr_o, (pos, cons_e, cons_o.value, len(out_e), wr_o)
        assert np.array_equal(out_e.view(np.uint32), out_o[:wr_o].view(np.uint32))
        assert open_e == (1 if rs.gate_status == 0 else 0)
        pos += cons_o.value
        if wr_o and rs.gate_status == 1:      # window closed: decoder+reader re-arm the gate
            typ ^= 1
            rs.gate_status = 3 if typ else 2
            seek = typ
            n_closed += 1
    assert n_closed == 4


def test_emulated_primitives(emu_mod):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(64) * 1e-3).astype(np.float32)
    num = (rng.standard_normal(64) * 30).astype(np.float32)
    den = np.where(np.arange(64) % 2 == 0, 100.0, 48.0).astype(np.float32)
    chain, div, hyp, shr = emu_mod.selftest(x, num, den, 12.5)
    acc = np.float32(12.5)
    for i in range(64):
        acc = np.float32(acc + x[i])
        assert chain[i] == acc
    assert np.array_equal(div, num / den)
    assert np.array_equal(shr[1:], x[:-1]) and shr[0] == 0


@pytest.mark.parametrize("chunk", [512, 1536, 777])
def test_emulated_time_chunked_gate(emu_mod, oracle_mod, synth_mod, chunk):
    """rfid_batch_process() cuts the traces along time and carries the gate state from chunk
    to chunk (so the gate scan can overlap the matched filter): same result as one launch."""
    t = synth_mod.make_trace(n_rounds=2, seed=71, sigma=0.02, t1_jitter_raw=5).samples
    raw = np.stack([t, np.roll(t, 3)])
    r = emu_mod.batch_process(raw, gate_chunk=chunk)
    for b, (wb, rb, sb) in enumerate(parity.split_by_stream(r["windows"], r["results"], r["scores"], 2)):
        parity.compare_trace(wb, rb, sb, r["stats"][b], oracle_mod.run_trace(raw[b]))


@pytest.mark.parametrize("unaligned", [False, True])
def test_emulated_fused_front_end(emu_mod, oracle_mod, synth_mod, unaligned):
    """front_end_fused_kernel (what rfid_batch_process() launches by default): the gate's
    producer waves run the matched filter from the raw samples.  Same windows, decisions, scores
    and matched-filter output as the oracle, for 16-byte aligned rows (float4 loads) and 8-byte
    aligned rows, ragged lengths included."""
    t = synth_mod.make_trace(n_rounds=3, seed=91, sigma=0.02, t1_jitter_raw=7).samples
    L = len(t)
    raw = np.stack([t, np.roll(t, 5), t * np.float32(0.5)])
    lens = np.array([L, L - 1237, L - 20001], dtype=np.int64)
    r = emu_mod.batch_process(raw, lens=lens, gate_chunk=-1, want_y=True, unaligned=unaligned)
    for b, (wb, rb, sb) in enumerate(parity.split_by_stream(r["windows"], r["results"], r["scores"], 3)):
        o = oracle_mod.run_trace(raw[b][: lens[b]])
        parity.compare_trace(wb, rb, sb, r["stats"][b], o)
        n = int(lens[b]) // 5
        assert np.array_equal(r["y"][b][:n].view(np.uint32), oracle_mod.fir(raw[b][: lens[b]])[:n].view(np.uint32))


def test_emulated_replica_generator(emu_mod):
    """synth_replicas_kernel (workload generator): Philox4x32-10 reproduces the Random123 known-answer
    vectors; the noise is a pure function of (seed, replica, sample) -- generating in pieces gives the
    same bytes -- with unit-variance independent components."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert [int(v) for v in emu_mod.philox4x32_10(ctr, key)] == want
    base = (np.arange(4099) * (0.25 + 0.5j)).astype(np.complex64)          # odd length: the tail sample
    a = emu_mod.synth_replicas(base, 4, 0.5, seed=99)
    b = np.concatenate([emu_mod.synth_replicas(base, 1, 0.5, seed=99, first_replica=0),
                        emu_mod.synth_replicas(base, 3, 0.5, seed=99, first_replica=1)])
    assert a.tobytes() == b.tobytes()
    assert emu_mod.synth_replicas(base, 1, 0.5, seed=100).tobytes() != a[:1].tobytes()
    nz = (a - base[None, :]) / np.float32(0.5)
    assert abs(nz.real.mean()) < 0.03 and abs(nz.imag.mean()) < 0.03
    assert abs(nz.real.std() - 1) < 0.03 and abs(nz.imag.std() - 1) < 0.03
    assert abs(np.corrcoef(nz[0].real, nz[1].real)[0, 1]) < 0.06 and abs(np.corrcoef(nz[0].real, nz[0].imag)[0, 1]) < 0.06
    assert np.array_equal(emu_mod.synth_replicas(base, 2, 0.0, seed=1), np.stack([base, base]))


@pytest.mark.parametrize("kw", [dict(n_rounds=2, fixed_q=0, tag_ids=(0x27,), seed=3),
                                dict(n_rounds=2, fixed_q=2, tag_ids=(1, 2, 3, 4, 5, 6), seed=8, t1_jitter_raw=7,
                                     corrupt_rounds=(1,)),
                                dict(n_rounds=1, fixed_q=4, tag_ids=tuple(range(9)), seed=2, leak=-0.8 + 0.3j)])
def test_emulated_gen2_synthesiser_equals_numpy_generator(emu_mod, synth_mod, kw):
    """synth_gen2_kernel (reader PIE envelope incl. Query CRC-5, ACKs, FM0 backscatter with
    collisions and empty slots, built on the device from a slot table) equals rfid/synth.py sample for sample on the
    noise-free part; with noise it is the noise-free trace plus the replica generator's noise."""
    t = synth_mod.make_trace(sigma=0.0, noise=False, **kw)
    x = emu_mod.synth_gen2(t.plan)
    assert len(x) == len(t.samples)
    assert np.array_equal(x, t.samples)
    if "leak" not in kw:
        assert np.array_equal(x.view(np.uint32), t.samples.view(np.uint32))
    xn = emu_mod.synth_gen2(t.plan, sigma=0.01, seed=77, replica=5)
    want = emu_mod.synth_replicas(t.samples, 1, 0.01, seed=77, first_replica=5)[0]
    assert np.array_equal(xn.view(np.uint32), want.view(np.uint32))


def test_emulated_chain_scan_is_exact(emu_mod):
    """chain_add_scan (the in-order binary32 sum as ONE integer prefix sum on the mantissa, taken only when every
    partial sum provably stays inside the carry's binade and no addend is a rounding tie) equals the 63-deep
    sequential chain bit for bit -- on random data of the receive path's magnitudes, negative carries, exact ties,
    partial sums that cross a power of two, zero / tiny / huge carries -- and does apply in the common case."""
    rng = np.random.default_rng(11)
    n_clean = 0
    cases = []
    for _ in range(300):
        carry = float(rng.choice([23.456789, -19.12345, 31.99999, 16.000002, 0.0, 1e-30, -15.99999, 3.0e5, 25.0, 8.5, -0.75]))
        scale = float(rng.choice([1e-4, 1e-3, 1e-2, 0.3, 1e-8, 1e3]))
        x = (rng.standard_normal(64) * scale).astype(np.float32)
        kind = rng.integers(0, 4)
        if kind == 1:
            x[::3] = np.ldexp(rng.integers(-7, 8, len(x[::3])).astype(np.float32) + 0.5, -19)   # half-ulp at 16..32
        if kind == 2:
            x[rng.integers(0, 64)] = 0.0
            x[5] = -0.0
        cases.append((x, carry))
    cases.append((np.zeros(64, np.float32), 25.0))
    cases.append((np.full(64, 2.0 ** -19, np.float32), 31.9999))      # walks up to and across 32
    cases.append((np.full(64, -2.0 ** -20, np.float32), 16.00001))    # walks down across 16: ties and a binade edge
    for x, carry in cases:
        chain, scan, clean = emu_mod.chain_scan(x, carry)
        acc = np.float32(carry)
        for i in range(64):
            acc = np.float32(acc + x[i])
            assert chain[i].view(np.uint32) == acc.view(np.uint32)
        assert np.array_equal(scan.view(np.uint32), chain.view(np.uint32)), (carry, clean)
        n_clean += clean
    assert n_clean > 60       # the scan path is exercised, not just the fallback


@pytest.mark.parametrize("kw", [dict(max_num_queries=3), dict(number_unique_tags=1), dict(number_unique_tags=0),
                                dict(max_num_queries=1, number_unique_tags=2), dict(max_num_queries=0)])
def test_emulated_termination_limits(emu_mod, oracle_mod, synth_mod, kw):
    """stream_stats_kernel finds the TERMINATED cut-off (gate_impl.cc:101-109) without replaying the windows one by
    one: queries limit, distinct-tag limit, both, and limits that stop the run before its first window."""
    t = synth_mod.make_trace(n_rounds=2, fixed_q=3, tag_ids=(0x21, 0x43, 0x65), seed=78, sigma=0.01).samples
    assert oracle_mod.run_trace(t, oracle_mod.config(fixed_q=3)).state.n_unique_tags == 3     # the limits below do cut the run
    r = emu_mod.batch_process(t[None, :], fixed_q=3, **kw)
    o = oracle_mod.run_trace(t, oracle_mod.config(fixed_q=3, **kw))
    st = r["stats"][0]
    assert st["status"] == o.state.status == 1
    assert st["n_windows_used"] == o.n_windows
    for k in ("n_queries_sent", "cur_inventory_round", "cur_slot_number", "n_epc_correct", "n_unique_tags"):
        assert st[k] == getattr(o.state, k), k
    assert np.array_equal(st["tag_reads"], np.array(o.state.tag_reads[:], dtype=np.int32))
