"""GPU tests added in round 2: robustness of the batch workspace, decoder reuse, the real-blob replay hook."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.gpu
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

README_BLOCK = ("| Number of queries/queryreps sent : 71\n| Current Inventory round : 72\n --------------------------\n"
                "| Correctly decoded EPC : 70\n| Number of unique tags : 1\n| Tag ID : 27  Num of reads : 70\n")


def _real_blob():
    """misc/data/file_source_test is missing from the reference checkout (.MISSING_LARGE_BLOBS:2).  The first
    environment that has it pins parity: point RFID_FILE_SOURCE_TEST at it or drop it into tests/data/."""
    for p in (os.environ.get("RFID_FILE_SOURCE_TEST", ""),
              os.path.join(os.path.dirname(__file__), "data", "file_source_test")):
        if p and os.path.isfile(p) and os.path.getsize(p) > 1 << 20:
            return p
    return None


def test_failed_plan_leaves_context_unplanned(gpu_ctx):
    """rfid_batch_plan that cannot allocate (a matched-filter output far beyond HBM) must not leave a half-planned
    context: the next rfid_batch_process returns RFID_ERR_STATE instead of launching on null pointers."""
    import rfid
    import torch
    lib = rfid.capi.load()
    st = lib.rfid_batch_plan(gpu_ctx._h, 60000, 2_000_000_000)        # ~192 TB of workspace
    assert st in (rfid.capi.ERR_HIP, rfid.capi.ERR_UNSUPPORTED)
    dummy = torch.zeros(64, dtype=torch.float32, device="cuda:0")
    st2 = lib.rfid_batch_process(gpu_ctx._h, C.c_void_p(dummy.data_ptr()), 16, 16, None, 0)
    assert st2 == rfid.capi.ERR_STATE
    n = C.c_int64(0)
    assert lib.rfid_batch_get_windows(gpu_ctx._h, None, None, None, 0, C.byref(n)) == rfid.capi.ERR_STATE
    gpu_ctx.batch_plan(1, 4096)                                       # and the context is still usable
    assert lib.rfid_batch_set_streams(gpu_ctx._h, 2) == rfid.capi.ERR_CAPACITY


def test_batch_decoder_reuse_with_smaller_batch(oracle_mod, synth_mod):
    """One BatchDecoder, first 3 traces then 1: the second call must process exactly its own row (no phantom
    windows from the stale rows of the larger plan, no out-of-bounds read of the 1-entry lens array)."""
    import rfid
    traces = [synth_mod.make_trace(n_rounds=r, seed=400 + r, sigma=0.01).samples for r in (2, 3, 1)]
    dec = rfid.batch.BatchDecoder(device=0)
    try:
        stats, w, r, s = dec.decode(traces, want_scores=True)
        assert len(stats) == 3 and set(w["stream"]) == {0, 1, 2}
        small = synth_mod.make_trace(n_rounds=2, seed=499, sigma=0.02).samples
        stats, w, r, s = dec.decode([small], want_scores=True)
        assert len(stats) == 1 and set(w["stream"]) == {0}
        parity.compare_trace(w, r, s, stats[0], oracle_mod.run_trace(small))
        stats, w, r, s = dec.decode(traces[:2], want_scores=True)      # and growing again inside the plan
        for b, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, 2)):
            parity.compare_trace(wb, rb, sb, stats[b], oracle_mod.run_trace(traces[b]))
    finally:
        dec.close()


def test_streaming_matched_filter_decimation_phase(gpu_ctx, oracle_mod, synth_mod):
    """rfid_mf_work over ragged call sizes: floor(N/5) outputs in total (a decimator emits y[n] once the group
    x[5n..5n+4] is complete), bit-identical to the batch filter / oracle on the same samples."""
    t = synth_mod.make_trace(n_rounds=1, sigma=0.05, seed=6).samples[:7013]
    for sizes in ([7013], [1, 2, 3, 4, 5, 6, 7, 1000, 3, 5982], [4] * 3 + [7001], [2048, 2048, 2917]):
        gpu_ctx.reset()
        out, pos = [], 0
        for n in sizes:
            out.append(gpu_ctx.mf_work(t[pos:pos + n]))
            pos += n
            assert sum(map(len, out)) == pos // 5
        y = np.concatenate(out)
        assert np.array_equal(y.view(np.uint32), oracle_mod.fir(t[:pos]).view(np.uint32)), sizes


def test_unfused_mf_launch_beyond_65535_traces(gpu_ctx, oracle_mod, synth_mod):
    """The stage kernel's trace index rides on gridDim.y (<= 65535): larger batches are launched in slices."""
    import torch
    t = synth_mod.make_trace(n_rounds=1, sigma=0.02, seed=8).samples[:640]
    B, L = 65535 + 70, len(t)
    host = np.tile(t, (B, 1))
    host[-1] *= np.float32(0.5)
    dev = torch.from_numpy(host.view(np.float32)).to("cuda:0")
    gpu_ctx.batch_plan(B, L)
    gpu_ctx.batch_stage("mf", dev.data_ptr(), L, L, 0)
    gpu_ctx.batch_sync()
    for b in (0, 65534, 65535, B - 1):
        y = gpu_ctx.batch_mf_output(b)
        assert np.array_equal(y.view(np.uint32), oracle_mod.fir(host[b]).view(np.uint32)), b
    gpu_ctx.batch_plan(1, 4096)   # release the large workspace


@pytest.mark.skipif(_real_blob() is None, reason="the reference's misc/data/file_source_test is not available "
                    "(.MISSING_LARGE_BLOBS:2); set RFID_FILE_SOURCE_TEST or add tests/data/file_source_test")
def test_real_file_source_test_replay_matches_readme():
    """README.md:48-53: 71 queries / round 72 / 70 EPC / 1 unique tag / Tag ID 27 x 70 on the bundled trace,
    through the C++ offline flowgraph (rfid_reader_offline) AND the batched path -- the pin for parity."""
    import rfid
    exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
    out = subprocess.run([exe, _real_blob()], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert README_BLOCK in out.stdout, out.stdout
    dec = rfid.batch.BatchDecoder(device=0)
    try:
        stats, _, _, _ = dec.decode_files([_real_blob()])
        assert README_BLOCK in rfid.batch.format_results(stats[0])
    finally:
        dec.close()


def _gen2_on_device(ctx, plan, sigma=0.0, seed=0, replica=0):
    import torch
    n = ctx.synth_gen2_size(plan)
    assert n == plan.n_raw
    out = torch.zeros(2 * n, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    assert ctx.synth_gen2_ptr(plan, out.data_ptr(), n, sigma=sigma, seed=seed, replica=replica) == n
    ctx.batch_sync()
    return out


@pytest.mark.parametrize("kw", [dict(n_rounds=3, fixed_q=0, tag_ids=(0x27,), seed=3, corrupt_rounds=(2,)),
                                dict(n_rounds=2, fixed_q=3, tag_ids=(1, 2, 3, 4, 5, 6), seed=8, t1_jitter_raw=7),
                                dict(n_rounds=1, fixed_q=4, tag_ids=tuple(range(9)), seed=2)])
def test_device_gen2_synthesiser_equals_numpy_generator(gpu_ctx, oracle_mod, synth_mod, kw):
    """rfid_synth_gen2 (f4: device-side PIE / CRC-5 / FM0 generator) == rfid/synth.py sample for sample at
    sigma = 0; with noise == base + rfid_synth_replicas noise; and the generated trace decodes to the slot
    table's truth through the batched path, bit-identical to the oracle on the same samples."""
    import rfid
    import torch
    t = synth_mod.make_trace(sigma=0.0, noise=False, **kw)
    out = _gen2_on_device(gpu_ctx, t.plan)
    x = out.cpu().numpy().view(np.complex64)
    assert np.array_equal(x.view(np.uint32), t.samples.view(np.uint32))
    # noisy: same bytes as the noise-free trace + the replica generator
    outn = _gen2_on_device(gpu_ctx, t.plan, sigma=0.01, seed=5, replica=3)
    L = len(x)
    ref = torch.zeros(2 * L, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    gpu_ctx.synth_replicas_ptr(out.data_ptr(), L, ref.data_ptr(), L, 1, 0.01, 5, first_replica=3)
    gpu_ctx.batch_sync()
    assert torch.equal(outn, ref)
    # decode it
    q = kw["fixed_q"]
    ctx = rfid.Context(device=0, fixed_q=q)
    try:
        ctx.batch_plan(1, L)
        ctx.batch_process_ptr(outn.data_ptr(), L, L, 0, want_scores=True)
        ctx.batch_sync()
        w, r, s = ctx.batch_windows(want_scores=True)
        st = ctx.batch_stats()
        xn = outn.cpu().numpy().view(np.complex64)
        parity.compare_trace(w, r, s, st[0], oracle_mod.run_trace(xn, oracle_mod.config(fixed_q=q)))
        n_valid = sum(1 for sl in t.slots if sl.epc_valid)
        assert st[0]["n_epc_correct"] == n_valid
        # every single-responder slot's RN16 comes back exactly
        rn = [rfid.unpack_bits(rr["bits"], 16).tolist() for rr in r[r["type"] == 0]]
        for i, sl in enumerate(t.slots):
            if sl.n_tags == 1:
                assert rn[i] == sl.rn16, i
    finally:
        ctx.close()


def test_gen2_slot_table_validation(gpu_ctx, synth_mod):
    import rfid
    import torch
    lib = rfid.capi.load()
    t = synth_mod.make_trace(n_rounds=1, render=False)
    buf = torch.zeros(2 * t.plan.n_raw, dtype=torch.float32, device="cuda:0")
    with pytest.raises(rfid.capi.RfidError):            # capacity
        gpu_ctx.synth_gen2_ptr(t.plan, buf.data_ptr(), t.plan.n_raw - 2)
    bad = synth_mod.make_trace(n_rounds=1, render=False).plan
    bad.slots["rn16_off_raw"] = 2 * 1295               # reply would run into the ACK
    with pytest.raises(rfid.capi.RfidError):
        gpu_ctx.synth_gen2_ptr(bad, buf.data_ptr(), t.plan.n_raw)
    with pytest.raises(rfid.capi.RfidError):            # unaligned output
        gpu_ctx.synth_gen2_ptr(t.plan, buf.data_ptr() + 8, t.plan.n_raw)


def test_cxx_reference_block_api_offline_run(tmp_path, oracle_mod, synth_mod):
    """rfid_reader_offline is written against the reference's block API only (gate::make(int), tag_decoder::make(int),
    reader::make(int,int), print_results(), the reader_state global) in apps/reader.py:75-78 order: same report as the
    oracle, reader_state mirrors the stream, and the per-stage debug taps (apps/reader.py:67-72: matched_filter and
    gate file sinks) equal the oracle's matched filter / the gated windows of the batch path."""
    import rfid
    exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
    t = synth_mod.make_trace(n_rounds=3, seed=21, sigma=0.01, t1_jitter_raw=4)
    path, mfp, gp = tmp_path / "t.bin", tmp_path / "mf.c64", tmp_path / "gate.c64"
    rfid.batch.write_trace_file(str(path), t.samples)
    env = dict(os.environ, RFID_PRINT_READER_STATE="1")
    out = subprocess.run([exe, str(path), "--mf-out", str(mfp), "--gate-out", str(gp), "--chunk", "1000"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    o = oracle_mod.run_trace(t.samples)
    assert out.stdout.startswith(o.print_results())
    assert "reader_state: n_queries_sent=%d n_epc_correct=3 unique=1" % o.state.n_queries_sent in out.stdout
    assert "windows=6" in out.stdout
    y = np.fromfile(str(mfp), dtype=np.complex64)
    assert np.array_equal(y.view(np.uint32), oracle_mod.fir(t.samples).view(np.uint32))
    g = np.fromfile(str(gp), dtype=np.complex64)
    # gated samples = in[i] - dc_est over each window (gate_impl.cc:176,187)
    want = np.concatenate([(y[s:s + (1370 if k & 1 else 250)] - np.complex64(dc)).astype(np.complex64)
                           for k, (s, dc) in enumerate(zip(o.open_idx, o.dc))])
    assert len(g) >= len(want) and np.array_equal(g[: len(want)].view(np.uint32), want.view(np.uint32))


def test_python_blocks_bind_per_flowgraph(oracle_mod, synth_mod):
    """Two flowgraphs in one process, blocks built in the reference's order (matched filter BEFORE the gate,
    apps/reader.py:75-76) and interleaved: each tag_decoder / reader / matched_filter belongs to its own gate's
    stream -- no shared or stolen state."""
    import rfid
    ta = synth_mod.make_trace(n_rounds=2, seed=31, sigma=0.01)
    tb_ = synth_mod.make_trace(n_rounds=3, seed=32, sigma=0.01, tag_ids=(0x42,))
    a = rfid.reader_top_block(samples=ta.samples, device=0)
    b = rfid.reader_top_block(samples=tb_.samples, device=0, chunk=3000)
    try:
        assert a.ctx is not b.ctx and a.matched_filter.ctx is a.ctx and b.reader.ctx is b.ctx
        b.run()
        a.run()
        assert a.ctx.stats() == oracle_mod.run_trace(ta.samples).stats()
        assert b.ctx.stats() == oracle_mod.run_trace(tb_.samples).stats()
        with pytest.raises(RuntimeError):
            rfid.matched_filter().work(ta.samples[:100])      # a filter whose gate does not exist yet
        rfid.blocks._bind.pending_filters.clear()
    finally:
        a.ctx.close()
        b.ctx.close()


TWO_RANK_WORKER = r"""
import os, sys, json
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gen2-uhf-rfid-reader_amd"))
import torch, torch.distributed as dist
import rfid
from rfid import shard, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = sys.argv[2]
torch.cuda.set_device(0)                      # both ranks on the one GPU of the box
dist.init_process_group(backend, rank=rank, world_size=world)
N, sigma, seed = 10, 0.004, 4321
t = synth.fst_like_trace(sigma=0.0)           # noise-free base; replica g = base + noise(seed, g)
plan = synth.make_trace(n_rounds=71, fixed_q=0, tag_ids=(0x27,), sigma=0.0, seed=7, corrupt_rounds=(36,), noise=False, render=False).plan
b, e = shard.partition(N, world, rank)
ctx = rfid.Context(device=0)                  # one context (one HIP stream) per rank
L = ctx.synth_gen2_size(plan); stride = (L + 1) & ~1
base = torch.zeros(2 * stride, dtype=torch.float32, device="cuda:0")
data = torch.zeros((e - b, 2 * stride), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
ctx.synth_gen2_ptr(plan, base.data_ptr(), stride)
ctx.synth_replicas_ptr(base.data_ptr(), L, data.data_ptr(), stride, e - b, sigma, seed, first_replica=b)
ctx.batch_plan(e - b, L)
ctx.batch_process_ptr(data.data_ptr(), stride, L, 0, want_scores=False)
ctx.batch_sync()
stats = ctx.batch_stats()
dist.barrier()
dev = torch.device("cuda:0") if backend == "nccl" else None
tot = shard.reduce_totals(shard.local_totals(stats), dist, device=dev)
first = data[0].cpu().numpy().view(np.complex64)[:L].copy() if rank == world - 1 else None
if rank == world - 1:
    np.save(sys.argv[3], first)
if rank == 0:
    print("TOTALS " + json.dumps(tot[:5].tolist() + [int(tot[5 + 0x27])]))
print("NATIVE " + str("librfid_mi355x.so" in open("/proc/self/maps").read()))
ctx.close()
dist.destroy_process_group()
"""


def test_two_ranks_shard_a_batch_through_the_hip_path(tmp_path, gpu_ctx, oracle_mod, synth_mod):
    """N > 1 with the HIP path under test: two ranks (both on this box's one GPU; RCCL refuses two ranks per
    device, so the totals reduction -- control plane only -- runs over gloo) each build their own context, generate
    and decode their rfid.shard.partition slice with rfid_batch_process, and the reduced totals equal the
    single-rank pass over all replicas and the oracle on one of them."""
    import json
    import sys
    import torch
    script = tmp_path / "worker.py"
    script.write_text(TWO_RANK_WORKER)
    first_path = str(tmp_path / "first.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT_DIR, "gloo", first_path], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("NATIVE True" in o for o in outs)
    tot = json.loads([l for l in outs[0].splitlines() if l.startswith("TOTALS ")][0][7:])
    # single-rank pass over all 10 replicas in this process
    from rfid import shard
    plan = synth_mod.make_trace(n_rounds=71, fixed_q=0, tag_ids=(0x27,), sigma=0.0, seed=7, corrupt_rounds=(36,),
                                noise=False, render=False).plan
    N = 10
    L = gpu_ctx.synth_gen2_size(plan)
    stride = (L + 1) & ~1
    base = torch.zeros(2 * stride, dtype=torch.float32, device="cuda:0")
    data = torch.zeros((N, 2 * stride), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    gpu_ctx.synth_gen2_ptr(plan, base.data_ptr(), stride)
    gpu_ctx.synth_replicas_ptr(base.data_ptr(), L, data.data_ptr(), stride, N, 0.004, 4321, first_replica=0)
    gpu_ctx.batch_plan(N, L)
    gpu_ctx.batch_process_ptr(data.data_ptr(), stride, L, 0, want_scores=False)
    gpu_ctx.batch_sync()
    single = shard.local_totals(gpu_ctx.batch_stats())
    assert tot == single[:5].tolist() + [int(single[5 + 0x27])]
    assert tot == [N, 2 * 71 * N, 70 * N, 72 * N, 0, 70 * N]
    # rank 1's first replica (index 5) is the same bytes as this process's row 5, and the oracle agrees on it
    first = np.load(first_path)
    mine = data[5].cpu().numpy().view(np.complex64)[:L]
    assert np.array_equal(first.view(np.uint32), mine.view(np.uint32))
    o = oracle_mod.run_trace(first)
    assert o.state.n_epc_correct == 70 and gpu_ctx.batch_stats()[5]["n_epc_correct"] == 70
    gpu_ctx.batch_plan(1, 4096)


@pytest.mark.parametrize("sizes", [[400_000], [150_000, 333_333, 123_457], [1_000_000]])
def test_whole_chain_streaming_equals_batch_and_oracle(oracle_mod, synth_mod, sizes):
    """rfid_stream_work: raw chunks in, decoded windows out, all block state carried on the device, upload of chunk
    k+1 overlapping the processing of chunk k.  Whatever the chunk sizes (also from the pinned staging buffers), the
    windows (global positions, dc_est), every decoded field and the final READER_STATE / print_results text equal
    the oracle's run over the whole stream."""
    import rfid
    t = synth_mod.make_trace(n_rounds=80, fixed_q=1, tag_ids=(0x27, 0x3C), seed=812, sigma=0.01, t1_jitter_raw=5).samples
    o = oracle_mod.run_trace(t, oracle_mod.config(fixed_q=1))
    ctx = rfid.Context(device=0, fixed_q=1)
    try:
        ctx.stream_begin(max(sizes) + 1000)
        ws, rs = [], []
        pos, k = 0, 0
        while pos < len(t):
            n = min(sizes[k % len(sizes)], len(t) - pos)
            if k % 2 == 1:                                   # every other chunk goes through a pinned staging buffer
                stg = ctx.stream_staging(k % 2)              # (the slot the library itself would stage chunk k in)
                stg[:n] = t[pos:pos + n]
                w, r = ctx.stream_work(stg[:n])
            else:
                w, r = ctx.stream_work(t[pos:pos + n])
            ws.append(w); rs.append(r)
            pos += n
            k += 1
        w, r = ctx.stream_work(flush=True)
        ws.append(w); rs.append(r)
        w, r = np.concatenate(ws), np.concatenate(rs)
        assert len(w) == o.n_windows, ([len(x) for x in ws], o.n_windows)
        assert np.array_equal(w["start"], o.open_idx) and np.array_equal(w["type"], o.dumps["type"])
        assert np.array_equal(w["dc_re"].view(np.uint32), o.dc.real.view(np.uint32))
        assert np.array_equal(w["dc_im"].view(np.uint32), o.dc.imag.view(np.uint32))
        fake = np.zeros(len(w), dtype=rfid.capi.WINDOW_DTYPE)
        fake["start"], fake["type"], fake["dc_re"], fake["dc_im"] = w["start"], w["type"], w["dc_re"], w["dc_im"]
        parity.compare_trace_fast(fake, r, None, o)
        assert ctx.stats() == o.stats()
        assert ctx.print_results() == o.print_results()
        ctx.stream_end()
    finally:
        ctx.close()


def test_cxx_offline_binary_whole_chain_mode(tmp_path, oracle_mod, synth_mod):
    """rfid_reader_offline --whole-chain N: the same blocks, fed through rfid_stream_work -- same report as block by
    block and as the oracle.  (print_results reads the context both modes share.)"""
    import rfid
    exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
    t = synth_mod.make_trace(n_rounds=40, seed=23, sigma=0.01, t1_jitter_raw=3, corrupt_rounds=(7,))
    path = tmp_path / "t.bin"
    rfid.batch.write_trace_file(str(path), t.samples)
    want = oracle_mod.run_trace(t.samples).print_results()
    for extra in ([], ["--whole-chain", "250000"], ["--whole-chain", "100000000"]):
        out = subprocess.run([exe, str(path), "--time"] + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert out.stdout == want, extra
        print(out.stderr.strip())


def test_batch_debug_taps(gpu_ctx, oracle_mod, synth_mod):
    """Per-stage debug sinks of apps/reader.py:67-72 in batch mode: the matched-filter output of a trace and the gated,
    DC-removed samples of any of its windows (what the gate block emits: in[i] - dc_est)."""
    import torch
    t = synth_mod.make_trace(n_rounds=3, seed=44, sigma=0.02).samples
    L = len(t)
    dev = torch.from_numpy(np.concatenate([t, np.zeros(L & 1, np.complex64)]).view(np.float32)).to("cuda:0")
    gpu_ctx.batch_plan(1, L)
    gpu_ctx.batch_process_ptr(dev.data_ptr(), (L + 1) & ~1, L, 0, want_scores=False)
    gpu_ctx.batch_sync()
    o = oracle_mod.run_trace(t)
    y = oracle_mod.fir(t)
    assert np.array_equal(gpu_ctx.batch_mf_output(0).view(np.uint32), y.view(np.uint32))
    for seq in range(o.n_windows):
        g = gpu_ctx.batch_gated_output(0, seq)
        n = 1370 if seq & 1 else 250
        want = (y[o.open_idx[seq]: o.open_idx[seq] + n] - np.complex64(o.dc[seq])).astype(np.complex64)
        assert len(g) == n and np.array_equal(g.view(np.uint32), want.view(np.uint32)), seq
    with pytest.raises(Exception):
        gpu_ctx.batch_gated_output(0, o.n_windows)


def test_long_stream_front_end_ragged_batch_and_limits(oracle_mod, synth_mod):
    """The long-stream front end on a small ragged batch (per-trace lengths, one trace too short to cut, one empty), with
    FIXED_Q = 2 collisions and a max_num_queries limit that terminates two of the traces: windows, results, scores and
    statistics equal the oracle's; the report says verified."""
    import rfid
    import torch
    kw = dict(fixed_q=2, tag_ids=(0x11, 0x22, 0x33), sigma=0.01, t1_jitter_raw=5)
    traces = [synth_mod.make_trace(n_rounds=r, seed=900 + r, **kw).samples for r in (9, 6, 1)]
    L = max(map(len, traces))
    raw = np.zeros((4, L), dtype=np.complex64)
    lens = []
    for i, t in enumerate(traces):
        raw[i, : len(t)] = t
        lens.append(len(t))
    lens.append(0)
    lens[1] -= 12345                                   # cut inside a slot: the last window is incomplete
    ctx = rfid.Context(device=0, fixed_q=2, max_num_queries=20)
    try:
        ctx.batch_set_long_stream(2)
        stride = (L + 1) & ~1
        host = np.zeros((4, stride), dtype=np.complex64)
        host[:, :L] = raw
        dev = torch.from_numpy(host.view(np.float32)).to("cuda:0")
        d_lens = torch.tensor(np.asarray(lens, dtype=np.int64)).to("cuda:0")
        ctx.batch_plan(4, L)
        ctx.batch_process_ptr(dev.data_ptr(), stride, L, d_lens.data_ptr(), want_scores=True)
        ctx.batch_sync()
        rep = ctx.batch_ls_report()
        assert rep["verified"] == 1 and rep["pieces"] > 4, rep
        w, r, s = ctx.batch_windows(want_scores=True)
        st = ctx.batch_stats()
        cfg = oracle_mod.config(fixed_q=2, max_num_queries=20)
        full = oracle_mod.config(fixed_q=2, max_num_queries=1 << 30)
        n_term = 0
        for b, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, 4)):
            o_all = oracle_mod.run_trace(raw[b, : lens[b]], full)        # every window the gate produces ...
            parity.compare_trace(wb, rb, sb, None, o_all)
            o = oracle_mod.run_trace(raw[b, : lens[b]], cfg)             # ... and the statistics up to the TERMINATED cut-off
            for k in ("n_queries_sent", "cur_inventory_round", "cur_slot_number", "n_epc_correct", "n_unique_tags"):
                assert st[b][k] == getattr(o.state, k), (b, k)
            assert st[b]["status"] == o.state.status and st[b]["n_windows_used"] == o.n_windows
            n_term += int(o.state.status)
        assert n_term == 2 and st[3]["n_windows"] == 0
    finally:
        ctx.close()


def test_streaming_terminates_like_the_blocks(oracle_mod, synth_mod):
    """rfid_stream_work with the reference's default MAX_NUM_QUERIES-style limit: after TERMINATED the gate swallows the
    rest of the stream (gate_impl.cc:101-109,125): no further windows are delivered, READER_STATE equals the oracle's."""
    import rfid
    t = synth_mod.make_trace(n_rounds=60, seed=5150, sigma=0.01).samples
    cfg = oracle_mod.config(max_num_queries=20)
    o = oracle_mod.run_trace(t, cfg)
    assert o.state.status == 1
    ctx = rfid.Context(device=0, max_num_queries=20)
    try:
        ctx.stream_begin(300_000)
        n_win = 0
        for pos in range(0, len(t), 300_000):
            w, r = ctx.stream_work(t[pos:pos + 300_000])
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
