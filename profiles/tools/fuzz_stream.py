# developer aid: rfid_stream_work with random call sizes and noise up to 8 % (passes that end in the sequential scan,
# escalation to the full round count) against the oracle -- windows, dc_est, decoded fields, report
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "gen2-uhf-rfid-reader_amd"); sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.is_available()
import rfid, parity
from rfid import synth
from oracle import oracle

def run(seed, sigma, lo, hi):
    rng = np.random.default_rng(seed)
    t = synth.make_trace(n_rounds=70, fixed_q=1, tag_ids=(0x27, 0x3C), seed=500 + seed, sigma=sigma, t1_jitter_raw=5).samples
    o = oracle.run_trace(t, oracle.config(fixed_q=1))
    ctx = rfid.Context(device=0, fixed_q=1)
    try:
        ctx.stream_begin(hi + 1000)
        ws, rs = [], []
        pos = 0
        while pos < len(t):
            n = min(int(rng.integers(lo, hi + 1)), len(t) - pos)
            w, r = ctx.stream_work(t[pos:pos + n])
            ws.append(w); rs.append(r)
            pos += n
        w, r = ctx.stream_work(flush=True)
        ws.append(w); rs.append(r)
        w, r = np.concatenate(ws), np.concatenate(rs)
        assert len(w) == o.n_windows, (len(w), o.n_windows)
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

ok = bad = 0
for seed in range(45):
    sigma = (0.01, 0.05, 0.08)[seed % 3]
    lo, hi = ((60000, 400000), (30000, 1200000), (25000, 60000))[(seed // 3) % 3]
    try:
        run(seed, sigma, lo, hi)
        ok += 1
    except Exception as e:
        bad += 1
        print("seed", seed, "sigma", sigma, "calls", (lo, hi), "FAILED:", repr(e)[:300])
print("passed", ok, "failed", bad)
