# developer aid (round 6): the long-stream front end (forced) on batches whose carrier leak puts one component of dc_est next to a
# power of two -- the sums hover across a binade edge, nothing settles by rounds, the finishing walk takes the units -- for trace
# counts that give the walk 512 / a few dozen / sixteen / fewer than sixteen waves per trace (one window per unit there) -- against
# the oracle: windows, dc_est, scores, statistics.
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "gen2-uhf-rfid-reader_amd"); sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.is_available()
import rfid, parity
from rfid import synth
from oracle import oracle

def run(seed):
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 2, 3, 17, 40, 64, 70, 130]))
    sigma = float(rng.choice([0.01, 0.03, 0.06, 0.08]))
    fixed_q = int(rng.integers(0, 3))
    # 25 |sin phi| (or cos) within a few per cent of a power of two
    target = float(rng.choice([4.0, 8.0, 16.0])) * (1.0 + float(rng.uniform(-0.012, 0.012)))
    phi = float(np.arcsin(target / 25.0)) if rng.random() < 0.5 else float(np.arccos(target / 25.0))
    if rng.random() < 0.5: phi = -phi
    leak = complex(np.exp(1j * phi))
    n_rounds = int(rng.integers(12, 60)) if B <= 3 else int(rng.integers(2, 7))
    tags = tuple(int(x) for x in rng.choice(np.arange(1, 200), size=int(rng.integers(1, 3)), replace=False))
    traces = [synth.make_trace(n_rounds=n_rounds, seed=int(rng.integers(1, 1 << 30)), sigma=sigma, fixed_q=fixed_q, tag_ids=tags,
                               t1_jitter_raw=int(rng.integers(0, 6)), leak=leak).samples for _ in range(B)]
    lens = [int(len(t) - rng.integers(0, min(len(t) // 2, 40000))) if rng.random() < 0.3 else len(t) for t in traces]
    L = max(map(len, traces))
    stride = (L + 1) & ~1
    host = np.zeros((B, stride), dtype=np.complex64)
    for i, t in enumerate(traces):
        host[i, : len(t)] = t
    dev = torch.from_numpy(host.view(np.float32)).to("cuda:0")
    d_lens = torch.tensor(np.asarray(lens, dtype=np.int64)).to("cuda:0")
    cfg = oracle.config(fixed_q=fixed_q, max_num_queries=1 << 30)
    refs = [oracle.run_trace(host[b, : lens[b]], cfg) for b in range(B)]
    out = []
    for dc_rounds in (None, "0"):      # the rounds as enqueued; the walk alone
        if dc_rounds is None: os.environ.pop("RFID_LS2_DC_ROUNDS", None)
        else: os.environ["RFID_LS2_DC_ROUNDS"] = dc_rounds
        ctx = rfid.Context(device=0, fixed_q=fixed_q, max_num_queries=1 << 30)
        try:
            ctx.batch_set_long_stream(2)
            ctx.batch_plan(B, L)
            ctx.batch_process_ptr(dev.data_ptr(), stride, L, d_lens.data_ptr(), want_scores=True)
            ctx.batch_sync()
            w, r, s = ctx.batch_windows(want_scores=True)
            st = ctx.batch_stats()
            for b, (wb, rb, sb) in enumerate(parity.split_by_stream(w, r, s, B)):
                parity.compare_trace(wb, rb, sb, st[b], refs[b])
            rep = ctx.batch_ls_report()
            out.append((B, sigma, rep["verified"], rep["gave_up"], rep["dc_finished"], rep["units"]))
        finally:
            ctx.close()
    return out

ok = bad = 0
rows = []
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for seed in range(first, first + count):
    try:
        rows += run(seed); ok += 1
    except Exception as e:
        bad += 1; print("seed", seed, "FAILED:", repr(e)[:300], flush=True)
os.environ.pop("RFID_LS2_DC_ROUNDS", None)
print("passed %d failed %d" % (ok, bad))
for B in sorted({r[0] for r in rows}):
    rr = [r for r in rows if r[0] == B]
    why = {}
    for r in rr:
        if r[3]: why[r[3]] = why.get(r[3], 0) + 1
    print("  traces %3d  passes %3d  verified %3d  gave up by reason %s  the walk engaged in %3d passes (%d of %d units)" %
          (B, len(rr), sum(r[2] for r in rr), why, sum(1 for r in rr if r[4] > 0), sum(r[4] for r in rr), sum(r[5] for r in rr)))
sys.exit(1 if bad else 0)
