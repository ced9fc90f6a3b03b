# developer aid: the per-block calls with look-ahead driven by a randomised scheduler (call sizes, re-asks, noise up to 8 %:
# passes that end in the sequential scan) against the oracle -- 60 cases, ~10 s on one MI355X
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "gen2-uhf-rfid-reader_amd"); sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.is_available()
import rfid
from rfid import synth
from oracle import oracle

def run(seed, sigma, lo, hi, rounds, foreign_filter=False):
    """foreign_filter: the matched filter is numpy on the host (rfid.flowgraph.fir_filter_ccc_ones), the look-ahead keyed on
    the gate's input; the gate is shown ragged views of its input buffer (unconsumed samples again, new ones behind them)"""
    rng = np.random.default_rng(seed)
    t = synth.make_trace(n_rounds=rounds, seed=100 + seed, sigma=sigma, fixed_q=1, tag_ids=(0x11, 0x2A), t1_jitter_raw=4,
                         corrupt_rounds=(7,)).samples
    o = oracle.run_trace(t, oracle.config(fixed_q=1))
    tb = rfid.reader_top_block(samples=t, chunk=hi // 5 + 1, lookahead=True, fixed_q=1, external_filter=foreign_filter)
    try:
        tb._reader_until_idle(0)
        gq = np.zeros(0, dtype=np.complex64); dq = np.zeros(0, dtype=np.complex64)
        src = rfid.flowgraph.fir_filter_ccc_ones(t) if foreign_filter else t
        per = 5 if foreign_filter else 1
        pos, n, idle, flushed = 0, len(src), 0, False
        while pos < n or len(gq):
            if pos < n:
                blk = src[pos:pos + max(1, int(rng.integers(lo, hi + 1)) // per)]
                pos += len(blk)
                y = blk if foreign_filter else tb.matched_filter.work(blk)
                gq = np.concatenate([gq, y]) if len(gq) else y
            while len(gq):
                take = gq if pos >= n else gq[: int(rng.integers(50, 30001))]   # (a scheduler shows everything at the end of the input)
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
                            break
                        idle += 1
                        if idle > 6:
                            idle = 0
                            break
                    elif not flushed:
                        tb.ctx.lookahead_flush(); flushed = True; idle = 0
                    else:
                        idle += 1                 # (keyed on the gate the flush is carried out on the third idle call)
                        if idle > 4:
                            gq = gq[:0]
                else:
                    idle = 0
        assert tb.ctx.stats() == o.stats(), "stats"
        assert tb.ctx.print_results() == o.print_results(), "report"
        assert len(tb.decoded) == o.n_windows, "windows"
        for (res, sc), d in zip(tb.decoded, o.dumps):
            assert res["n_bits"] == d["n_bits"] and res["crc_ok"] == d["crc_ok"] and res["index"] == d["index"]
            assert np.array_equal(rfid.unpack_bits(res["bits"], int(d["n_bits"])), d["bits"][: d["n_bits"]])
    finally:
        tb.ctx.close()

import rfid.flowgraph
for foreign in (False, True):
    ok = bad = 0
    for seed in range(60 if not foreign else 45):
        sigma = (0.01, 0.05, 0.08)[seed % 3]
        lo, hi = ((1000, 150000), (20000, 400000), (200, 9000))[(seed // 3) % 3]
        try:
            run(seed, sigma, lo, hi, 30 if hi < 10000 else 60, foreign_filter=foreign)
            ok += 1
        except Exception as e:
            bad += 1
            print("seed", seed, "sigma", sigma, "calls", (lo, hi), "foreign filter" if foreign else "", "FAILED:", repr(e)[:300])
    print("look-ahead keyed on the", "gate (filter on the host)" if foreign else "matched filter", ": passed", ok, "failed", bad)
