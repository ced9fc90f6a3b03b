"""Generates the committed golden fixtures in tests/golden/*.npz.

The reference has no golden vectors for this path and cannot be run here (it needs GNU
Radio), so these vectors are produced by the CPU oracle (oracle/rfid_oracle.c) on synthetic
traces: inputs (raw complex64 I/Q) plus every value the path computes for them.  They freeze
the oracle's behaviour (a guard against drift) and give the GPU tests inputs that do not
depend on numpy's RNG.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gen2-uhf-rfid-reader_amd"))
from oracle import oracle  # noqa: E402
from rfid import synth  # noqa: E402

CASES = {
    # name: (make_trace kwargs, oracle config kwargs)
    "q0_sigma0p03": (dict(n_rounds=1, sigma=0.03, seed=101, t1_jitter_raw=7), {}),
    "q0_sigma0p002_cut": (dict(n_rounds=2, sigma=0.002, seed=102), {}),          # cut mid-EPC below
    "q1_two_tags": (dict(n_rounds=1, fixed_q=1, tag_ids=(0x3C, 0xA5), sigma=0.01, seed=103), dict(fixed_q=1)),
}


def build(name):
    tk, ck = CASES[name]
    t = synth.make_trace(**tk)
    x = t.samples
    if name.endswith("_cut"):
        x = x[: int(len(x) * 0.83)]          # last EPC window incomplete -> never decoded
    o = oracle.run_trace(x, oracle.config(**ck))
    s = o.state
    return dict(
        raw=x, fixed_q=np.int32(ck.get("fixed_q", 0)),
        mf=oracle.fir(x),
        open_idx=o.open_idx.astype(np.int64), dc=o.dc,
        type=o.dumps["type"], index=o.dumps["index"], corr=o.dumps["corr"], h_est=o.dumps["h_est"],
        energy=o.dumps["energy"], T=o.dumps["T"], n_bits=o.dumps["n_bits"], bits=o.dumps["bits"],
        crc_ok=o.dumps["crc_ok"], tag_id=o.dumps["tag_id"],
        stats=np.array([s.n_queries_sent, s.cur_inventory_round, s.cur_slot_number, s.n_epc_correct,
                        s.n_unique_tags, s.status], dtype=np.int32),
        tag_reads=np.array(s.tag_reads[:], dtype=np.int32),
        print_results=np.frombuffer(o.print_results().encode(), dtype=np.uint8),
    )


if __name__ == "__main__":
    for name in CASES:
        d = build(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, len(d["raw"]), "samples,", len(d["type"]), "windows ->", os.path.getsize(path), "bytes")
