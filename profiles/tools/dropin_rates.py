# developer aid: rates of the drop-in path (rfid_reader_offline) block by block and whole-chain, beside the oracle on one core
import os, subprocess, sys, time
sys.path.insert(0, "gen2-uhf-rfid-reader_amd"); sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.is_available()
import rfid
from rfid import synth
from oracle import oracle
exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
for rounds in (71, 2000):
    t = synth.make_trace(n_rounds=rounds, seed=7, sigma=0.002, corrupt_rounds=(36,)).samples
    path = "/tmp/trace_%d.bin" % rounds
    rfid.batch.write_trace_file(path, t)
    tt = oracle.time_trace(t, reps=3, cfg=oracle.config(max_num_queries=1 << 30))
    print("%d rounds, %d raw samples: oracle on one core %.1f Msamples/s" % (rounds, len(t), 3 * len(t) / tt["total_s"] / 1e6))
    for extra, la in (([], "1"), ([], "0"), (["--chunk", "65536"], "1"), (["--chunk", "65536"], "0"), (["--chunk", "262144"], "1"), (["--whole-chain", "4000000"], "1"), (["--whole-chain", "32000000"], "1"),
                      (["--host-fir"], "1"), (["--host-fir", "--chunk", "65536"], "1"), (["--host-fir", "--chunk", "65536"], "0")):
        out = subprocess.run([exe, path, "--time", "--max-queries", "100000000"] + extra, capture_output=True, text=True, timeout=600, env=dict(os.environ, RFID_LOOKAHEAD=la))
        print("   ", " ".join(extra) or "(default --chunk 8192)", "look-ahead" if la == "1" else "no look-ahead", "->", out.stderr.strip().split("rfid_reader_offline: ")[-1], "|", out.stdout.split("\n")[5] if out.returncode == 0 else out.stderr)
