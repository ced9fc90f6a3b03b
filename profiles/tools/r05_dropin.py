# round 5: rates of the drop-in path (rfid_reader_offline) at small scheduler buffers, beside the oracle on one core; and the
# 1.08 M-sample file through a WARM process (a second file through a context that exists)
import os, subprocess, sys, time
sys.path.insert(0, "gen2-uhf-rfid-reader_amd"); sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.is_available()
import rfid
from rfid import synth
from oracle import oracle
exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
t = synth.make_trace(n_rounds=2000, seed=7, sigma=0.002, corrupt_rounds=(36,)).samples
path = "/tmp/trace_2000.bin"
rfid.batch.write_trace_file(path, t)
tt = oracle.time_trace(t, reps=3, cfg=oracle.config(max_num_queries=1 << 30))
print("2000 rounds, %d raw samples: oracle on one core %.1f Msamples/s" % (len(t), 3 * len(t) / tt["total_s"] / 1e6))
E0 = {}
ROWS = ((["--chunk", "8192"], "1", E0), (["--chunk", "8192"], "1", E0), (["--chunk", "8192"], "0", E0), (["--chunk", "65536"], "1", E0), (["--chunk", "262144"], "1", E0),
        (["--host-fir", "--chunk", "8192"], "1", E0), (["--host-fir", "--chunk", "65536"], "1", E0),
        # every filter call returns its own outputs (the C-ABI's default), the trace in page-locked / in ordinary memory
        (["--chunk", "8192"], "1", {"RFID_MF_LATE_OUTPUTS": "0"}), (["--chunk", "8192", "--pageable"], "1", {"RFID_MF_LATE_OUTPUTS": "0"}),
        # GNU Radio's scheduling rules, bounded buffers: the gate consumes ahead (the adaptors' default) / decides at once
        (["--scheduler", "bounded", "--buffer", "8192", "--host-fir"], "1", E0), (["--scheduler", "bounded", "--buffer", "8192"], "1", E0),
        (["--scheduler", "bounded", "--buffer", "65536"], "1", E0), (["--scheduler", "bounded", "--buffer", "65536", "--host-fir"], "1", E0),
        (["--scheduler", "bounded", "--buffer", "8192", "--host-fir"], "1", {"RFID_GATE_CONSUME_AHEAD": "0"}),
        (["--scheduler", "bounded", "--buffer", "8192"], "1", {"RFID_GATE_CONSUME_AHEAD": "0"}),
        (["--whole-chain", "4000000"], "1", E0))
for extra, la, env in ROWS:
    out = subprocess.run([exe, path, "--time", "--max-queries", "100000000"] + extra, capture_output=True, text=True, timeout=600, env=dict(os.environ, RFID_LOOKAHEAD=la, **env))
    print("   ", " ".join(extra), " ".join("%s=%s" % kv for kv in env.items()), "look-ahead" if la == "1" else "no look-ahead", "->", out.stderr.strip().split("rfid_reader_offline: ")[-1], "|", out.stdout.split("\n")[5] if out.returncode == 0 else out.stderr[-300:])
print("\n== where the time goes (RFID_LA_PROFILE=1)")
for extra in (["--chunk", "8192"], ["--chunk", "8192", "--host-fir"], ["--scheduler", "bounded", "--buffer", "8192", "--host-fir"], ["--scheduler", "bounded", "--buffer", "8192"]):
    out = subprocess.run([exe, path, "--time", "--max-queries", "100000000"] + extra, capture_output=True, text=True, timeout=600, env=dict(os.environ, RFID_LA_PROFILE="1"))
    print("== " + " ".join(extra))
    for line in out.stderr.splitlines():
        if line.startswith("rfid_reader_offline:") or line.startswith("[la]"):
            print(line)
# warm process: the 71-round file (1 076 066 raw samples, the size of the reference's own test trace) through rfid.batch / a stream of a context that exists
small = synth.make_trace(n_rounds=71, seed=7, sigma=0.002, corrupt_rounds=(36,)).samples
ctx = rfid.Context(device=0, max_num_queries=1 << 30)
for rep in range(4):
    t0 = time.perf_counter()
    ctx.stream_begin(len(small))
    w, r = ctx.stream_work(small)
    w2, r2 = ctx.stream_work(flush=True)
    dt = time.perf_counter() - t0
    ctx.stream_end()
    print("warm process, pass %d: rfid_stream_begin + one rfid_stream_work call + flush over the 71-round file: %.2f ms, %d windows, %d EPC ok" % (rep, 1e3 * dt, len(w) + len(w2), int(r["crc_ok"].sum() + r2["crc_ok"].sum())))
ctx.close()
