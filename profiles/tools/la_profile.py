# where the time of the block-by-block path with look-ahead goes (RFID_LA_PROFILE=1), 30 M-sample trace
import os, subprocess, sys
sys.path.insert(0, "gen2-uhf-rfid-reader_amd"); sys.path.insert(0, ".")
import torch
torch.cuda.is_available()
import rfid
from rfid import synth
exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
t = synth.make_trace(n_rounds=2000, seed=7, sigma=0.002, corrupt_rounds=(36,)).samples
path = "/tmp/trace_2000.bin"
rfid.batch.write_trace_file(path, t)
for chunk in ("65536", "131072"):
    for rep in range(2):
        out = subprocess.run([exe, path, "--time", "--max-queries", "100000000", "--chunk", chunk], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, RFID_LA_PROFILE="1"))
    print(chunk, out.stderr.strip())
