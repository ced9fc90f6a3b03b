import sys, os
sys.path.insert(0, "gen2-uhf-rfid-reader_amd"); sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.is_available()
import rfid
from rfid import synth
from oracle import oracle
t = synth.make_trace(n_rounds=80, fixed_q=1, tag_ids=(0x27, 0x3C), seed=812, sigma=0.01, t1_jitter_raw=5).samples
o = oracle.run_trace(t, oracle.config(fixed_q=1))
ctx = rfid.Context(device=0, fixed_q=1)
ctx.stream_begin(401000)
pos = 0; ws=[]
while pos < len(t):
    n = min(400000, len(t) - pos)
    w, r = ctx.stream_work(t[pos:pos + n]); ws.append(w)
    pos += n
w, r = ctx.stream_work(flush=True); ws.append(w)
w = np.concatenate(ws)
print(len(w), o.n_windows)
bad = np.nonzero(w["start"] != o.open_idx)[0]
print("bad", bad[:10], w["start"][bad[:5]], o.open_idx[bad[:5]], [len(x) for x in ws])
print((w["start"] - o.open_idx)[:60])
