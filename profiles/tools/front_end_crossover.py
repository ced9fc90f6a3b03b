# measures both front ends of rfid_batch_process at several batch sizes (the crossover the calibrated cost model of mode 1 has to find)
import sys, json, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/gen2-uhf-rfid-reader_amd')
import numpy as np, torch, rfid
from rfid import synth
t = synth.make_trace(n_rounds=71, fixed_q=0, tag_ids=(0x27,), sigma=0.0, seed=7, noise=False, render=False)
ctx = rfid.Context(device=0, max_num_queries=(1<<31)-2)
L = ctx.synth_gen2_size(t.plan); stride=(L+1)&~1
base = torch.zeros(2*stride, dtype=torch.float32, device='cuda:0')
torch.cuda.synchronize()
ctx.synth_gen2_ptr(t.plan, base.data_ptr(), stride, sigma=0.0)
out=[]
for B in (1,8,32,64,128,256,512):
    data = torch.empty((B,2*stride), dtype=torch.float32, device='cuda:0')
    torch.cuda.synchronize()
    ctx.synth_replicas_ptr(base.data_ptr(), L, data.data_ptr(), stride, B, 0.003, 5, first_replica=0)
    ctx.batch_sync()
    ctx.batch_plan(B, L)
    res={}
    for mode in (0,2,1):
        ctx.batch_set_long_stream(mode)
        for rep in range(3):
            torch.cuda.synchronize(); t0=time.perf_counter()
            ctx.batch_process_ptr(data.data_ptr(), stride, L, 0, want_scores=False); ctx.batch_sync()
            dt=(time.perf_counter()-t0)*1e3
        res[mode]=round(dt,3)
        if mode==1: res['auto_used_ls']=int(ctx.batch_ls_report()['pieces']>0)
    out.append((B,res)); print(B,res, flush=True)
    del data
