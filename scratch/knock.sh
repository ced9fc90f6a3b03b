#!/bin/bash
# developer aid: time the fused front end with parts knocked out (results are wrong by design)
for k in 0 32 64 96 111 79; do
  RFID_GATE_KNOCK=$k python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
v=d['roofline_by_kernel']['front_end_fused']
print('knock $k: front', v['ms_per_step'], 'min', v['min_ms_per_step'])"
done
