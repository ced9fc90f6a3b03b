#!/bin/bash
# developer check: GPU clock / power while the fused front end runs back to back (1024 traces, then 512)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for streams in 1024 512; do
  (RFID_LONG_STREAM=0 python bench.py --streams $streams --steps 8000 --warmup 5 --no-cpu-baseline --no-stream-leg > /tmp/b_$streams.json 2>/dev/null) &
  BP=$!
  for i in $(seq 1 45); do
    s=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' \t' ' ' | sed 's/GPU\[0\] ://' | tr '\n' ';')
    echo "streams=$streams t=$i $s"
    kill -0 $BP 2>/dev/null || break
    sleep 1
  done
  wait $BP
  tail -1 /tmp/b_$streams.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', $streams, 'ms_per_step', d['ms_per_step'], 'front', d['front_end_ms'])"
done
