#!/bin/bash
# what the rounds of the long-stream front end did on configs[2] (RFID_LS_DEBUG: per-round counts, margin / |D| histograms), fused and not
# usage: ls2_rounds.sh <outdir under gpurun_out> [bench config, default 2]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${1:-r05b}; mkdir -p $O
export TMPDIR=/tmp RFID_LS_CALIBRATE=0
for f in 1 0; do
  ( timeout 300 env RFID_LS_FUSED=$f RFID_LS_DEBUG=1 python bench.py --config ${2:-2} --steps 1 --warmup 0 --no-cpu-baseline --no-stream-leg --no-other-configs 2>&1 >/dev/null | grep "^\[ls2\]" | head -16 ) > $O/rounds_fused$f.txt
  echo "== RFID_LS_FUSED=$f"; cat $O/rounds_fused$f.txt
done
