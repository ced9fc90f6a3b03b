#!/bin/bash
# round 5, review item 3: the wave-per-piece mapping (the long-stream front end, RFID_LONG_STREAM=2) against the four-role fused
# front end (the default for many traces) on configs[1]'s data: bench lines, rocprofv3 --stats of the wave-per-piece run, and the
# SQ instruction counters of both.   usage: r05_front_map.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
export TMPDIR=/tmp RFID_LS_CALIBRATE=0
BA="--config 1 --no-cpu-baseline --no-stream-leg --no-other-configs"
for v in fused:1 pieces:2; do
  tag=${v%%:*}; ls=${v#*:}
  timeout 300 env RFID_LONG_STREAM=$ls python bench.py $BA --steps 10 --warmup 2 > $O/bench_1_$tag.json 2> $O/bench_1_$tag.err
  python - $O/bench_1_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-8s b2b %.4f ms  each %.4f ms  kernels %s  %s" % (sys.argv[2], d["ms_per_step"], d["passes_each_waited_for"]["ms_per_step"],
          {k: round(v["ms_per_step"], 4) for k, v in d["roofline_by_kernel"].items()}, d["parity_check"][:60]))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e))
PY
done 2>&1 | tee $O/summary.txt
( cd /tmp; timeout 300 env RFID_LONG_STREAM=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pieces -o s -- python $R/bench.py $BA --steps 5 --warmup 1 --no-back-to-back > $O/stats_pieces.log 2>&1 )
cp $O/stats_pieces/s_kernel_stats.csv $O/kernel_stats_1_pieces.csv 2>/dev/null; rm -rf $O/stats_pieces
for v in fused:1 pieces:2; do
  tag=${v%%:*}; ls=${v#*:}
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    st=$(echo $set | tr ' ' '_' | cut -c1-30)
    ( cd /tmp; timeout 300 env RFID_LONG_STREAM=$ls rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/sq_${tag}_$st -o f -- python $R/bench.py $BA --steps 1 --warmup 1 --no-back-to-back > $O/sq_${tag}_$st.log 2>&1 )
  done
  python - $O $tag > $O/sq_$tag.txt <<'PY'
import csv, collections, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/sq_" + sys.argv[2] + "_*/f_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rfidk::" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("rfidk::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print("%-26s %-20s launches=%3d per_pass=%16.0f" % (k[0], k[1], len(v), sum(v) / 2.0))   # (warm-up pass + one step)
PY
  rm -rf $O/sq_${tag}_*/
  cat $O/sq_$tag.txt
done
