#!/bin/bash
# round 6: the long-stream front end (configs[2], configs[3]'s per-GPU stream) at SURVEY 8(d)'s stress noise levels.
# usage (GPU box, repo root): bash profiles/tools/r06_noise_sweep.sh [out-dir]
out=${1:-gpurun_out/noise}
mkdir -p $out
for cfg in 3stream 2; do
  for sg in 0.002 0.03 0.06; do
    timeout 900 python bench.py --config $cfg --sigma $sg --no-cpu-baseline > $out/bench_${cfg}_s${sg}.json 2> $out/bench_${cfg}_s${sg}.err
    echo "rc=$? cfg=$cfg sigma=$sg" >> $out/summary.txt
    python - $out/bench_${cfg}_s${sg}.json >> $out/summary.txt <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ls = d.get("long_stream", {})
    print("  ms_per_step(b2b) %.4f  each-waited %.4f  parity: %s" % (d["ms_per_step"], d["passes_each_waited_for"]["ms_per_step"], d["parity_check"][:60]))
    print("  long_stream:", {k: v for k, v in ls.items() if k != "note"})
    print("  kernels:", {k: v["ms_per_step"] for k, v in d["roofline_by_kernel"].items()})
except Exception as e:
    print("  (no line: %r)" % (e,))
PY
  done
done
# the same at other phases of the carrier leak (L = e^{j 0.7}: 25 sin(0.7) = 16.1 sits next to a power of two)
for cfg in 3stream 2; do
  for ph in 0.6 0.3; do
    sg=0.06
    timeout 900 python bench.py --config $cfg --sigma $sg --leak-phase $ph --no-cpu-baseline > $out/bench_${cfg}_s${sg}_ph${ph}.json 2> $out/bench_${cfg}_s${sg}_ph${ph}.err
    echo "rc=$? cfg=$cfg sigma=$sg leak-phase=$ph" >> $out/summary.txt
    python - $out/bench_${cfg}_s${sg}_ph${ph}.json >> $out/summary.txt <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ls = d.get("long_stream", {})
    print("  ms_per_step(b2b) %.4f  each-waited %.4f  parity: %s" % (d["ms_per_step"], d["passes_each_waited_for"]["ms_per_step"], d["parity_check"][:60]))
    print("  long_stream:", {k: v for k, v in ls.items() if k != "note"})
except Exception as e:
    print("  (no line: %r)" % (e,))
PY
  done
done
cat $out/summary.txt
