#!/bin/bash
# configs[2] (and optionally configs[3]) bench lines for library build variants under gen2-uhf-rfid-reader_amd/lib_ab/<name>/
# usage: r05_c2_variants.sh <outdir under gpurun_out> <variant|main>[:ENV=VAL[,ENV=VAL...]] ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp RFID_LS_CALIBRATE=0
L=$R/gen2-uhf-rfid-reader_amd/lib/librfid_mi355x.so
cp $L /tmp/keep_main.so
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  if [ "$v" = main ]; then cp /tmp/keep_main.so $L; else cp $R/gen2-uhf-rfid-reader_amd/lib_ab/$v/librfid_mi355x.so $L; fi
  tag=$(echo $spec | tr ':=,' '___')
  for cfg in ${CFGS:-2}; do
    steps=6; [ $cfg = 3stream ] && steps=20
    ( timeout 300 env $envs python bench.py --config $cfg --steps $steps --warmup 1 --no-cpu-baseline --no-stream-leg --no-other-configs > $O/bench_${cfg}_$tag.json 2> $O/bench_${cfg}_$tag.err )
    python - $O/bench_${cfg}_$tag.json "$cfg $spec" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("%-40s FAILED (%r)" % (sys.argv[2], e)); sys.exit(0)
rk = {k: v["ms_per_step"] for k, v in d["roofline_by_kernel"].items()}
ls = d.get("long_stream", {})
print("%-40s b2b %8.4f  each %8.4f  kernels %s  avg_rounds %s verified %s | %s" % (
    sys.argv[2], d["ms_per_step"], d["passes_each_waited_for"]["ms_per_step"], rk, ls.get("avg_rounds"), ls.get("verified"), d["parity_check"][:40]))
PY
  done
done 2>&1 | tee $O/summary.txt
cp /tmp/keep_main.so $L
