#!/bin/bash
# usage: sqset.sh <tag> "<counters>"   (configs[2])
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/sqset_$1; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp RFID_LS_CALIBRATE=0
timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/c -o f -- python $R/bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $OUT/log.txt 2>&1
python - $OUT <<'PY'
import csv, collections, glob, sys
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/c/f_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rfidk::" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("rfidk::",""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if "seq" in k[0] or "clear" in k[0] or "assemble" in k[0] or "pieces" in k[0] or "dc_cut" in k[0]: continue
    v = agg[k]; print("%-26s %-20s n=%3d first=%14.0f sum=%14.0f" % (k[0], k[1], len(v), v[0], sum(v)))
PY
rm -rf $OUT/c
