#!/bin/bash
# usage: lsprof.sh <tag> [lib.so]  -> per-kernel averages of configs[2] (rocprofv3 --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; O=$R/gpurun_out/lsprof_$tag; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp RFID_LS_CALIBRATE=0
L=$R/gen2-uhf-rfid-reader_amd/lib/librfid_mi355x.so; cp $L /tmp/keep_ls.so; [ -n "$2" ] && cp $R/$2 $L
( cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $R/bench.py --config ${CFG:-2} --steps 5 --warmup 2 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $O/log.txt 2>&1 )
python - $O/t/t_kernel_stats.csv $tag <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rfidk" in r["Name"] and "synth" not in r["Name"]]
tot = 0.0
passes = float([r for r in rows if "mf_boxcar" in r["Name"]][0]["Calls"])
print("==", sys.argv[2])
for r in rows:
    n = r["Name"].split("(")[0].replace("rfidk::", "")
    per_pass = float(r["TotalDurationNs"]) / (passes * 1e6)
    tot += per_pass
    print("  %-28s calls/pass %5.1f  ms/pass %7.3f" % (n, int(r["Calls"]) / passes, per_pass))
print("  total kernel ms per pass %.3f" % tot)
PY
tail -c 300 $O/log.txt | grep -o '"ms_per_step": [0-9.]*' | head -1
rm -rf $O/t
cp /tmp/keep_ls.so $L
