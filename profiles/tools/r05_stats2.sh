#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py --config ${CFG:-2} (each pass waited for) -> per-kernel ms per pass
# usage: r05_stats2.sh <outdir under gpurun_out> <tag> [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; tag=$2; shift; shift; mkdir -p $O
export TMPDIR=/tmp RFID_LS_CALIBRATE=0
( cd /tmp; timeout 400 env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$tag -o t -- python $R/bench.py --config ${CFG:-2} --steps 5 --warmup 2 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $O/stats_$tag.log 2>&1 )
cp $O/t_$tag/t_kernel_stats.csv $O/kernel_stats_$tag.csv 2>/dev/null
python - $O/kernel_stats_$tag.csv $tag <<'PY' | tee $O/stats_$tag.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rfidk" in r["Name"] and "synth" not in r["Name"]]
passes = float([r for r in rows if "decode_all" in r["Name"] or "decode_epc3" in r["Name"]][0]["Calls"])
tot = 0.0
print("==", sys.argv[2], "(%d passes)" % passes)
for r in rows:
    n = r["Name"].split("(")[0].replace("rfidk::", "")
    per_pass = float(r["TotalDurationNs"]) / (passes * 1e6)
    tot += per_pass
    print("  %-28s calls/pass %5.1f  ms/pass %7.3f" % (n, int(r["Calls"]) / passes, per_pass))
print("  total kernel ms per pass %.3f" % tot)
PY
rm -rf $O/t_$tag
