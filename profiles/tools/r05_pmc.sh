#!/bin/bash
# HBM traffic per kernel of bench.py --config ${CFG} (each pass waited for): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate
# passes (MI355X_MICROARCH.md: FETCH_SIZE x2 on gfx950), -> <outdir>/pmc_<cfg>.csv
# usage: r05_pmc.sh <outdir under gpurun_out> <config: 1|2|3stream> [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; cfg=$2; shift; shift; mkdir -p $O
export TMPDIR=/tmp RFID_LS_CALIBRATE=0
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp; timeout 400 env "$@" rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${cfg}_$ctr -o f -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $O/pmc_${cfg}_$ctr.log 2>&1 )
done
python - $O $cfg > $O/pmc_$cfg.csv <<'PY'
import csv, collections, sys
O, cfg = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        rows = list(csv.DictReader(open(f"{O}/pmc_{cfg}_{ctr}/f_counter_collection.csv")))
    except OSError:
        continue
    for r in rows:
        if "rfidk" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            agg[r["Kernel_Name"].split("(")[0].replace("rfidk::", "")][ctr].append(float(r["Counter_Value"]))
print("kernel,launches_seen,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,hbm_bytes_avg_per_launch(2*F+W)*1024")
for k in sorted(agg):
    f, w = agg[k]["FETCH_SIZE"], agg[k]["WRITE_SIZE"]
    fa = sum(f) / len(f) if f else 0.0
    wa = sum(w) / len(w) if w else 0.0
    print("%s,%d,%.1f,%.1f,%.1f,%.1f,%.0f" % (k, max(len(f), len(w)), fa, wa, sum(f), sum(w), (2 * fa + wa) * 1024))
PY
rm -rf $O/pmc_${cfg}_FETCH_SIZE $O/pmc_${cfg}_WRITE_SIZE
cat $O/pmc_$cfg.csv
