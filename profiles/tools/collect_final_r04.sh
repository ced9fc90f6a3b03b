#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/final_r04c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp RFID_LS_CALIBRATE=0
( cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_1 -o t -- python $R/bench.py --config 1 --steps 5 --warmup 2 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $O/trace_1.log 2>&1 )
cp $O/trace_1/t_kernel_stats.csv $O/kernel_stats_1.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp; timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_1_$ctr -o f -- python $R/bench.py --config 1 --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $O/pmc_1_$ctr.log 2>&1 )
done
python - $O 1 > $O/pmc_1.csv <<'PY'
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
bash profiles/tools/sq_counters_config2.sh final_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" > $O/sq_config2_a.txt 2>&1
bash profiles/tools/sq_counters_config2.sh final_b "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" > $O/sq_config2_b.txt 2>&1
rm -rf $O/trace_1 $O/pmc_1_FETCH_SIZE $O/pmc_1_WRITE_SIZE
ls $O
