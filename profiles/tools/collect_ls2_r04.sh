#!/bin/bash
# refresh of the configs[2] / configs[3] records of profiles/r04 after the long-stream work of the round's second half
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/final_r04b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for cfg in 2 3stream; do
  timeout 600 python bench.py --config $cfg > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
export RFID_LS_CALIBRATE=0
for cfg in 2 3stream; do
  ( cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -o t -- python $R/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $O/trace_$cfg.log 2>&1 )
  cp $O/trace_$cfg/t_kernel_stats.csv $O/kernel_stats_$cfg.csv 2>/dev/null
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp; timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_2_$ctr -o f -- python $R/bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $O/pmc_2_$ctr.log 2>&1 )
done
python - $O 2 > $O/pmc_2.csv <<'PY'
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
rm -rf $O/trace_* $O/pmc_2_FETCH_SIZE $O/pmc_2_WRITE_SIZE
ls -la $O
unset RFID_LS_CALIBRATE
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_1.json 2> $O/bench_1.err
