#!/bin/bash
# Round 4: everything profiles/r03 holds, from one build on one box.  Outputs under gpurun_out/final_r04/:
#   bench_<config>.json                       the bench.py line of every BASELINE configuration's per-GPU workload
#   kernel_stats_<config>.csv                 rocprofv3 --kernel-trace --stats of the same command (no CPU leg)
#   pmc_<config>.csv                          FETCH_SIZE / WRITE_SIZE per kernel, one counter per run, summed per pass
#   sq_counters.txt                           SQ instruction / cycle counters of configs[1] (pmc_inst.sh)
#   drop_in_path.txt, crossover.txt           the block-by-block adaptor rates; long-stream vs fused front end by batch size
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/final_r04; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for cfg in 1 2 3stream 4shard; do
  timeout 600 python bench.py --config $cfg > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
(python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/pytest_gpu.log
timeout 300 python bench.py --streams 4096 --no-cpu-baseline --no-stream-leg > $O/bench_1_4096_traces.json 2> $O/bench_1_4096.err
# the profiled runs: without the front-end calibration rfid_ctx_create does (two small batches through both front ends --
# their launches would sit in the per-kernel averages); the choice of front end is the same for these workloads
export RFID_LS_CALIBRATE=0
for cfg in 1 2 3stream; do
  ( cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -o t -- python $R/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $O/trace_$cfg.log 2>&1 )
  cp $O/trace_$cfg/t_kernel_stats.csv $O/kernel_stats_$cfg.csv 2>/dev/null
done
for cfg in 1 2; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp; timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${cfg}_$ctr -o f -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > $O/pmc_${cfg}_$ctr.log 2>&1 )
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
done
# (SQ counters of configs[1]: profiles/r04/front_end_ab.txt, collected with the A/B of the round)
unset RFID_LS_CALIBRATE
{ echo "== scratch-free probe of the fixed costs (profiles/tools/ctx_time.py)"; python profiles/tools/ctx_time.py; echo "== profiles/tools/dropin_rates.py"; timeout 600 python profiles/tools/dropin_rates.py; } > $O/drop_in_path.txt 2>&1
rm -rf $O/trace_* $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
ls -la $O; tail -c 700 $O/bench_1.json
