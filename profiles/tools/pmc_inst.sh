#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/pmc_inst; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "FETCH_SIZE WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg $BENCH_ARGS > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, collections, glob
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/*/f_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rfidk" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("rfidk::",""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg): print("%-28s %-22s %14.0f" % (k[0], k[1], sum(agg[k])/len(agg[k])))
PY
