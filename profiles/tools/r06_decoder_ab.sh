#!/bin/bash
# round 6: tag_decoder A/B -- the EPC half-period search with its 256 gather positions computed (6 VALU + 1 LDS per gather: "base") or taken
# from a table of LDS byte offsets ("table": 2 VALU + 1 LDS per gather, 512 B of table per lane and pack out of L2).  Two builds of the
# library (scratch/lib_dec_{base,table}.so), both at 2 waves per SIMD (232 / 233 VGPRs), alternating runs on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-dec}; mkdir -p $O; export TMPDIR=/tmp
L=$R/gen2-uhf-rfid-reader_amd/lib/librfid_mi355x.so; cp $L /tmp/keep.so
for streams in 1024 4096; do for rep in 1 2; do for v in base table; do
  cp $R/scratch/lib_dec_$v.so $L
  python $R/bench.py --streams $streams --no-cpu-baseline --no-stream-leg --no-other-configs > $O/b.json 2>/dev/null
  python - $O/b.json $v $rep $streams <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["roofline_by_kernel"]["tag_decoder"]
print("%-5s streams=%s rep=%s  ms_per_step %.4f  decoder ms %.4f (min %.4f)  frac %.4f  parity: %s" % (sys.argv[2], sys.argv[4], sys.argv[3], d["ms_per_step"], k["ms_per_step"], k["min_ms_per_step"], k["frac"], d["parity_check"][:40]))
PY
done; done; done
for v in base table; do
  cp $R/scratch/lib_dec_$v.so $L
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$v -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > /dev/null 2>&1 )
  echo "$v rocprofv3 (Name,Calls,TotalNs,AvgNs,%,MinNs):"; grep "decode_all" $O/t_$v/t_kernel_stats.csv | cut -d, -f1-6 | cut -c1-120
  ( cd /tmp; timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/p_$v -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg --no-other-configs --no-back-to-back > /dev/null 2>&1 )
  python - $O/p_$v/p_counter_collection.csv $v <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "decode_all" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("%s SQ counters per launch:" % sys.argv[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in agg.items()})
except Exception as e:
    print("no counters:", e)
PY
  rm -rf $O/t_$v $O/p_$v
done
cp /tmp/keep.so $L
