#!/bin/bash
# round 5: the long-stream front end with its fused first pass (ls2_front_kernel) against the unfused list, and build variants
# of the fused kernel (read-ahead depth / occupancy cap), on configs[2] (and configs[3]'s per-GPU stream).
# usage: r05_front_ab.sh <outdir under gpurun_out> [variant dirs under gen2-uhf-rfid-reader_amd/lib_ab ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/${1:-r05a}; shift; mkdir -p $O
export TMPDIR=/tmp RFID_LS_CALIBRATE=0
L=$R/gen2-uhf-rfid-reader_amd/lib/librfid_mi355x.so
cp $L /tmp/keep_main.so
{
echo "== configs[2] (bench.py --config 2 --steps 6 --warmup 1)"
BENCH_ARGS="--config 2 --steps 6 --warmup 1"
run() { tag=$1; shift; ( timeout 300 env "$@" python bench.py $BENCH_ARGS --no-cpu-baseline --no-stream-leg --no-other-configs > $O/bench_$tag.json 2> $O/bench_$tag.err ); line_print $tag; }
line_print() {
  python - $O/bench_$1.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("%-22s FAILED (%r)" % (sys.argv[2], e)); sys.exit(0)
rk = {k: v["ms_per_step"] for k, v in d["roofline_by_kernel"].items()}
ls = d.get("long_stream", {})
print("%-22s b2b %8.4f  each %8.4f  kernels %s  pieces %s avg_rounds %s reruns %s dc_rounds %s verified %s | %s" % (
    sys.argv[2], d["ms_per_step"], d["passes_each_waited_for"]["ms_per_step"], rk, ls.get("pieces"), ls.get("avg_rounds"),
    ls.get("avg_reruns"), ls.get("dc_rounds"), ls.get("verified"), d["parity_check"][:60]))
PY
}
run c2_unfused RFID_LS_FUSED=0
run c2_fused_main RFID_LS_FUSED=1
for v in "$@"; do
  cp $R/gen2-uhf-rfid-reader_amd/lib_ab/$v/librfid_mi355x.so $L
  run c2_fused_$v RFID_LS_FUSED=1
done
cp /tmp/keep_main.so $L
echo "== configs[3] per GPU (bench.py --config 3stream --steps 20 --warmup 3)"
BENCH_ARGS="--config 3stream --steps 20 --warmup 3"
run c3_unfused RFID_LS_FUSED=0
run c3_fused_main RFID_LS_FUSED=1
} 2>&1 | tee $O/summary.txt
cp /tmp/keep_main.so $L
