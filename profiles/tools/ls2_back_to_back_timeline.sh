#!/bin/bash
# per-launch timeline of configs[2] passes enqueued back to back (both streams): where the next pass's filter parts run
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/lstrace_b2b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp RFID_LS_CALIBRATE=0 GPU_MAX_HW_QUEUES=8
( cd /tmp; timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python $R/bench.py --config 2 --steps 4 --warmup 1 --no-cpu-baseline --no-stream-leg --no-other-configs > $O/log.txt 2>&1 )
python - $O/t/t_kernel_trace.csv > $O/timeline.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rfidk" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("rfidk::", "")
# the timed region: the four passes enqueued back to back = the densest run of passes; take the window between the third-last
# and the last ls2_clear launch that belongs to it
clears = [i for i, r in enumerate(rows) if name(r) == "ls2_clear_kernel"]
# passes: wake (8) + warm-up (1) + timed (4) + kernel series (3) + each-waited (4); the timed ones are 10th..13th
a, b = clears[10], clears[12]
t0 = int(rows[a]["Start_Timestamp"])
print("configs[2], passes enqueued back to back: launches between the start of the 2nd and of the 4th timed pass (us from the 2nd pass's first launch)")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 - 4_500_000 or s > int(rows[b]["Start_Timestamp"]): continue
    if (e - s) < 20_000 and name(r) not in ("ls2_clear_kernel",): continue      # (launches under 20 us left out)
    print("%-28s queue %-3s start %9.1f  dur %8.1f" % (name(r), r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3))
PY
rm -rf $O/t
tail -c 400 $O/log.txt | grep -o '"ms_per_step": [0-9.]*' | head -1
