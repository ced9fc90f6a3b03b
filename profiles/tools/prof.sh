#!/bin/bash
# rocprofv3 collection for profiles/: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r02 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stream-leg > $OUT/trace_bench.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg > $OUT/fetch_bench.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg > $OUT/write_bench.log 2>&1
find $OUT -type f | head -30; tail -2 $OUT/trace_bench.log | cut -c1-300
