#!/bin/bash
# collects what profiles/rNN holds for a build: the bench line of every configuration, rocprofv3 stats + PMC traffic
# (prof.sh) and the SQ counters (pmc_inst.sh); outputs under gpurun_out/final_r02/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/final_r02; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_1.json 2> $O/bench_1.err
python bench.py --config 2 > $O/bench_2.json 2> $O/bench_2.err
python bench.py --config 3stream > $O/bench_3stream.json 2> $O/bench_3stream.err
python bench.py --config 4shard > $O/bench_4shard.json 2> $O/bench_4shard.err
python bench.py --streams 4096 --no-cpu-baseline --no-stream-leg > $O/bench_1_4096_traces.json 2> $O/bench_1_4096.err
bash profiles/tools/prof.sh > $O/prof.log 2>&1
bash profiles/tools/pmc_inst.sh > $O/sq_counters.txt 2> $O/pmc_inst.err
tail -c 600 $O/bench_1.json
