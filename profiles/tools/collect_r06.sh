#!/bin/bash
# round 6: the records behind DESIGN.md / README.md -- the GPU test log, bench lines of every configuration's per-GPU workload, rocprofv3
# kernel stats and PMC traffic of configs[1], configs[2] and configs[3]'s stream, the noise sweep (sigma 0.002 / 0.03 / 0.06, two other
# leak phases), drop-in rates, fuzzers with their verdicts by sigma
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/${1:-final_r06}; mkdir -p $O
export TMPDIR=/tmp RFID_LS_CALIBRATE=0
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for cfg in 1 2 3stream 4shard; do
  python bench.py --config $cfg --no-cpu-baseline --no-stream-leg --no-other-configs > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
python bench.py --config 1 --streams 4096 --no-cpu-baseline --no-stream-leg --no-other-configs > $O/bench_1_4096_traces.json 2>> $O/bench_1.err
for cfg in 1 2 3stream; do
  CFG=$cfg bash profiles/tools/r05_stats2.sh ${1:-final_r06} $cfg > /dev/null 2>&1
  bash profiles/tools/r05_pmc.sh ${1:-final_r06} $cfg > /dev/null 2>&1
done
bash profiles/tools/r06_noise_sweep.sh $O/noise > /dev/null 2>&1
python profiles/tools/r05_dropin.py > $O/drop_in_rates.txt 2>&1
( timeout 600 python profiles/tools/fuzz_lookahead.py; timeout 600 python profiles/tools/fuzz_lookahead_r05.py ) > $O/fuzz_lookahead.log 2>&1
timeout 600 python profiles/tools/fuzz_stream.py > $O/fuzz_stream.log 2>&1
timeout 900 python profiles/tools/fuzz_batch.py 0 120 > $O/fuzz_batch.log 2>&1
timeout 900 python profiles/tools/fuzz_batch.py 1000 24 long > $O/fuzz_batch_long.log 2>&1
ls -la $O
