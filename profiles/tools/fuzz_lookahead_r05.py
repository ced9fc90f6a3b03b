# developer aid, round 5: the two new protocols of the per-block calls under many more seeds than the test suite runs --
# late filter outputs (rfid_lookahead_set_late_outputs) and the consume-ahead gate (rfid_lookahead_set_consume_ahead, both
# keyings) driven by randomised schedulers (tests/test_gpu_round5.py holds the drivers), every run against the oracle
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "gen2-uhf-rfid-reader_amd"); sys.path.insert(0, ".")
import torch; torch.cuda.is_available()
from rfid import synth
from oracle import oracle
import test_gpu_round5 as t5

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for name, fn in (("late filter outputs", lambda s: t5.test_late_filter_outputs_through_the_c_abi(oracle, synth, s)),
                 ("consume-ahead, keyed on the filter", lambda s: t5.test_gate_consumes_ahead_through_the_c_abi(oracle, synth, "filter", s)),
                 ("consume-ahead, keyed on the gate", lambda s: t5.test_gate_consumes_ahead_through_the_c_abi(oracle, synth, "gate", s))):
    ok = bad = 0
    for seed in range(first, first + count):
        try:
            fn(seed)
            ok += 1
        except Exception as e:
            bad += 1
            print(name, "seed", seed, "FAILED:", repr(e)[:300])
    print(name, ": passed", ok, "failed", bad)
