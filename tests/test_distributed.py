"""N>1 path on CPU: two gloo processes shard a batch of traces, each decodes its shard, and the
per-rank totals are summed with a (control-plane) all_reduce.  The GPU decode is replaced by
the oracle here -- this test covers the sharding / reduction logic that bench.py and
rfid.shard use, not the kernels."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_properties():
    sys.path.insert(0, os.path.join(ROOT, "gen2-uhf-rfid-reader_amd"))
    from rfid import shard
    for n in (0, 1, 7, 8, 1024, 1025):
        for w in (1, 2, 3, 8):
            parts = [shard.partition(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent(r"""
    import os, sys, json
    import numpy as np
    ROOT = sys.argv[1]
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gen2-uhf-rfid-reader_amd"))
    import torch.distributed as dist
    from rfid import shard, synth, capi
    from oracle import oracle
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = 5
    b, e = shard.partition(N, world, rank)
    stats = np.zeros(e - b, dtype=capi.STATS_DTYPE)
    for i, g in enumerate(range(b, e)):
        o = oracle.run_trace(synth.make_trace(n_rounds=1 + g % 3, seed=100 + g).samples)
        stats[i]["n_windows"] = o.n_windows
        stats[i]["n_epc_correct"] = o.state.n_epc_correct
        stats[i]["n_queries_sent"] = o.state.n_queries_sent
        stats[i]["tag_reads"][:] = np.array(o.state.tag_reads[:])
    dist.barrier()
    tot = shard.reduce_totals(shard.local_totals(stats), dist)
    if rank == 0:
        print("TOTALS " + json.dumps(tot[:5].tolist() + [int(tot[5 + 0x27])]))
    dist.destroy_process_group()
""")


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith("TOTALS ")][0]
    import json
    tot = json.loads(line[7:])
    rounds = [1 + g % 3 for g in range(5)]
    assert tot == [5, 2 * sum(rounds), sum(rounds), sum(r + 1 for r in rounds), 0, sum(rounds)]
