#!/usr/bin/env python3
"""bench.py -- Gen2 receive path (matched_filter -> gate -> tag_decoder) on MI355X.

One "step" = one pass of the whole hot path over a batch of synthetic traces already
resident in HBM (BASELINE.json configs[1]: 1024 noise-replicas of the 71-round
file_source_test stand-in, FM0 40 kHz BLF @ 2 Msps, per GPU).  Prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams B]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU, traces sharded per rank, no data-path collective (weak
scaling); torch.distributed is used only for the barrier and the max-over-ranks time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gen2-uhf-rfid-reader_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)
RN16_WIN, EPC_WIN = 250, 1370


def build_batch(torch, ctx, device, n_streams, sigma, seed, rank):
    """Noise-free 71-round stand-in trace, replicated n_streams times in HBM with per-replica noise by the library's
    device-side generator (rfid_synth_replicas; replica index = rank * n_streams + row, so every GPU of a multi-GPU
    run holds different replicas).  torch only owns the buffers."""
    from rfid import synth
    base = synth.make_trace(n_rounds=71, fixed_q=0, tag_ids=(0x27,), sigma=0.0, seed=7,
                            corrupt_rounds=(36,), noise=False)
    L = len(base.samples)
    stride = (L + 1) & ~1
    base_dev = torch.from_numpy(base.samples.view(np.float32).copy()).to(device)     # [2*L] float32
    data = torch.zeros((n_streams, 2 * stride), dtype=torch.float32, device=device)
    torch.cuda.synchronize()     # the library runs on its own (non-blocking) stream: torch's fill must be over
    ctx.synth_replicas_ptr(base_dev.data_ptr(), L, data.data_ptr(), stride, n_streams, sigma, seed,
                           first_replica=rank * n_streams)
    ctx.batch_sync()
    del base_dev
    return data, L, stride, base


def _usable_cores():
    """Host threads worth starting: the affinity mask, capped by a cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.999)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, int(q / int(f2.read().split()[0]) + 0.999)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(base_samples, sigma, seconds_target):
    """The oracle (CPU port of the reference algorithm) timed on this host on a bounded sample: repeated passes
    over ONE replica of the same workload -- first on one thread (the reference is single-threaded per stream;
    per-stage split), then one independent copy per host core at once (`value`, `cores`)."""
    from oracle import oracle
    rng = np.random.default_rng(123)
    n = rng.standard_normal((len(base_samples), 2), dtype=np.float32)
    x = (base_samples + np.float32(sigma) * (n[:, 0] + 1j * n[:, 1])).astype(np.complex64)
    t = oracle.time_trace(x, reps=1)
    t_pass = max(t["total_s"], 1e-4)
    reps1 = max(1, min(20000, int(0.5 * seconds_target / t_pass)))
    t = oracle.time_trace(x, reps=reps1)
    msps1 = len(x) * reps1 / t["total_s"] / 1e6
    cores = _usable_cores()
    # calibrate the all-cores leg with one pass per thread, then size it to ~half the time budget of wall time
    cal = oracle.time_trace_mt(x, reps=1, nthreads=cores)
    reps_mt = max(1, min(20000, int(0.5 * seconds_target / max(cal["wall_s"], 1e-4))))
    m = oracle.time_trace_mt(x, reps=reps_mt, nthreads=cores)
    msps = len(x) * reps_mt * cores / m["wall_s"] / 1e6
    return {"value": round(msps, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{cores} host threads x {reps_mt} passes over 1 replica of the workload trace ({len(x)} raw samples "
                      f"each) in {m['wall_s']:.2f} s wall, oracle/rfid_oracle.c; one thread alone: {reps1} passes, "
                      f"FIR {t['fir_s']:.2f}s + gate/decoder {t['gate_decoder_s']:.2f}s",
            "single_thread_msamples_per_s": round(msps1, 3),
            "epc_per_s": round(m["n_epc_correct"] * reps_mt * cores / m["wall_s"], 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=1024, help="traces per GPU")
    ap.add_argument("--sigma", type=float, default=0.002)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import rfid

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the receive path has no CPU fallback")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with "
                         f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"bench.py --gpus {args.gpus} ...")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("RFID_BENCH_FORCE_DIST"):   # (the env var lets a 1-GPU box exercise the RCCL path)
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    n_gpus = world

    B = args.streams
    ctx = rfid.Context(device=local_rank)
    data, L, stride, base = build_batch(torch, ctx, device, B, args.sigma, args.seed, rank)
    ctx.batch_plan(B, L)
    ptr = data.data_ptr()

    def step():
        ctx.batch_process_ptr(ptr, stride, L, 0, want_scores=False)
        ctx.batch_sync()

    for _ in range(args.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    k_ms = {"mf_ms": 0.0, "gate_ms": 0.0, "decode_ms": 0.0, "stats_ms": 0.0, "front_ms": 0.0}
    launches = {"front_chunks": 1, "decode_launches": 2}
    fused = False
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = ctx.batch_timing()      # HIP events on the ctx stream, recorded around each kernel
        for k in k_ms:
            k_ms[k] += t[k]
        launches = {"front_chunks": int(t["front_chunks"]), "decode_launches": int(t["decode_launches"])}
        fused = bool(t["fused_front"])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)     # control plane only: max time over ranks
        elapsed = float(tt.item())
    for k in k_ms:
        k_ms[k] /= max(args.steps, 1)

    # ---- result checks (size-independent properties of the workload) ----------------------
    st = ctx.batch_stats()
    n_epc_ok = int(st["n_epc_correct"].sum())
    n_windows = int(st["n_windows"].sum())
    expect_ok = 70 * B
    parity_ok = bool((st["n_epc_correct"] == 70).all() and (st["n_queries_sent"] == 72).all()
                     and (st["tag_reads"][:, 0x27] == 70).all() and (st["n_unique_tags"] == 1).all())

    # ---- roofline: algorithmic bytes per launch / measured kernel time
    n_dec = L // 5
    n_rn16 = int(n_windows // 2 + n_windows % 2)
    n_epc = int(n_windows // 2)
    dec_bytes = n_rn16 * (8.0 * RN16_WIN + 48) + n_epc * (8.0 * EPC_WIN + 48)
    if fused:
        # rfid_batch_process() default: one front-end launch (matched filter inside the gate's producer
        # waves): reads every raw sample once, writes y once for the decoder, writes the window records
        alg = {"front_end_fused": B * (8.0 * L + 8.0 * n_dec) + 24.0 * n_windows, "tag_decoder": dec_bytes}
        dur_ms = {"front_end_fused": k_ms["gate_ms"], "tag_decoder": k_ms["decode_ms"]}
        n_launch = {"front_end_fused": 1, "tag_decoder": launches["decode_launches"]}
    else:
        alg = {"mf_boxcar25_decim5": B * (8.0 * L + 8.0 * n_dec),
               "gate_scan": B * 8.0 * n_dec + 24.0 * n_windows, "tag_decoder": dec_bytes}
        dur_ms = {"mf_boxcar25_decim5": k_ms["mf_ms"], "gate_scan": k_ms["gate_ms"], "tag_decoder": k_ms["decode_ms"]}
        n_launch = {"mf_boxcar25_decim5": launches["front_chunks"], "gate_scan": launches["front_chunks"],
                    "tag_decoder": launches["decode_launches"]}
    traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        if pt.get("streams") == B and pt.get("raw_per_stream") == L:
            traffic = pt.get("hbm_bytes_per_step", {})
    except Exception:
        pass

    def roof(name):
        # a pass issues n_launch launches of this kernel (time chunks / window types), each moving
        # 1/n_launch of the bytes: the per-launch ratio equals the per-pass ratio
        nl = max(1, n_launch[name])
        ach = alg[name] / (dur_ms[name] * 1e-3) / 1e9 if dur_ms[name] > 0 else 0.0
        tr = traffic.get(name)
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": (int(tr / nl) if tr else None),
                "algorithmic_bytes": int(alg[name] / nl), "launches_per_step": nl,
                "avg_launch_ms": round(dur_ms[name] / nl, 4), "ms_per_step": round(dur_ms[name], 4)}

    dominant = max(dur_ms, key=lambda k: dur_ms[k])
    total_raw = float(B) * L * args.steps * n_gpus
    out = {
        "metric": "I/Q Msamples/s through matched_filter->gate->tag_decoder (+ EPC decodes/s), 40 kHz FM0",
        "value": round(total_raw / elapsed / 1e6, 2),
        "unit": "Msamples/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: batch of %d noise-replicas per GPU of the 71-round file_source_test "
                               "stand-in trace, FM0 40 kHz BLF @ 2 Msps (%d raw I/Q samples each, sigma=%g), "
                               "resident in HBM" % (B, L, args.sigma),
                   "streams_per_gpu": B, "raw_samples_per_stream": L, "parallelism": "traces sharded per GPU, no collective"},
        "epc_decodes_per_s": round(n_epc_ok * n_gpus * args.steps / elapsed, 1),
        "decoder_gated_msamples_per_s": round((n_rn16 * RN16_WIN + n_epc * EPC_WIN) / (k_ms["decode_ms"] * 1e-3) / 1e6, 1)
        if k_ms["decode_ms"] > 0 else None,
        "windows_per_step": n_windows,
        "parity_check": "ok: every replica 70/71 EPC, tag 0x27" if parity_ok else
                        "FAILED: %d EPC ok, expected %d" % (n_epc_ok, expect_ok),
        "roofline": roof(dominant),
        "roofline_by_kernel": {k: roof(k) for k in alg},
        "front_end_ms": round(k_ms["front_ms"], 4),
    }
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(base.samples, args.sigma, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if not parity_ok:
        raise SystemExit(2)


if __name__ == "__main__":
    main()
