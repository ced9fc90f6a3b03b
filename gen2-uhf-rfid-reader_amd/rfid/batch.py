"""Bulk offline decoding of recorded traces.

Trace files are what the reference's flowgraph reads and writes: headerless little-endian
interleaved float32 I,Q (blocks.file_source / file_sink, apps/reader.py:68-72,102-103;
misc/code/plot_signal.m:5-9).  Traces are packed into one [n_traces][stride] HBM buffer
(pinned host staging, asynchronous copies) and decoded by one rfid_batch_process() pass.
torch is used to own the pinned / device memory only.
"""
from __future__ import annotations

import time
from typing import List, Optional, Sequence

import numpy as np

from .context import Context


def read_trace_file(path: str) -> np.ndarray:
    """Interleaved float32 I,Q file -> complex64 array (zero-copy view of the file's bytes)."""
    return np.fromfile(path, dtype=np.complex64)


def write_trace_file(path: str, samples: np.ndarray) -> None:
    np.ascontiguousarray(samples, dtype=np.complex64).tofile(path)


class BatchDecoder:
    """Decode many independent traces per pass on one GPU."""

    def __init__(self, device: int = 0, **params):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("rfid.batch needs a GPU: the receive path has no CPU fallback")
        self._torch = torch
        self.device = int(device)
        self.ctx = Context(device=device, **params)
        self._planned = (0, 0)
        self._dev = None
        self._pinned = None
        self._lens_dev = None

    def close(self) -> None:
        self.ctx.close()

    def _ensure(self, n_traces: int, max_len: int) -> int:
        torch = self._torch
        stride = (max_len + 1) & ~1
        if self._planned[0] < n_traces or self._planned[1] < max_len:
            self.ctx.batch_plan(max(n_traces, self._planned[0]), max(max_len, self._planned[1]))
            self._planned = (max(n_traces, self._planned[0]), max(max_len, self._planned[1]))
        # the plan may be larger than this batch (decoder reuse): process exactly n_traces rows
        self.ctx.batch_set_streams(n_traces)
        need = n_traces * stride * 2
        if self._dev is None or self._dev.numel() < need:
            self._dev = torch.empty(need, dtype=torch.float32, device=f"cuda:{self.device}")
            self._pinned = torch.empty(need, dtype=torch.float32, pin_memory=True)
        return stride

    def decode(self, traces: Sequence[np.ndarray], want_scores: bool = False, timing: Optional[dict] = None):
        """traces: list of complex64 arrays (ragged).  Returns (stats, windows, results, scores).

        `timing` (optional dict) receives h2d_s / gpu_s / total_s of this call."""
        torch = self._torch
        n = len(traces)
        lens = np.array([len(t) for t in traces], dtype=np.int64)
        max_len = int(lens.max()) if n else 0
        if n == 0 or max_len == 0:
            raise ValueError("no samples")
        stride = self._ensure(n, max_len)
        t0 = time.perf_counter()
        host = self._pinned[: n * stride * 2].numpy().view(np.complex64).reshape(n, stride)
        for i, t in enumerate(traces):
            host[i, : len(t)] = t
        dev = self._dev[: n * stride * 2]
        with torch.cuda.device(self.device):
            dev.copy_(self._pinned[: n * stride * 2], non_blocking=True)
            self._lens_dev = torch.from_numpy(lens).to(dev.device, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        t1 = time.perf_counter()
        self.ctx.batch_process_ptr(dev.data_ptr(), stride, max_len, self._lens_dev.data_ptr(), want_scores=want_scores)
        self.ctx.batch_sync()
        t2 = time.perf_counter()
        stats = self.ctx.batch_stats()[:n]
        w, r, s = self.ctx.batch_windows(want_scores=want_scores)
        if timing is not None:
            timing.update(h2d_s=t1 - t0, gpu_s=t2 - t1, total_s=time.perf_counter() - t0,
                          raw_samples=int(lens.sum()))
        return stats, w, r, s

    def decode_files(self, paths: Sequence[str], **kw):
        return self.decode([read_trace_file(p) for p in paths], **kw)


def summarize(stats: np.ndarray) -> List[dict]:
    """Per-trace READER_STATS as dicts (what reader_impl::print_results reports)."""
    out = []
    for s in stats:
        out.append(dict(n_queries_sent=int(s["n_queries_sent"]) - 1, cur_inventory_round=int(s["cur_inventory_round"]),
                        n_epc_correct=int(s["n_epc_correct"]), n_unique_tags=int(s["n_unique_tags"]),
                        tag_reads={i: int(c) for i, c in enumerate(s["tag_reads"]) if c}))
    return out


def format_results(stats_row) -> str:
    """The text reader_impl::print_results writes (lib/reader_impl.cc:173-192) for one trace of a batch."""
    s = stats_row
    lines = ["", " --------------------------", "| Number of queries/queryreps sent : %d" % (int(s["n_queries_sent"]) - 1),
             "| Current Inventory round : %d" % int(s["cur_inventory_round"]), " --------------------------",
             "| Correctly decoded EPC : %d" % int(s["n_epc_correct"]),
             "| Number of unique tags : %d" % int(s["n_unique_tags"])]
    for i, c in enumerate(s["tag_reads"]):
        if c:
            lines.append("| Tag ID : %x  Num of reads : %d" % (i, int(c)))
    lines.append(" --------------------------")
    return "\n".join(lines) + "\n"


def main(argv=None) -> int:
    """python -m rfid.batch [--device N] [--fixed-q Q] TRACE_FILE...  -- decode recorded traces in one batched pass."""
    import argparse
    ap = argparse.ArgumentParser(prog="python -m rfid.batch", description=main.__doc__)
    ap.add_argument("files", nargs="+")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--fixed-q", type=int, default=0)
    ap.add_argument("--max-queries", type=int, default=1000)
    args = ap.parse_args(argv)
    dec = BatchDecoder(device=args.device, fixed_q=args.fixed_q, max_num_queries=args.max_queries)
    try:
        timing = {}
        stats, _, _, _ = dec.decode_files(args.files, timing=timing)
        for path, row in zip(args.files, stats):
            print(path)
            print(format_results(row), end="")
        print("%d traces, %.1f M raw samples: %.3f s (host->HBM %.3f s, GPU pass %.4f s)" %
              (len(args.files), timing["raw_samples"] / 1e6, timing["total_s"], timing["h2d_s"], timing["gpu_s"]))
    finally:
        dec.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
