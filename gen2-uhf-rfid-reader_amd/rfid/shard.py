"""Sharding of independent traces over the GPUs of a node (one process per GPU).

The receive path has no exchange step: traces (RX streams / inventory captures) are
independent, so multi-GPU = contiguous partition of the batch, each rank decodes its own
traces, and only the few-hundred-byte per-rank totals are summed at the end (a control-plane
reduction of results, not a data-path collective).  Works with any torch.distributed
backend ("nccl" = RCCL on the GPUs, "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

TOTAL_FIELDS = ("n_traces", "n_windows", "n_epc_correct", "n_queries_sent", "n_terminated")


def partition(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of n_items for `rank`; sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank")
    base, extra = divmod(int(n_items), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def local_totals(stats: np.ndarray) -> np.ndarray:
    """int64 vector [len(TOTAL_FIELDS) + 256]: totals of a rank's per-trace stats records
    (rfid.capi.STATS_DTYPE) followed by the summed tag_reads histogram."""
    head = np.array([len(stats), int(stats["n_windows"].sum()), int(stats["n_epc_correct"].sum()),
                     int(stats["n_queries_sent"].sum()), int((stats["status"] == 1).sum())], dtype=np.int64)
    hist = stats["tag_reads"].astype(np.int64).sum(axis=0) if len(stats) else np.zeros(256, np.int64)
    return np.concatenate([head, hist])


def reduce_totals(vec: np.ndarray, dist=None, device=None) -> np.ndarray:
    """Sum the totals vector over all ranks (no-op without an initialised process group)."""
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return vec
    import torch
    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
