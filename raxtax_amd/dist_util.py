"""Multi-GPU plumbing for the query-sharded configuration (BASELINE.json configs[3]):
the index is replicated, every rank classifies its own contiguous shard of the queries, and the only
collective is the gather of the per-rank result records on rank 0 (RCCL on GPUs, gloo in the CPU test)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

REC_WIDTH = 12  # query index, lineage index, depth, local signal, 8 confidence values


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n units for `rank` of `world`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(row_off, row_lineage, row_depth, row_conf, row_local, first_query: int = 0) -> np.ndarray:
    """Result rows of one rank as a dense float64 [n_rows, REC_WIDTH] array (8 confidence levels kept)."""
    n_rows = int(row_off[-1])
    rec = np.zeros((n_rows, REC_WIDTH), dtype=np.float64)
    if n_rows == 0:
        return rec
    counts = np.diff(np.asarray(row_off, dtype=np.int64))
    rec[:, 0] = np.repeat(np.arange(len(counts), dtype=np.float64) + first_query, counts)
    rec[:, 1] = np.asarray(row_lineage[:n_rows], dtype=np.float64)
    rec[:, 2] = np.asarray(row_depth[:n_rows], dtype=np.float64)
    rec[:, 3] = np.asarray(row_local[:n_rows], dtype=np.float64)
    conf = np.asarray(row_conf)[:n_rows]
    rec[:, 4:4 + min(8, conf.shape[1])] = conf[:, :8]
    return rec


def gather_records(dist, rec: np.ndarray, rank: int, world: int, device: str = "cpu") -> Optional[List[np.ndarray]]:
    """Gathers variable-length record arrays on rank 0: all_gather of the row counts, then one
    dist.gather of buffers padded to the largest count.  Returns the per-rank arrays on rank 0."""
    import torch

    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros((cap, REC_WIDTH), dtype=torch.float64, device=device)
    if rec.shape[0]:
        buf[: rec.shape[0]] = torch.from_numpy(np.ascontiguousarray(rec)).to(device)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank != 0:
        return None
    return [g[:sizes[i]].cpu().numpy() for i, g in enumerate(gathered)]
