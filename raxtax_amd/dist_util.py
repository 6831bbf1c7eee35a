"""Multi-GPU plumbing for the query-sharded configuration (BASELINE.json configs[3]):
the index is replicated, every rank classifies its own contiguous shard of the queries, and the only
collective is the gather of the per-rank result records on rank 0 (RCCL on GPUs, gloo in the CPU test).

A rank's results travel as ONE byte buffer (written natively by rtx_result_pack, host_format.cpp):
    header  int64[4]            n_queries, n_rows, L = confidence levels per row (depth of the deepest row), version 2
    begin   int64[n_queries]    first row of every query (rows travel in the library's processing order)
    global  float64[n_queries]  global signal per query
    count   uint32[n_queries]   rows of every query
    t       uint32[n_queries]   distinct k-mers of every query
    status  uint8[n_queries]    RTX_Q_* status of every query
    lineage uint32[n_rows]      index into tree.lineages
    depth   uint8[n_rows]
    conf    uint8[n_rows][L]    confidence in hundredths (values are k/100 exactly), every level of every row
    local   float64[n_rows]     local signal
(3.8 MB per 100k single-row six-level queries instead of 9.6 MB of float64 records)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

RECORD_VERSION = 2


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n units for `rank` of `world`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(row_off, row_lineage, row_depth, row_conf, row_local, global_signal=None, row_begin=None,
                 row_count=None, t=None, status=None) -> np.ndarray:
    """Result rows of one rank as one uint8 buffer (layout in the module docstring).  Either a CSR `row_off`
    (rows in query order) or the library view's `row_begin` / `row_count`."""
    if row_begin is None:
        row_off = np.ascontiguousarray(row_off, dtype=np.int64)
        row_begin, row_count = row_off[:-1], np.diff(row_off)
    row_begin = np.ascontiguousarray(row_begin, dtype=np.int64)
    row_count = np.ascontiguousarray(row_count, dtype=np.uint32)
    n_q = len(row_begin)
    n_rows = int(row_count.sum())
    gs = np.zeros(n_q) if global_signal is None else np.asarray(global_signal[:n_q], dtype=np.float64)
    t = np.zeros(n_q, np.uint32) if t is None else np.asarray(t[:n_q], dtype=np.uint32)
    status = np.zeros(n_q, np.uint8) if status is None else np.asarray(status[:n_q], dtype=np.uint8)
    depth = np.ascontiguousarray(row_depth[:n_rows], dtype=np.uint8)
    levels = max(int(depth.max()) if n_rows else 1, 1)
    conf = np.asarray(row_conf)
    if conf.ndim == 1:                              # flat [rows][levels], as long as row_depth (callers may pass views with spare rows)
        conf = conf.reshape(max(len(row_depth), 1), -1)
    conf = conf[:n_rows, :levels]                   # slice the rows first, like every other field
    conf_u8 = np.rint(conf * 100.0).astype(np.uint8)
    if conf_u8.shape[1] < levels:
        conf_u8 = np.pad(conf_u8, ((0, 0), (0, levels - conf_u8.shape[1])))
    parts = [np.array([n_q, n_rows, levels, RECORD_VERSION], dtype=np.int64).view(np.uint8), row_begin.view(np.uint8),
             np.ascontiguousarray(gs).view(np.uint8), row_count.view(np.uint8), np.ascontiguousarray(t).view(np.uint8),
             np.ascontiguousarray(status),
             np.ascontiguousarray(row_lineage[:n_rows], dtype=np.uint32).view(np.uint8),
             depth,
             np.ascontiguousarray(conf_u8).reshape(-1),
             np.ascontiguousarray(row_local[:n_rows], dtype=np.float64).view(np.uint8)]
    return np.concatenate(parts)


def unpack_records(buf: np.ndarray) -> dict:
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    n_q, n_rows, levels, version = (int(x) for x in buf[:32].view(np.int64))
    if version != RECORD_VERSION:
        raise ValueError(f"result record version {version}, expected {RECORD_VERSION}")
    p = 32

    def take(nbytes, dtype):
        nonlocal p
        out = buf[p:p + nbytes].view(dtype)
        p += nbytes
        return out

    begin = take(8 * n_q, np.int64)
    gs = take(8 * n_q, np.float64)
    count = take(4 * n_q, np.uint32).astype(np.int64)
    t = take(4 * n_q, np.uint32)
    status = take(n_q, np.uint8)
    lineage = take(4 * n_rows, np.uint32)
    depth = take(n_rows, np.uint8)
    conf = take(n_rows * levels, np.uint8).reshape(n_rows, levels).astype(np.float64) / 100.0
    local = take(8 * n_rows, np.float64)
    # into query order
    row_off = np.zeros(n_q + 1, dtype=np.int64)
    row_off[1:] = np.cumsum(count)
    src = np.repeat(begin - row_off[:-1], count) + np.arange(n_rows)
    lineage, depth, conf, local = lineage[src], depth[src], conf[src], local[src]
    return dict(n_queries=n_q, n_rows=n_rows, row_off=row_off, global_signal=gs, t=t, status=status, row_lineage=lineage,
                row_depth=depth, row_conf=conf, row_local_signal=local)


def format_records(tree, buf: np.ndarray, labels, exact_one=None, flags: int = 0, threads: int = 0, out: Optional[np.ndarray] = None):
    """The `.out` texts of every query of one packed record buffer, formatted natively (rtx_records_format: what a writer on rank 0 does
    with the gathered records of a rank).  labels: the labels of the buffer's queries; exact_one: per query the id of its only exact
    match or 0xFFFFFFFF (raxtax.rs:73-84) or None.  out: a uint8 buffer of the caller's to format into (a writer keeps one per rank: a
    fresh 190-MB array per call is 26 000 first-touch page faults inside the call); it is used if it is large enough.
    Returns (text bytes, line_off[n + 1])."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    if len(buf) < 32 or int(buf[24:32].view(np.int64)[0]) != RECORD_VERSION:      # (first: a short or foreign buffer is an RtxError, not an IndexError)
        raise _lib.RtxError(-1, "format_records: not a record buffer of this version")
    n_q = int(buf[:8].view(np.int64)[0])
    if len(labels) != n_q:
        raise _lib.RtxError(-1, f"format_records: {len(labels)} labels for a record buffer of {n_q} queries")
    arr = labels if isinstance(labels, C.Array) else (C.c_char_p * n_q)(*[l if isinstance(l, bytes) else l.encode() for l in labels])
    ex = None if exact_one is None else np.ascontiguousarray(exact_one, dtype=np.uint32)
    exp = _lib.ptr(ex, _lib.u32p) if ex is not None else None
    off = np.zeros(n_q + 1, dtype=np.uint64)
    cap = 4 * len(buf) + (1 << 20)     # text is ~2.3 x the records on the bench workload: one pass as a rule ...
    for attempt in range(2):
        if attempt or out is None or out.dtype != np.uint8 or not out.flags.c_contiguous or len(out) < cap:
            out = np.empty(cap, dtype=np.uint8)
        else:
            cap = len(out)
        n = lib.rtx_records_format(tree._h, _lib.ptr(buf, _lib.u8p), len(buf), arr, exp, flags, out.ctypes.data_as(C.c_char_p), cap, _lib.ptr(off, _lib.u64p), threads)
        if n >= 0:
            return out[:n], off
        if n > -1024:          # a real error (RTX_ERR_*)
            _lib.check(n)
        cap = -n - 1024 + 1    # ... else the failed call has said what it needs (RTX_NEED_BASE)
    _lib.check(n)
    return out[:0], off


def pinned_bytes(n: int) -> np.ndarray:
    """A uint8 numpy buffer in page-locked host memory when CUDA is there (fast H2D/D2H of the records), else plain."""
    import torch

    if torch.cuda.is_available():
        return torch.empty((n,), dtype=torch.uint8, pin_memory=True).numpy()
    return np.empty(n, dtype=np.uint8)


def gather_start(dist, rec: np.ndarray, rank: int, world: int, device: str = "cpu", cache: Optional[dict] = None) -> dict:
    """Starts the gather of variable-length byte buffers on rank 0: all_gather of the sizes (blocking, 8 bytes per
    rank), then ONE asynchronous dist.gather of buffers padded to the largest size.  `cache` (a dict kept by the caller
    across steps; use two of them alternately when gathers overlap with the next step) holds the device and pinned
    staging buffers, so that a step costs one H2D per rank and one D2H on rank 0 at PCIe rate instead of pageable
    copies and allocations.  Returns a ticket for gather_finish."""
    import torch

    cache = {} if cache is None else cache
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    if cache.get("cap", 0) < cap:  # (re)allocate with head room
        c2 = int(cap * 1.25) + 64
        cache["cap"] = c2
        cache["buf"] = torch.zeros((c2,), dtype=torch.uint8, device=device)
        cache["gathered"] = [torch.zeros((c2,), dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
        cache["host"] = torch.from_numpy(pinned_bytes(c2 * world)) if (rank == 0 and device != "cpu") else None
    buf = cache["buf"]
    if rec.shape[0]:
        buf[: rec.shape[0]].copy_(torch.from_numpy(rec), non_blocking=True)   # rec in pinned memory: asynchronous H2D
    work = dist.gather(buf, cache["gathered"], dst=0, async_op=True)
    return dict(work=work, sizes=sizes, cache=cache, rank=rank, world=world, device=device)


def gather_finish(ticket: dict) -> Optional[List[np.ndarray]]:
    """Waits for a gather_start; on rank 0 brings the gathered buffers to (pinned) host memory and returns them."""
    import torch

    ticket["work"].wait()
    if ticket["rank"] != 0:
        return None
    cache, sizes, world = ticket["cache"], ticket["sizes"], ticket["world"]
    if ticket["device"] == "cpu":
        return [g[:sizes[i]].numpy().copy() for i, g in enumerate(cache["gathered"])]
    c2, host = cache["cap"], cache["host"]
    for i, g in enumerate(cache["gathered"]):
        host[i * c2: i * c2 + sizes[i]].copy_(g[:sizes[i]], non_blocking=True)
    torch.cuda.current_stream().synchronize()   # the copies only: a device-wide sync would wait for the classification running beside them
    return [host[i * c2: i * c2 + sizes[i]].numpy() for i in range(world)]


def gather_records(dist, rec: np.ndarray, rank: int, world: int, device: str = "cpu", cache: Optional[dict] = None) -> Optional[List[np.ndarray]]:
    """gather_start + gather_finish."""
    return gather_finish(gather_start(dist, rec, rank, world, device, cache))
