"""Reference-sharded database (BASELINE.json configs[4], SURVEY.md 8e mode B).

The lineage-sorted references are cut into contiguous ranges, one per GPU; every rank classifies the
SAME queries against its range.  Per sub-batch two exchanges make the result identical to the
unsharded one:

  1. hit-count histograms      all-reduce(sum) of [n][t+1] uint32  -> every rank computes the same
                               probability table (prob.rs:13-103 needs the histogram of ALL references)
  2. boundary prefix sums      all-gather of [n][n_bnd_local] float64; rank s's prefix is offset by the
                               totals of the ranks before it -> prefix sums over the whole database at
                               the taxonomy boundaries (lineage.rs:61-66), then the walk (lineage.rs:119-179)

Communication volume is independent of the number of references per k-mer: 2.6 KB + 8 n_bnd bytes per
query.  The exchanges go through a small `Comm` interface: `TorchComm` (torch.distributed, backend
"nccl" = RCCL over xGMI, one process per GPU) in production; `LocalComm` runs several shards inside one
process (tests on a single GPU).  The library side is the staged C ABI rtx_shard_* (include/raxtax_hip.h).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import check, ptr, u8p, u32p, u64p
from .api import Index, Result, Tree

RTX_BUF_HIST, RTX_BUF_PREFIX, RTX_BUF_COUNTS, RTX_BUF_BEST = 1, 2, 3, 4


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, addr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (addr, False),
                                         "version": 2, "strides": None}


def device_tensor(addr: int, shape, typestr: str, device: int):
    import torch

    return torch.as_tensor(_DevArray(addr, shape, typestr), device=f"cuda:{device}")


def shard_cuts(n_refs: int, world: int) -> List[int]:
    """Balanced contiguous cut points [0, ..., n_refs] (world + 1 values)."""
    base, rem = divmod(n_refs, world)
    cuts = [0]
    for r in range(world):
        cuts.append(cuts[-1] + base + (1 if r < rem else 0))
    return cuts


class _StagedMixin:
    """The staged C ABI (rtx_shard_*) of a handle: kernels of one sub-batch in three groups, exchanges in between."""

    def begin(self):
        n_sub, b = C.c_uint32(), C.c_uint32()
        check(self._lib.rtx_shard_begin(self._h, C.byref(n_sub), C.byref(b)))
        return n_sub.value, b.value

    def count(self, sb: int, flags: int = 0):
        check(self._lib.rtx_shard_count(self._h, sb, flags))

    @property
    def prunes(self) -> bool:
        """After begin(): this run prunes tiles with the threshold of the whole database (RTX_OPT_SHARD_PRUNE): bounds(sb), the
        exchange of RTX_BUF_BEST, then count(sb)."""
        return bool(self._lib.rtx_shard_prunes(self._h))

    def bounds(self, sb: int, flags: int = 0):
        check(self._lib.rtx_shard_bounds(self._h, sb, flags))

    def rehist(self, sb: int):
        check(self._lib.rtx_shard_rehist(self._h, sb))

    def prob(self, sb: int):
        check(self._lib.rtx_shard_prob(self._h, sb))

    def walk(self, sb: int, prefix_global):
        assert prefix_global.is_contiguous() and prefix_global.shape[1] == self.n_bnd
        check(self._lib.rtx_shard_walk(self._h, sb, C.c_void_p(prefix_global.data_ptr())))

    def buffer(self, which: int, n_rows: int, sb: int = 0):
        """Device buffer of sub-batch sb (a staged run alternates between two scratch sets) as a torch tensor."""
        p, stride = C.c_void_p(), C.c_uint64()
        check(self._lib.rtx_shard_buffer(self._h, sb, which, C.byref(p), C.byref(stride)))
        if which == RTX_BUF_COUNTS:
            # u16 counts, seen as int32 pairs: RCCL and gloo reduce no 16-bit integers, and adding the pairs adds both
            # halves correctly -- the summed counts stay <= t <= 65535, so the low half never carries into the high one
            return device_tensor(p.value, (n_rows, stride.value // 2), "<i4", self.device)
        typ = {RTX_BUF_HIST: "<i4", RTX_BUF_PREFIX: "<f8", RTX_BUF_BEST: "<i4"}[which]
        return device_tensor(p.value, (n_rows, stride.value), typ, self.device)

    @property
    def torch_stream(self):
        """The HIP stream of the handle as a torch stream: collectives issued under `torch.cuda.stream(...)` of it are
        ordered against the library's kernels on the device, without host synchronisation."""
        import torch

        if getattr(self, "_tstream", None) is None:
            p = C.c_void_p()
            check(self._lib.rtx_index_stream(self._h, C.byref(p)))
            self._tstream = torch.cuda.ExternalStream(p.value, device=torch.device("cuda", self.device))
        return self._tstream

    def _shard_info(self):
        lo, hi, ng, nl, fb = C.c_uint64(), C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(self._lib.rtx_shard_info(self._h, C.byref(lo), C.byref(hi), C.byref(ng), C.byref(nl), C.byref(fb)))
        self.ref_lo, self.ref_hi, self.n_bnd, self.n_bnd_local, self.first_bnd = lo.value, hi.value, ng.value, nl.value, fb.value


class ShardIndex(_StagedMixin, Index):
    """Device index holding references [cuts[rank], cuts[rank+1]) of the tree (SURVEY.md 8e mode B)."""

    def __init__(self, tree: Tree, rank: int, cuts: Sequence[int], device: int = 0, sub_batch: int = 1024, tile_prune: bool = True):
        self._lib = _lib.load()
        self.tree = tree
        self.device = device
        self.rank = rank
        off, post = tree.csr()
        nd = tree.nodes()
        cuts_a = np.asarray(cuts, dtype=np.uint64)
        post = post if len(post) else np.zeros(1, np.uint32)
        h = C.c_void_p()
        check(self._lib.rtx_index_create_shard(
            device, tree.num_tips, int(cuts[rank]), int(cuts[rank + 1]), ptr(cuts_a, u64p), len(cuts_a),
            ptr(off, u64p), ptr(post, u32p), len(nd["type"]), ptr(nd["begin"], u32p), ptr(nd["end"], u32p),
            ptr(nd["first_child"], u32p), ptr(nd["n_children"], u32p), ptr(nd["type"], u8p), C.byref(h)))
        self._h = h
        self.n_refs = int(cuts[rank + 1]) - int(cuts[rank])      # local references (debug taps)
        if sub_batch:
            check(self._lib.rtx_index_set_batch(self._h, sub_batch))
        # a shard of 4 tiles or more counts only the tiles that can matter, with the threshold of the whole database (RTX_OPT_SHARD_PRUNE)
        self.tile_prune = bool(tile_prune)
        check(self._lib.rtx_index_set_option(self._h, 16, int(tile_prune)))
        from ._lib import ResultView

        self._view = ResultView()
        self._keep = None
        self._shard_info()


def kmer_cuts(tree_off: np.ndarray, world: int) -> List[int]:
    """Cut points of the 65 536 k-mers that give every rank about the same number of postings (world + 1 values)."""
    total = int(tree_off[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(tree_off, total * r // world, side="left")))
    cuts.append(65536)
    return [min(max(c, 0), 65536) for c in cuts]


class KmerShardIndex(_StagedMixin, Index):
    """Device index holding the posting lists of k-mers [kcuts[rank], kcuts[rank+1]) over ALL references
    (SURVEY.md 8e mode A, the literal wording of BASELINE.json configs[4]): its hit counts are partial sums that the
    ranks all-reduce.  Counts travel as u16 here (RTX_OPT_PACKED_COUNTS = 0: packed counts cannot be added)."""

    def __init__(self, tree: Tree, rank: int, kcuts: Sequence[int], device: int = 0, sub_batch: int = 256):
        self._lib = _lib.load()
        self.tree = tree
        self.device = device
        self.rank = rank
        off, post = tree.csr()
        nd = tree.nodes()
        lo, hi = int(kcuts[rank]), int(kcuts[rank + 1])
        a, b = int(off[lo]), int(off[hi])
        off_r = np.zeros(65537, dtype=np.uint64)
        off_r[lo:hi + 1] = off[lo:hi + 1] - np.uint64(a)
        off_r[hi + 1:] = np.uint64(b - a)
        post_r = np.ascontiguousarray(post[a:b]) if b > a else np.zeros(1, np.uint32)
        h = C.c_void_p()
        check(self._lib.rtx_index_create(device, tree.num_tips, ptr(off_r, u64p), ptr(post_r, u32p), len(nd["type"]),
                                         ptr(nd["begin"], u32p), ptr(nd["end"], u32p), ptr(nd["first_child"], u32p),
                                         ptr(nd["n_children"], u32p), ptr(nd["type"], u8p), C.byref(h)))
        self._h = h
        self.n_refs = tree.num_tips
        check(self._lib.rtx_index_set_option(self._h, 8, 0))           # u16 counts
        if sub_batch:
            check(self._lib.rtx_index_set_batch(self._h, sub_batch))
        from ._lib import ResultView

        self._view = ResultView()
        self._keep = None
        self._shard_info()


class LocalComm:
    """All shards live in this process (single-GPU tests): the collectives are plain tensor ops."""

    def allreduce_hist(self, hists):
        total = hists[0].clone()
        for h in hists[1:]:
            total += h
        for h in hists:
            h.copy_(total)

    allreduce_counts = allreduce_hist

    def agree_min(self, value: int, device=None) -> int:
        return int(value)       # every shard lives in this process: the caller has taken the minimum already

    def allgather_prefix(self, locals_):
        return [p.clone() for p in locals_]

    def select_best(self, bests):
        """Per query the candidate with the largest bound (column 0) over the shards, ties to the lowest shard, into every buffer."""
        import torch

        allb = torch.stack([b.to(bests[0].device) for b in bests])            # [shards][n][66]
        pick = torch.argmax(allb[:, :, 0], dim=0)                             # the first maximum: the lowest shard
        chosen = allb[pick, torch.arange(allb.shape[1], device=allb.device)]
        for b in bests:
            b.copy_(chosen.to(b.device))


class TorchComm:
    """One shard per process; torch.distributed (backend nccl = RCCL on GPUs, gloo on CPU tensors)."""

    def __init__(self, dist, world: int, widths: Sequence[int]):
        self.dist, self.world, self.widths = dist, world, list(widths)

    def allreduce_hist(self, hists):
        (h,) = hists           # int32 view of the uint32 histogram (counts < 2^31)
        self.dist.all_reduce(h)

    def allreduce_start(self, t):
        """Asynchronous all-reduce(sum): ordered behind the work already on the current stream; wait() makes the
        current stream (not the host) wait for it."""
        return self.dist.all_reduce(t, async_op=True)

    allreduce_counts = allreduce_hist   # int32 view of pairs of u16 counts (_StagedMixin.buffer)

    def agree_min(self, value: int, device: Optional[int] = None) -> int:
        """The minimum of an integer over the ranks: all-reduce(min) of one int32 (on the shard's GPU under nccl = RCCL, which
        reduces device memory only; on the host under gloo)."""
        import torch

        on_gpu = str(self.dist.get_backend()).lower() == "nccl"
        if on_gpu:
            dev = torch.cuda.current_device() if device is None else int(device)
            t = torch.tensor([int(value)], dtype=torch.int32, device=torch.device("cuda", dev))
        else:
            t = torch.tensor([int(value)], dtype=torch.int32)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def allgather_prefix(self, locals_):
        import torch

        (p,) = locals_
        wmax = max(self.widths)
        pad = torch.zeros((p.shape[0], wmax), dtype=p.dtype, device=p.device)
        pad[:, : p.shape[1]] = p
        out = [torch.zeros_like(pad) for _ in range(self.world)]
        self.dist.all_gather(out, pad)
        return [o[:, : self.widths[s]] for s, o in enumerate(out)]

    def select_best(self, bests):
        """All-gather of the shards' candidates for the best block ([n][66] int32: 264 B per query and shard), then per query the one
        with the largest bound, ties to the lowest rank -- the same choice on every rank."""
        import torch

        (b,) = bests
        out = [torch.zeros_like(b) for _ in range(self.world)]
        self.dist.all_gather(out, b.contiguous())
        allb = torch.stack(out)
        pick = torch.argmax(allb[:, :, 0], dim=0)
        b.copy_(allb[pick, torch.arange(allb.shape[1], device=allb.device)])


def assemble_prefix(parts):
    """Concatenates per-shard prefix sums ([n][w_s], column 0 = 0, last column = shard total) into the prefix
    over the whole database: [n][1 + sum_s (w_s - 1)]."""
    import torch

    n = parts[0].shape[0]
    cols = [torch.zeros((n, 1), dtype=parts[0].dtype, device=parts[0].device)]
    offset = torch.zeros((n,), dtype=parts[0].dtype, device=parts[0].device)
    for p in parts:
        cols.append(p[:, 1:] + offset[:, None])
        offset = offset + p[:, -1]
    return torch.cat(cols, dim=1).contiguous()


class ShardedClassifier:
    """Drives one (TorchComm) or several (LocalComm) ShardIndex handles through the staged path."""

    def __init__(self, shards: Sequence[ShardIndex], comm):
        self.shards, self.comm = list(shards), comm

    def upload(self, bases: np.ndarray, base_off: np.ndarray, exact_ids=None, exact_off=None):
        """Queries (and their exact-match ids) to every shard's HBM; they stay resident for any number of run()s."""
        for s in self.shards:
            # an earlier batch may have switched this shard to plain counting (a transient cause: a failed scratch allocation, a batch of
            # long reads): every upload starts from what the shard was built with, the verdict below is taken anew
            check(s._lib.rtx_index_set_option(s._h, 16, int(getattr(s, "tile_prune", True))))
            s.upload(bases, base_off, exact_ids, exact_off)
        self._n_q = len(base_off) - 1
        self._agree_on_pruning(bases, base_off, exact_ids, exact_off)

    def _agree_on_pruning(self, bases, base_off, exact_ids, exact_off):
        """Every shard prunes, or none does.  Whether a shard CAN prune is local state (a union bitmap exists only from
        RTX_PRUNE_MIN_TILES local tiles on and if its allocation succeeded; the scratch of the bounds must have fitted) -- balanced
        cuts can leave one rank a tile short of the limit -- but a pruning shard processes the queries in min-hash order and exchanges
        best blocks, a counting one keeps the input order and does not: mixed, the histogram all-reduce would add rows of different
        queries and the collectives would not pair up.  So the shards take the minimum of their verdicts; a shard that
        could prune alone is switched to plain counting (RTX_OPT_SHARD_PRUNE = 0 drops its upload: uploaded again)."""
        local = []
        for s in self.shards:
            s.begin()
            local.append(int(s.prunes))
        agreed = self.comm.agree_min(min(local), device=getattr(self.shards[0], "device", None))
        for s, mine in zip(self.shards, local):
            if mine and not agreed:
                check(s._lib.rtx_index_set_option(s._h, 16, 0))
                s.upload(bases, base_off, exact_ids, exact_off)
        self._prunes = bool(agreed)

    def classify(self, bases: np.ndarray, base_off: np.ndarray, exact_ids=None, exact_off=None,
                 skip_exact_matches: bool = False) -> Result:
        self.upload(bases, base_off, exact_ids, exact_off)
        return self.run(skip_exact_matches)

    def run(self, skip_exact_matches: bool = False, copy: bool = True):
        """One pass over the uploaded queries; returns the Result (copy=True) or the library-owned view."""
        if len(self.shards) == 1 and hasattr(self.comm, "allreduce_start"):
            return self._run_pipelined(skip_exact_matches, copy)
        return self._run_blocking(skip_exact_matches, copy)

    def _run_pipelined(self, skip_exact_matches: bool, copy: bool):
        """One shard per process (TorchComm): nothing synchronises the host between the sub-batches.  The handle's HIP
        stream is the current torch stream, so the collectives are ordered on the device behind the kernels they depend
        on; the histogram all-reduce of sub-batch i is started BEFORE sub-batch i + 1 is counted (its own scratch set)
        and waited for (by the stream) behind it: the exchange overlaps with the dominant kernel."""
        import torch

        (s,) = self.shards
        flags = _lib.RTX_SKIP_EXACT_MATCHES if skip_exact_matches else 0
        n_sub, B = s.begin()
        n_q = self._n_q
        keep = []
        prunes = s.prunes
        assert prunes == self._prunes, "this shard's tile pruning differs from what the ranks agreed on at the upload"

        def count(sb):
            if prunes:   # bounds -> the best block of the whole database -> counting of the live tiles
                s.bounds(sb, flags)
                self.comm.select_best([s.buffer(RTX_BUF_BEST, min(B, n_q - sb * B), sb)])
            s.count(sb, flags)
        with torch.cuda.stream(s.torch_stream):
            count(0)
            for sb in range(n_sub):
                nq = min(B, n_q - sb * B)
                work = self.comm.allreduce_start(s.buffer(RTX_BUF_HIST, nq, sb))
                if sb + 1 < n_sub:
                    count(sb + 1)
                work.wait()
                s.prob(sb)
                pref = assemble_prefix(self.comm.allgather_prefix([s.buffer(RTX_BUF_PREFIX, nq, sb)]))
                s.walk(sb, pref)
                keep = [pref] + keep[:1]      # stream-ordered reuse by the allocator; two generations kept for clarity
        return s.download(copy=copy)

    def _run_blocking(self, skip_exact_matches: bool, copy: bool):
        import torch

        flags = _lib.RTX_SKIP_EXACT_MATCHES if skip_exact_matches else 0
        n_sub, B = self.shards[0].begin()
        for s in self.shards[1:]:
            assert s.begin() == (n_sub, B), "all shards must use the same sub-batch size"
        n_q = self._n_q
        prunes = self._prunes
        assert all(s.prunes == prunes for s in self.shards), "all shards must agree on the tile pruning"
        for sb in range(n_sub):
            nq = min(B, n_q - sb * B)
            if prunes:
                for s in self.shards:
                    s.bounds(sb, flags)
                for s in self.shards:
                    s.sync()
                self.comm.select_best([s.buffer(RTX_BUF_BEST, nq, sb) for s in self.shards])
                torch.cuda.synchronize()
            for s in self.shards:
                s.count(sb, flags)
            for s in self.shards:
                s.sync()
            self.comm.allreduce_hist([s.buffer(RTX_BUF_HIST, nq, sb) for s in self.shards])
            torch.cuda.synchronize()
            for s in self.shards:
                s.prob(sb)
            for s in self.shards:
                s.sync()
            parts = self.comm.allgather_prefix([s.buffer(RTX_BUF_PREFIX, nq, sb) for s in self.shards])
            pref = assemble_prefix(parts)
            torch.cuda.synchronize()
            for s in self.shards:
                s.walk(sb, pref)
            for s in self.shards:
                s.sync()
        return self.shards[0].download(copy=copy)


class KmerShardedClassifier:
    """SURVEY.md 8e mode A: the k-mer space is sharded, every rank counts ALL references against its k-mers, and the
    per-reference hit counts are all-reduced (u16 [n][N] per sub-batch: 2 N bytes per query over xGMI -- the literal
    wording of BASELINE.json configs[4]; mode B above moves 2.6 KB per query instead).  After the all-reduce every
    rank holds the complete counts and finishes the classification on its own; rank 0's results are reported."""

    def __init__(self, shards: Sequence[KmerShardIndex], comm):
        self.shards, self.comm = list(shards), comm

    def upload(self, bases, base_off, exact_ids=None, exact_off=None):
        for s in self.shards:
            s.upload(bases, base_off, exact_ids, exact_off)
        self._n_q = len(base_off) - 1

    def classify(self, bases, base_off, exact_ids=None, exact_off=None, skip_exact_matches: bool = False) -> Result:
        self.upload(bases, base_off, exact_ids, exact_off)
        return self.run(skip_exact_matches)

    def run(self, skip_exact_matches: bool = False, copy: bool = True):
        import torch

        flags = _lib.RTX_SKIP_EXACT_MATCHES if skip_exact_matches else 0
        n_sub, B = self.shards[0].begin()
        for s in self.shards[1:]:
            assert s.begin() == (n_sub, B)
        n_q = self._n_q
        for sb in range(n_sub):
            nq = min(B, n_q - sb * B)
            for s in self.shards:
                s.count(sb, flags)
            for s in self.shards:
                s.sync()
            self.comm.allreduce_counts([s.buffer(RTX_BUF_COUNTS, nq, sb) for s in self.shards])
            torch.cuda.synchronize()
            for s in self.shards:
                s.rehist(sb)
                s.prob(sb)
                s.walk(sb, s.buffer(RTX_BUF_PREFIX, nq, sb))
            for s in self.shards:
                s.sync()
        return self.shards[0].download(copy=copy)
