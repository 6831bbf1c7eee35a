"""Python face of libraxtax_hip.so, mirroring the reference's interface for the hot path
(same names, argument meaning and error behaviour):

    Tree.new(lineages, sequences)            src/tree.rs:46-140   (host mirror)
    parse_reference_fasta_str(text) -> Tree  src/parser.rs:46-105
    parse_query_fasta_str(text, skip)        src/parser.rs:117-154
    raxtax(queries, tree, skip_exact_matches, raw_confidence, chunk_size, sender, tsv)
                                             src/raxtax.rs:14-97  (device path)
    Index(tree).classify(...)                the body of raxtax() without string formatting

Everything device-side goes through the C ABI (include/raxtax_hip.h).  No fallbacks.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import (RTX_MAX_DEPTH, RTX_RAW_CONFIDENCE, RTX_SKIP_EXACT_MATCHES, NodesView, ResultView, RtxError, check,
                   f32p, f64p, ptr, u8p, u16p, u32p, u64p)


def _flatten(seqs: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    flat = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
    for i, s in enumerate(seqs):
        flat[int(off[i]):int(off[i + 1])] = np.asarray(s, dtype=np.uint8)
    return flat, off


class Tree:
    """Host mirror of `Tree` (src/tree.rs:36-43)."""

    def __init__(self, handle: int):
        self._h = C.c_void_p(handle)
        self._lib = _lib.load()

    def __del__(self):
        try:
            if self._h:
                self._lib.rtx_tree_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @classmethod
    def new(cls, lineages: Sequence[str], sequences: Sequence[np.ndarray]) -> "Tree":
        flat, off = _flatten(sequences)
        return cls.new_flat(lineages, flat, off)

    @classmethod
    def new_flat(cls, lineages: Sequence[str], seq_bytes: np.ndarray, seq_off: np.ndarray,
                 kmer_map: bool = True) -> "Tree":
        """kmer_map=False skips the host k-mer map; Index(tree) then builds the bitmaps on the GPU."""
        lib = _lib.load()
        enc = [l.encode() for l in lineages]
        loff = np.zeros(len(enc) + 1, dtype=np.uint64)
        if enc:
            loff[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
        lbytes = b"".join(enc)
        seq_bytes = np.ascontiguousarray(seq_bytes, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        h = C.c_void_p()
        check(lib.rtx_tree_build_ex(len(enc), lbytes, ptr(loff, u64p), ptr(seq_bytes, u8p), ptr(seq_off, u64p),
                                    0 if kmer_map else 1, C.byref(h)))
        return cls(h.value)

    def save_to_file(self, path: str) -> None:
        """Tree::save_to_file (tree.rs:147-153): the reference's bincode `.bin` database."""
        check(self._lib.rtx_tree_save_bin(self._h, str(path).encode()))

    @classmethod
    def load_from_file(cls, path: str) -> "Tree":
        """Tree::load_from_file (tree.rs:155-164)."""
        lib = _lib.load()
        h = C.c_void_p()
        check(lib.rtx_tree_load_bin(str(path).encode(), C.byref(h)))
        return cls(h.value)

    @property
    def num_tips(self) -> int:
        return self._lib.rtx_tree_num_tips(self._h)

    def lineage(self, i: int) -> str:
        return self._lib.rtx_tree_lineage(self._h, i).decode()

    @property
    def lineages(self) -> List[str]:
        return [self.lineage(i) for i in range(self.num_tips)]

    def original_index(self) -> np.ndarray:
        return np.array([self._lib.rtx_tree_original_index(self._h, i) for i in range(self.num_tips)], np.uint64)

    def csr(self) -> Tuple[np.ndarray, np.ndarray]:
        """Tree.k_mer_map as (offsets[65537], postings) -- views copied out of the handle."""
        po, pp = u64p(), u32p()
        check(self._lib.rtx_tree_kmer_csr(self._h, C.byref(po), C.byref(pp)))
        off = np.ctypeslib.as_array(po, shape=(65537,)).copy()
        tot = int(off[-1])
        post = np.ctypeslib.as_array(pp, shape=(tot,)).copy() if tot else np.zeros(0, np.uint32)
        return off, post

    def k_mer_map(self, kmer: int) -> np.ndarray:
        off, post = self.csr()
        return post[int(off[kmer]):int(off[kmer + 1])]

    def exact_matches(self, seq: np.ndarray) -> np.ndarray:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        p = u32p()
        n = self._lib.rtx_tree_exact_matches(self._h, ptr(seq, u8p), len(seq), C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def exact_matches_batch(self, bases: np.ndarray, base_off: np.ndarray):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        base_off = np.ascontiguousarray(base_off, dtype=np.uint64)
        n_q = len(base_off) - 1
        off = np.zeros(n_q + 1, dtype=np.uint64)
        ids = np.zeros(max(n_q, 16), dtype=np.uint32)
        tot = self._lib.rtx_tree_exact_matches_batch(self._h, n_q, ptr(bases, u8p), ptr(base_off, u64p), ptr(off, u64p),
                                                     ptr(ids, u32p), len(ids))
        if tot > len(ids):
            ids = np.zeros(tot, dtype=np.uint32)
            self._lib.rtx_tree_exact_matches_batch(self._h, n_q, ptr(bases, u8p), ptr(base_off, u64p), ptr(off, u64p),
                                                   ptr(ids, u32p), len(ids))
        return ids[:tot].copy(), off

    def nodes(self) -> dict:
        v = NodesView()
        check(self._lib.rtx_tree_nodes(self._h, C.byref(v)))
        n = v.n_nodes
        g = lambda p: np.ctypeslib.as_array(p, shape=(n,)).copy()
        return dict(begin=g(v.node_begin), end=g(v.node_end), first_child=g(v.node_first_child),
                    n_children=g(v.node_n_children), parent=g(v.node_parent), type=g(v.node_type))


def parse_reference_fasta_str(text: str) -> Tree:
    lib = _lib.load()
    b = text.encode()
    h = C.c_void_p()
    check(lib.rtx_tree_parse_reference_fasta(b, len(b), C.byref(h)))
    return Tree(h.value)


def parse_query_fasta_str(text: str, queries_to_skip: Sequence[str] = ()) -> List[Tuple[str, np.ndarray]]:
    lib = _lib.load()
    b = text.encode()
    skip = (C.c_char_p * max(len(queries_to_skip), 1))(*[s.encode() for s in queries_to_skip])
    h = C.c_void_p()
    check(lib.rtx_queries_parse_fasta(b, len(b), skip, len(queries_to_skip), C.byref(h)))
    try:
        n = lib.rtx_queries_len(h)
        pb, po = u8p(), u64p()
        check(lib.rtx_queries_data(h, C.byref(pb), C.byref(po)))
        off = np.ctypeslib.as_array(po, shape=(n + 1,)).copy()
        tot = int(off[-1])
        bases = np.ctypeslib.as_array(pb, shape=(tot,)).copy() if tot else np.zeros(0, np.uint8)
        return [(lib.rtx_queries_label(h, i).decode(), bases[int(off[i]):int(off[i + 1])].copy()) for i in range(n)]
    finally:
        lib.rtx_queries_destroy(h)


@dataclass
class EvaluationResult:
    """src/lineage.rs:8-14 (lineage given as index into tree.lineages)."""
    lineage: int
    node: int
    confidence_values: List[float]
    local_signal: float
    global_signal: float


class Result:
    """Arrays of one classified batch, copied out of the library-owned rtx_result_view and put into query
    order (the library stores the rows in processing order): rows of query q = row_off[q] .. row_off[q+1]."""

    def __init__(self, view: ResultView):
        nq, nr = view.n_queries, view.n_rows
        arr = lambda p, n: (np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0))
        self.n_queries = nq
        self.t = arr(view.t, nq)
        self.status = arr(view.status, nq)
        self.global_signal = arr(view.global_signal, nq)
        begin = arr(view.row_begin, nq).astype(np.int64)
        count = arr(view.row_count, nq).astype(np.int64)
        self.row_off = np.zeros(nq + 1, dtype=np.uint64)
        self.row_off[1:] = np.cumsum(count)
        # source row of every canonical row
        src = (np.repeat(begin - self.row_off[:-1].astype(np.int64), count) + np.arange(int(count.sum()))) if nq else np.zeros(0, np.int64)
        assert len(src) == nr
        self.row_lineage = arr(view.row_lineage, nr)[src]
        self.row_node = arr(view.row_node, nr)[src]
        self.row_depth = arr(view.row_depth, nr)[src]
        stride = int(view.row_conf_stride) or RTX_MAX_DEPTH     # ABI 5: the deepest lineage of the tree (0 in a hand-made view: RTX_MAX_DEPTH)
        self.row_conf = (np.ctypeslib.as_array(view.row_conf, shape=(nr, stride))[src] if nr
                         else np.zeros((0, stride)))
        self.row_local_signal = arr(view.row_local_signal, nr)[src]

    def rows(self, q: int) -> List[EvaluationResult]:
        out = []
        for r in range(int(self.row_off[q]), int(self.row_off[q + 1])):
            d = int(self.row_depth[r])
            out.append(EvaluationResult(int(self.row_lineage[r]), int(self.row_node[r]),
                                        [float(x) for x in self.row_conf[r, :d]],
                                        float(self.row_local_signal[r]), float(self.global_signal[q])))
        return out


DEFAULT_SEGMENT_CLASSES = 1   # RTX_DEFAULT_SEGMENT_CLASSES of the library (rtx_api_index.hip: g_seg_classes)


class Index:
    """Device-resident index + batch workspace of one GPU (rtx_index)."""

    def __init__(self, tree: Tree, device: int = 0, sub_batch: int = 0, prob_mode: int = 0,
                 stage_timing: bool = False, cluster: Optional[bool] = None, segment_classes=None,
                 packed_counts: Optional[bool] = None,
                 tile_skip: Optional[bool] = None, hit_pair=None, locator: Optional[bool] = None, tile_prune: Optional[bool] = None,
                 debug_taps: bool = False, device_exact: Optional[bool] = None, fine_bounds: Optional[bool] = None,
                 records: Optional[int] = None, overlap: Optional[bool] = None, two_level: Optional[int] = None,
                 prune_self_sample: Optional[bool] = None):
        self._lib = _lib.load()
        self.tree = tree
        if segment_classes is None:
            segment_classes = DEFAULT_SEGMENT_CLASSES
        check(self._lib.rtx_set_default_option(1, int(segment_classes)))   # creation-time default of the library
        h = C.c_void_p()
        check(self._lib.rtx_index_create_from_tree(device, tree._h, C.byref(h)))
        self._h = h
        self.n_refs = self._lib.rtx_index_num_refs(self._h)
        if sub_batch:
            check(self._lib.rtx_index_set_batch(self._h, sub_batch))
        if prob_mode:
            check(self._lib.rtx_index_set_option(self._h, 2, prob_mode))
        if stage_timing:
            check(self._lib.rtx_index_set_option(self._h, 6, 1))
        if cluster is not None:
            check(self._lib.rtx_index_set_option(self._h, 7, int(cluster)))
        if packed_counts is not None:
            check(self._lib.rtx_index_set_option(self._h, 8, int(packed_counts)))
        if tile_skip is not None:
            check(self._lib.rtx_index_set_option(self._h, 10, int(tile_skip)))
        if hit_pair is not None:
            check(self._lib.rtx_index_set_option(self._h, 11, int(hit_pair)))
        if locator is not None:
            check(self._lib.rtx_index_set_option(self._h, 12, int(locator)))
        if tile_prune is not None:
            check(self._lib.rtx_index_set_option(self._h, 13, int(tile_prune)))
            # NOTE: an explicit tile_prune=True also sets RTX_OPT_PRUNE_SELF_SAMPLE = 0 -- it asks for the pruned path whatever the
            # handle's self-sample said (A/B tests); the C API keeps the two options independent.  Pass prune_self_sample=True to keep the verdict.
            if tile_prune and prune_self_sample is None:
                check(self._lib.rtx_index_set_option(self._h, 22, 0))
        if debug_taps:
            check(self._lib.rtx_index_set_option(self._h, 14, 1))
        if device_exact is not None:
            check(self._lib.rtx_index_set_option(self._h, 15, int(device_exact)))
        if fine_bounds is not None:
            check(self._lib.rtx_index_set_option(self._h, 17, int(fine_bounds)))
        if records is not None:   # RTX_OPT_RECORDS: pruned queries with at most this many live tiles write records instead of counts (0: off)
            check(self._lib.rtx_index_set_option(self._h, 18, int(records)))
        if overlap is not None:   # RTX_OPT_OVERLAP: back half of a sub-batch on a second stream beside the front half of the next
            check(self._lib.rtx_index_set_option(self._h, 19, int(overlap)))
        if two_level is not None:  # RTX_OPT_TWO_LEVEL_BOUNDS: 0 = the bounds pass over blocks of 64 throughout (values above 1: the refine rule, packed)
            check(self._lib.rtx_index_set_option(self._h, 21, int(two_level)))
        if prune_self_sample is not None:  # RTX_OPT_PRUNE_SELF_SAMPLE: False = RTX_OPT_TILE_PRUNE alone decides (tests of the pruned path on real barcodes)
            check(self._lib.rtx_index_set_option(self._h, 22, int(prune_self_sample)))
        self._view = ResultView()
        self._keep = None

    @property
    def prune_verdict(self):
        """(tile pruning is on for this handle, share of (query, tile) combinations its self-sample kept live or -1): rtx_index_prune_verdict."""
        on, frac = C.c_int(), C.c_double()
        check(self._lib.rtx_index_prune_verdict(self._h, C.byref(on), C.byref(frac)))
        return bool(on.value), float(frac.value)

    @property
    def run_ahead_stats(self):
        """(batches enqueued ahead of the end of the batch before them, run-aheads abandoned) under RTX_OPT_RUN_AHEAD: rtx_index_run_ahead_stats."""
        a, b = C.c_uint64(), C.c_uint64()
        check(self._lib.rtx_index_run_ahead_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def __del__(self):
        try:
            if self._h:
                self._lib.rtx_index_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return self._lib.rtx_index_device_bytes(self._h)

    @property
    def workspace_bytes(self) -> int:
        """HBM of everything else the handle holds as of the last upload (tables, scratch sets, inputs, result arrays)."""
        return self._lib.rtx_index_workspace_bytes(self._h)

    def workspace_parts(self) -> dict:
        out = np.zeros(9, dtype=np.uint64)
        check(self._lib.rtx_index_workspace_parts(self._h, ptr(out, u64p)))
        names = ("prob_tables", "counts", "record_segments", "prefix_sums", "tile_masks_and_slot_lists", "other_scratch", "inputs_and_order", "results")
        d = {k: int(v) for k, v in zip(names, out[:8])}
        d["scratch_sets"] = int(out[8])
        return d

    # ---- staged interface -------------------------------------------------------------
    def upload(self, bases: np.ndarray, base_off: np.ndarray, exact_ids: Optional[np.ndarray] = None,
               exact_off: Optional[np.ndarray] = None):
        self.prefetch(bases, base_off, exact_ids, exact_off)
        self.activate()

    def prefetch(self, bases: np.ndarray, base_off: np.ndarray, exact_ids: Optional[np.ndarray] = None,
                 exact_off: Optional[np.ndarray] = None):
        """Stages a batch beside the current one (rtx_batch_prefetch): packed, pinned, asynchronous H2D; activate() makes it current."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        base_off = np.ascontiguousarray(base_off, dtype=np.uint64)
        n_q = len(base_off) - 1
        if exact_off is not None:
            exact_off = np.ascontiguousarray(exact_off, dtype=np.uint64)
            exact_ids = np.ascontiguousarray(exact_ids if exact_ids is not None and len(exact_ids) else
                                             np.zeros(1, np.uint32), dtype=np.uint32)
            check(self._lib.rtx_batch_prefetch(self._h, n_q, ptr(bases, u8p), ptr(base_off, u64p), ptr(exact_ids, u32p),
                                               ptr(exact_off, u64p)))
        else:
            check(self._lib.rtx_batch_prefetch(self._h, n_q, ptr(bases, u8p), ptr(base_off, u64p), None, None))

    def activate(self):
        check(self._lib.rtx_batch_activate(self._h))

    def run(self, flags: int = 0):
        check(self._lib.rtx_batch_run(self._h, flags))

    def sync(self):
        check(self._lib.rtx_batch_sync(self._h))

    def download(self, copy: bool = True):
        check(self._lib.rtx_batch_download(self._h, C.byref(self._view)))
        return Result(self._view) if copy else self._view

    def stage_times(self):
        ms = (C.c_float * len(_lib.STAGES))()
        n = (C.c_uint32 * len(_lib.STAGES))()
        check(self._lib.rtx_batch_stage_times(self._h, ms, n))
        return {s: (float(ms[i]), int(n[i])) for i, s in enumerate(_lib.STAGES)}

    def work(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._lib.rtx_batch_work(self._h, C.byref(a), C.byref(b), C.byref(c)))
        live, bounds = C.c_uint64(), C.c_uint64()
        check(self._lib.rtx_batch_work_split(self._h, C.byref(live), C.byref(bounds)))
        return dict(sum_hits=a.value, sum_query_bytes=b.value, bitmap_bytes_read=c.value, live_bytes=live.value, bounds_bytes=bounds.value)

    def prob_work(self):
        """Grid points D_q (n_q + 1) of prob.rs:43-90 summed over the queries of the last run."""
        a, b = C.c_uint64(), C.c_uint64()
        check(self._lib.rtx_batch_prob_work(self._h, C.byref(a), C.byref(b)))
        return dict(grid_points=a.value, distinct_counts=b.value)

    # ---- one call -----------------------------------------------------------------------
    def exact_matches(self, bases: np.ndarray, base_off: np.ndarray):
        """Tree.sequences.get() for every query (raxtax.rs:42) on the HOST -> (ids, offsets)."""
        return self.tree.exact_matches_batch(bases, base_off)

    @property
    def has_exact_lookup(self) -> bool:
        """The handle looks exact matches up itself when a batch comes without ids (rtx_exact.hip)."""
        return bool(self._lib.rtx_index_has_exact_lookup(self._h))

    def device_exact_matches(self):
        """Tree.sequences.get() for every query of the last download as the DEVICE found it -> (ids, offsets)."""
        po, pi = u64p(), u32p()
        check(self._lib.rtx_batch_exact_matches(self._h, C.byref(po), C.byref(pi)))
        n = self._view.n_queries
        off = np.ctypeslib.as_array(po, shape=(n + 1,)).copy()
        tot = int(off[-1])
        ids = np.ctypeslib.as_array(pi, shape=(tot,)).copy() if tot else np.zeros(0, np.uint32)
        return ids, off

    def classify(self, bases: np.ndarray, base_off: np.ndarray, exact_ids=None, exact_off=None,
                 skip_exact_matches: bool = False) -> Result:
        """exact_off = None: the exact matches are looked up on the device (has_exact_lookup), else the batch has none."""
        self.upload(bases, base_off, exact_ids, exact_off)
        self.run(RTX_SKIP_EXACT_MATCHES if skip_exact_matches else 0)
        return self.download()

    # ---- parity taps ----------------------------------------------------------------------
    def debug_kmers(self, q: int) -> np.ndarray:
        out = np.zeros(65536, dtype=np.uint16)
        t = C.c_uint32()
        check(self._lib.rtx_debug_kmers(self._h, q, ptr(out, u16p), C.byref(t)))
        return out[:t.value].copy()

    def debug_hit_counts(self, q: int) -> np.ndarray:
        out = np.zeros(self.n_refs, dtype=np.uint16)
        check(self._lib.rtx_debug_hit_counts(self._h, q, ptr(out, u16p)))
        return out

    def debug_prob_table(self, q: int, t: int):
        out = np.zeros(t + 1, dtype=np.float64)
        z = C.c_double()
        check(self._lib.rtx_debug_prob_table(self._h, q, ptr(out, f64p), C.byref(z)))
        return out, z.value

    def debug_pruned_prob_table(self, q: int, t: int):
        """table / Z of query q as the pruned run computed it, Z and the query's threshold (before any other tap)."""
        out = np.zeros(t + 1, dtype=np.float64)
        z = C.c_double()
        thr = C.c_uint32()
        check(self._lib.rtx_debug_pruned_prob_table(self._h, q, ptr(out, f64p), C.byref(z), C.byref(thr)))
        return out, z.value, int(thr.value)

    def debug_run_counts(self, q: int, t: int) -> dict:
        """Query q of the last sub-batch exactly as the (pruned) run left it -- no recount; before any other tap.  counts (0xFFFF in
        tiles hit_count did not visit), tile_live, hist (bin 0 = never-counted references + counted ones without a hit), threshold, i1."""
        counts = np.zeros(self.n_refs, dtype=np.uint16)
        live = np.zeros((self.n_refs + 8191) // 8192, dtype=np.uint8)
        hist = np.zeros(t + 1, dtype=np.uint32)
        thr, i1 = C.c_uint32(), C.c_uint32()
        check(self._lib.rtx_debug_run_counts(self._h, q, ptr(counts, u16p), ptr(live, u8p), ptr(hist, u32p), C.byref(thr), C.byref(i1)))
        nseg = C.c_uint32()
        check(self._lib.rtx_debug_run_mode(self._h, q, C.byref(nseg)))
        # record_segments > 0: the query took the records path (RTX_OPT_RECORDS) -- `counts` of its visited tiles hold the count of every
        # reference ABOVE the threshold and 0 for the others (it wrote records of those alone)
        return dict(counts=counts, tile_live=live.astype(bool), hist=hist, threshold=int(thr.value), i1=int(i1.value), record_segments=int(nseg.value))

    def debug_tile_bounds(self, q: int) -> np.ndarray:
        """The largest bound of every tile as the bounds pass left it for query q of the last sub-batch."""
        out = np.zeros((self.n_refs + 8191) // 8192, dtype=np.uint16)
        check(self._lib.rtx_debug_tile_bounds(self._h, q, ptr(out, u16p)))
        return out

    def debug_prune_detail(self, q: int) -> dict:
        """prune_kernel's view of query q (Index(debug_taps=True)): best block, its exact counts, M, threshold, i* + 1."""
        out = np.zeros(72, dtype=np.uint32)
        check(self._lib.rtx_debug_prune_detail(self._h, q, ptr(out, u32p)))
        return dict(block=int(out[0]), M=int(out[1]), threshold=int(out[2]), i1=int(out[3]), largest_bound=int(out[4]), t=int(out[5]),
                    block_counts=out[8:72].copy())

    def debug_probs(self, q: int) -> np.ndarray:
        out = np.zeros(self.n_refs, dtype=np.float64)
        check(self._lib.rtx_debug_probs(self._h, q, ptr(out, f64p)))
        return out

    def sub_batch_size(self) -> int:
        """Queries per sub-batch (kernel launch) of the uploaded batch."""
        b, n = C.c_uint32(), C.c_uint32()
        check(self._lib.rtx_batch_sub_batch(self._h, C.byref(b), C.byref(n)))
        return int(b.value)

    def last_sub_batch(self):
        """(first position, queries) of the last sub-batch of the uploaded batch in the processing order: what the taps can read."""
        first, n = C.c_uint64(), C.c_uint32()
        check(self._lib.rtx_batch_last_sub_batch(self._h, C.byref(first), C.byref(n)))
        return int(first.value), int(n.value)

    def batch_classes(self) -> list:
        """Length classes of the uploaded batch: dicts of queries, longest query, sub-batch size, planes and what the class runs through."""
        n = C.c_uint32()
        out = np.zeros(20, dtype=np.uint64)
        check(self._lib.rtx_batch_classes(self._h, C.byref(n), ptr(out, u64p)))
        res = []
        for c in range(int(n.value)):
            f = int(out[4 * c + 3])
            res.append(dict(queries=int(out[4 * c]), longest=int(out[4 * c + 1]), sub_batch=int(out[4 * c + 2]), planes=f & 0xFF, tables=bool(f >> 8 & 1),
                            pair=bool(f >> 9 & 1), prune=bool(f >> 10 & 1), records=bool(f >> 11 & 1), global_memory_forms=bool(f >> 12 & 1)))
        return res

    def debug_order(self, n_queries: int) -> np.ndarray:
        """Processing order of the last run: perm[position] = query."""
        out = np.zeros(n_queries, dtype=np.uint32)
        check(self._lib.rtx_debug_order(self._h, ptr(out, u32p)))
        return out

    def debug_prune_stats(self) -> dict:
        """Tile pruning of the last run: live tiles per pair, mean lower bound of the best hit, mean threshold, mean largest tile bound."""
        out = np.zeros(16, dtype=np.uint64)
        check(self._lib.rtx_debug_prune_stats(self._h, ptr(out, u64p)))
        pairs, nq = max(int(out[1]), 1), max(int(out[5]), 1)
        counted = int(out[12]) if int(out[11]) or int(out[12]) else int(out[0])   # after the fine bounds pass, if it ran
        return {"live_tiles_per_pair": counted / pairs, "live_tiles_per_pair_first_stage": int(out[0]) / pairs,
                "fine_blocks_per_pair": int(out[11]) / pairs, "fine_cleared_per_query": int(out[10]) / nq, "pairs": int(out[1]), "mean_best_hit_lower_bound": int(out[2]) / nq,
                "mean_threshold": int(out[3]) / nq, "mean_largest_tile_bound": int(out[4]) / nq, "bound_violations": int(out[6]), "live_tiles_per_query": int(out[7]) / nq,
                "tiles_above_threshold_per_query": int(out[8]) / max(int(out[9]), 1), "queries_with_threshold": int(out[9]),
                "record_queries": int(out[14]), "records_per_record_query": int(out[13]) / max(int(out[14]), 1), "record_slow_path_queries": int(out[15])}

    def debug_evaluate(self, probs) -> Result:
        """Lineage::new(label, tree, probs).evaluate() on the device (lineage.rs:61-112)."""
        probs = np.ascontiguousarray(probs, dtype=np.float64)
        assert len(probs) == self.n_refs
        check(self._lib.rtx_debug_evaluate(self._h, ptr(probs, f64p), C.byref(self._view)))
        return Result(self._view)

    def format_query(self, q: int, label: str, seq: np.ndarray, exact_ids: np.ndarray, flags: int = 0,
                     tsv: bool = False):
        """Lines of query q of the last download (after override raxtax.rs:73-84)."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ex = np.ascontiguousarray(exact_ids if len(exact_ids) else np.zeros(1, np.uint32), dtype=np.uint32)
        cap = 1 << 20
        out = C.create_string_buffer(cap)
        tbuf = C.create_string_buffer(cap) if tsv else None
        tlen = C.c_int64()
        n = self._lib.rtx_format_query(self.tree._h, C.byref(self._view), q, label.encode(), ptr(seq, u8p), len(seq),
                                       ptr(ex, u32p), len(exact_ids), flags, out, cap, tbuf, cap, C.byref(tlen))
        check(n)
        return out.raw[:n].decode(), (tbuf.raw[:tlen.value].decode() if tsv else None)


_SENDER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p)


def raxtax(queries: Sequence[Tuple[str, np.ndarray]], tree, skip_exact_matches: bool, raw_confidence: bool,
           chunk_size: int, sender: Callable[[str, str, Optional[str]], None], tsv: bool) -> None:
    """src/raxtax.rs:14-22 -- same arguments; `tree` is the device Index built from the Tree, or a list of them (one per GPU,
    all built from the same Tree): rtx_raxtax_multi then deals the chunks to the handles, one driving thread each.
    `sender(label, out_lines, tsv_lines_or_None)` is called once per query, in input order; raising from it
    plays the role of a closed channel."""
    lib = _lib.load()
    lib.rtx_raxtax.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_char_p), u8p, u64p, C.c_int, C.c_int,
                               C.c_uint64, _SENDER, C.c_void_p, C.c_int]
    lib.rtx_raxtax_multi.argtypes = [C.POINTER(C.c_void_p), C.c_uint32] + lib.rtx_raxtax.argtypes[1:]
    handles = list(tree) if isinstance(tree, (list, tuple)) else [tree]
    labels = (C.c_char_p * max(len(queries), 1))(*[q[0].encode() for q in queries])
    flat, off = _flatten([q[1] for q in queries])
    err: List[BaseException] = []

    def cb(_ctx, label, out, tsv_lines):
        try:
            sender(label.decode(), out.decode(), tsv_lines.decode() if tsv_lines is not None else None)
            return 0
        except BaseException as e:  # noqa: BLE001 - forwarded below
            err.append(e)
            return 1

    if len(handles) == 1:
        rc = lib.rtx_raxtax(handles[0]._h, handles[0].tree._h, len(queries), labels, ptr(flat, u8p), ptr(off, u64p),
                            int(skip_exact_matches), int(raw_confidence), chunk_size, _SENDER(cb), None, int(tsv))
    else:
        arr = (C.c_void_p * len(handles))(*[h._h.value for h in handles])
        rc = lib.rtx_raxtax_multi(arr, len(handles), handles[0].tree._h, len(queries), labels, ptr(flat, u8p), ptr(off, u64p),
                                  int(skip_exact_matches), int(raw_confidence), chunk_size, _SENDER(cb), None, int(tsv))
    if err:
        raise err[0]
    check(rc)


def raxtax_last_timing() -> Tuple[List[float], int]:
    """rtx_raxtax_last_timing: busy seconds of the pipeline stages (host lookup, device, format, sender) and the number of chunks of the
    last rtx_raxtax / rtx_raxtax_multi call of this process."""
    lib = _lib.load()
    busy = (C.c_double * 4)()
    n_chunks = C.c_uint64(0)
    check(lib.rtx_raxtax_last_timing(busy, C.byref(n_chunks)))
    return [float(b) for b in busy], int(n_chunks.value)
