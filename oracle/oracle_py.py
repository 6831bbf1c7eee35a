"""ctypes binding for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (raxtax_amd) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
MAXD = 64


class OrcRow(C.Structure):
    _fields_ = [
        ("idx", C.c_uint64),
        ("depth", C.c_uint32),
        ("pad", C.c_uint32),
        ("conf", C.c_double * MAXD),
        ("expd", C.c_double * MAXD),
        ("local_signal", C.c_double),
        ("global_signal", C.c_double),
    ]


def build(native: bool = False, force: bool = False) -> Path:
    """Compile the oracle with gcc.  native=True builds a -march=native copy (CPU baseline)."""
    out = HERE / ("liboracle_native.so" if native else "liboracle.so")
    src = [HERE / "oracle.c", HERE / "oracle.h"]
    if not force and out.exists() and all(out.stat().st_mtime >= s.stat().st_mtime for s in src):
        return out
    march = "native" if native else "x86-64-v3"
    subprocess.check_call(
        ["make", "-C", str(HERE), f"MARCH={march}", f"OUT={out.name}", "-B", out.name],
        stdout=subprocess.DEVNULL,
    )
    return out


def build_info(native: bool = False) -> dict:
    """Compiler and flags of build(native): what `cpu_baseline` prints beside its numbers."""
    march = "native" if native else "x86-64-v3"
    try:
        cc = subprocess.check_output(["gcc", "--version"], text=True).splitlines()[0]
    except (OSError, subprocess.CalledProcessError):
        cc = "gcc (version unknown)"
    return {"compiler": cc, "flags": f"-O3 -march={march} -std=gnu11 -fPIC -fopenmp (oracle/Makefile)"}


_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


class Oracle:
    def __init__(self, native: bool = False):
        path = build(native=native)
        self.lib = L = C.CDLL(str(path))
        L.orc_map_four_to_two_bit_repr.restype = C.c_int
        L.orc_map_four_to_two_bit_repr.argtypes = [C.c_uint8]
        L.orc_sequence_to_kmers.restype = C.c_uint32
        L.orc_sequence_to_kmers.argtypes = [_u8p, C.c_uint64, _u16p]
        L.orc_decompress_sequence.argtypes = [_u8p, C.c_uint64, C.c_char_p]
        for f in (L.orc_euclidean_distance_l1, L.orc_cosine_similarity):
            f.restype = C.c_double
            f.argtypes = [_f64p, _f64p, C.c_uint64]
        L.orc_euclidean_norm.restype = C.c_double
        L.orc_euclidean_norm.argtypes = [_f64p, C.c_uint64]
        L.orc_ln_gamma.restype = C.c_double
        L.orc_ln_gamma.argtypes = [C.c_double]
        L.orc_ln_factorial.restype = C.c_double
        L.orc_ln_factorial.argtypes = [C.c_uint64]
        L.orc_ln_binomial.restype = C.c_double
        L.orc_ln_binomial.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_map_dna_char.restype = C.c_int
        L.orc_map_dna_char.argtypes = [C.c_int]
        L.orc_parse_reference_fasta_str.restype = C.c_void_p
        L.orc_parse_reference_fasta_str.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.orc_parse_query_fasta_str.restype = C.c_void_p
        L.orc_parse_query_fasta_str.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint64,
                                                C.POINTER(C.c_int)]
        L.orc_queries_len.restype = C.c_uint64
        L.orc_queries_len.argtypes = [C.c_void_p]
        L.orc_queries_label.restype = C.c_char_p
        L.orc_queries_label.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_queries_seq.restype = C.c_uint64
        L.orc_queries_seq.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_u8p)]
        L.orc_queries_free.argtypes = [C.c_void_p]
        L.orc_tree_new.restype = C.c_void_p
        L.orc_tree_new.argtypes = [C.c_uint64, C.POINTER(C.c_char_p), _u8p, _u64p]
        L.orc_tree_free.argtypes = [C.c_void_p]
        L.orc_tree_num_tips.restype = C.c_uint64
        L.orc_tree_num_tips.argtypes = [C.c_void_p]
        L.orc_tree_lineage.restype = C.c_char_p
        L.orc_tree_lineage.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_tree_original_index.restype = C.c_uint64
        L.orc_tree_original_index.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_tree_kmer_list.restype = C.c_uint64
        L.orc_tree_kmer_list.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_u32p)]
        L.orc_tree_total_postings.restype = C.c_uint64
        L.orc_tree_total_postings.argtypes = [C.c_void_p]
        L.orc_tree_export_csr.argtypes = [C.c_void_p, _u64p, _u32p]
        L.orc_tree_exact_matches.restype = C.c_uint64
        L.orc_tree_exact_matches.argtypes = [C.c_void_p, _u8p, C.c_uint64, C.POINTER(_u32p)]
        L.orc_tree_num_nodes.restype = C.c_uint64
        L.orc_tree_num_nodes.argtypes = [C.c_void_p]
        L.orc_tree_export_nodes.argtypes = [C.c_void_p, _u64p, _u64p, _i64p, _u8p, _u32p]
        L.orc_tree_node_label.restype = C.c_char_p
        L.orc_tree_node_label.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_hit_counts.restype = C.c_uint32
        L.orc_hit_counts.argtypes = [C.c_void_p, _u8p, C.c_uint64, C.c_int, _u16p]
        L.orc_highest_hit_prob_per_reference.restype = C.c_int
        L.orc_highest_hit_prob_per_reference.argtypes = [C.c_uint16, C.c_uint64, _u16p, C.c_uint64, _f64p]
        L.orc_prob_table.restype = C.c_int
        L.orc_prob_table.argtypes = [C.c_uint16, C.c_uint64, _u16p, C.c_uint64, _f64p, _f64p]
        L.orc_iterative_pmf_ln.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, _f64p]
        L.orc_lineage_evaluate.restype = C.c_int
        L.orc_lineage_evaluate.argtypes = [C.c_void_p, _f64p, C.POINTER(OrcRow), C.c_int]
        L.orc_classify.restype = C.c_int
        L.orc_classify.argtypes = [C.c_void_p, _u8p, C.c_uint64, C.c_int, C.c_int, C.POINTER(OrcRow), C.c_int]
        L.orc_format_out.restype = C.c_int64
        L.orc_format_out.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(OrcRow), C.c_int, C.c_char_p, C.c_uint64]
        L.orc_format_tsv.restype = C.c_int64
        L.orc_format_tsv.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(OrcRow), C.c_int, _u8p, C.c_uint64,
                                     C.c_char_p, C.c_uint64]
        L.orc_classify_batch.restype = C.c_int64
        L.orc_classify_batch.argtypes = [C.c_void_p, C.c_uint64, _u8p, _u64p, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(OrcRow), C.c_int, _i32p, C.c_int]
        L.orc_classify_batch_ex.restype = C.c_int64
        L.orc_classify_batch_ex.argtypes = [C.c_void_p, C.c_uint64, _u8p, _u64p, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(OrcRow), C.c_int, _i32p, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.orc_physical_core_ids.restype = C.c_int
        L.orc_physical_core_ids.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.orc_hit_counts_batch.restype = None
        L.orc_hit_counts_batch.argtypes = [C.c_void_p, C.c_uint64, _u8p, _u64p, C.c_int, C.c_int, _u16p, _u32p]
        L.orc_prob_tables_batch.restype = None
        L.orc_prob_tables_batch.argtypes = [C.c_uint64, _u32p, _u16p, C.c_uint64, C.c_int, _f64p, C.c_uint64, _f64p, _i32p]

    def physical_core_ids(self):
        """utils.rs:160-197: one logical CPU per physical core, over the CPUs this process may run on."""
        ids = (C.c_int * 4096)()
        n = self.lib.orc_physical_core_ids(ids, 4096)
        return [int(ids[i]) for i in range(n)]

    def prob_tables_batch(self, t_arr, counts, threads: int = 1):
        """table[m]/Z of prob.rs:8-103 for every row of `counts` -> (tables [n_q][tmax+1], z [n_q], rc [n_q])."""
        t_arr = np.ascontiguousarray(t_arr, dtype=np.uint32)
        counts = np.ascontiguousarray(counts, dtype=np.uint16)
        n_q, n_refs = counts.shape
        stride = int(t_arr.max()) + 1 if n_q else 1
        tables = np.zeros((n_q, stride), dtype=np.float64)
        z = np.zeros(n_q, dtype=np.float64)
        rc = np.zeros(n_q, dtype=np.int32)
        self.lib.orc_prob_tables_batch(n_q, _ptr(t_arr, _u32p), _ptr(counts, _u16p), n_refs, threads, _ptr(tables, _f64p),
                                       stride, _ptr(z, _f64p), _ptr(rc, _i32p))
        return tables, z, rc

    # ---- utils -----------------------------------------------------------
    def map_four_to_two_bit_repr(self, c: int):
        r = self.lib.orc_map_four_to_two_bit_repr(c)
        return None if r < 0 else r

    def sequence_to_kmers(self, seq) -> np.ndarray:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        out = np.zeros(max(len(seq), 1), dtype=np.uint16)
        n = self.lib.orc_sequence_to_kmers(_ptr(seq, _u8p), len(seq), _ptr(out, _u16p))
        return out[:n].copy()

    def decompress_sequence(self, seq) -> str:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        buf = C.create_string_buffer(len(seq) + 1)
        self.lib.orc_decompress_sequence(_ptr(seq, _u8p), len(seq), buf)
        return buf.value.decode()

    def euclidean_distance_l1(self, a, b) -> float:
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert len(a) == len(b)
        return self.lib.orc_euclidean_distance_l1(_ptr(a, _f64p), _ptr(b, _f64p), len(a))

    def euclidean_norm(self, v) -> float:
        v = np.ascontiguousarray(v, dtype=np.float64)
        return self.lib.orc_euclidean_norm(_ptr(v, _f64p), len(v))

    def cosine_similarity(self, a, b) -> float:
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        return self.lib.orc_cosine_similarity(_ptr(a, _f64p), _ptr(b, _f64p), len(a))

    def ln_binomial(self, n, k):
        return self.lib.orc_ln_binomial(n, k)

    def map_dna_char(self, ch: str):
        r = self.lib.orc_map_dna_char(ord(ch))
        return None if r < 0 else r

    # ---- parser / tree -----------------------------------------------------
    def parse_reference_fasta_str(self, s: str) -> "OracleTree":
        err = C.c_int(0)
        h = self.lib.orc_parse_reference_fasta_str(s.encode(), C.byref(err))
        if not h:
            raise ValueError(f"reference FASTA parse error {err.value}")
        return OracleTree(self, h)

    def parse_query_fasta_str(self, s: str, skip=()):
        err = C.c_int(0)
        arr = (C.c_char_p * max(len(skip), 1))(*[x.encode() for x in skip])
        h = self.lib.orc_parse_query_fasta_str(s.encode(), arr, len(skip), C.byref(err))
        if not h:
            raise ValueError(f"query FASTA parse error {err.value}")
        out = []
        for i in range(self.lib.orc_queries_len(h)):
            p = _u8p()
            n = self.lib.orc_queries_seq(h, i, C.byref(p))
            out.append((self.lib.orc_queries_label(h, i).decode(),
                        np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint8)))
        self.lib.orc_queries_free(h)
        return out

    def tree_new(self, lineages, sequences) -> "OracleTree":
        """Tree::new(lineages, sequences) -- sequences: list of uint8 arrays (4-bit one-hot codes)."""
        n = len(lineages)
        assert n == len(sequences)
        off = np.zeros(n + 1, dtype=np.uint64)
        for i, s in enumerate(sequences):
            off[i + 1] = off[i] + len(s)
        flat = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
        for i, s in enumerate(sequences):
            flat[int(off[i]):int(off[i + 1])] = np.asarray(s, dtype=np.uint8)
        arr = (C.c_char_p * max(n, 1))(*[x.encode() for x in lineages])
        h = self.lib.orc_tree_new(n, arr, _ptr(flat, _u8p), _ptr(off, _u64p))
        if not h:
            raise ValueError("tree_new failed")
        return OracleTree(self, h)

    def tree_new_flat(self, lineages, flat: np.ndarray, off: np.ndarray) -> "OracleTree":
        n = len(lineages)
        flat = np.ascontiguousarray(flat, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        arr = (C.c_char_p * max(n, 1))(*[x.encode() for x in lineages])
        h = self.lib.orc_tree_new(n, arr, _ptr(flat, _u8p), _ptr(off, _u64p))
        if not h:
            raise ValueError("tree_new failed")
        return OracleTree(self, h)

    # ---- prob ----------------------------------------------------------------
    def highest_hit_prob_per_reference(self, t: int, n: int, sizes) -> np.ndarray:
        sizes = np.ascontiguousarray(sizes, dtype=np.uint16)
        out = np.zeros(len(sizes), dtype=np.float64)
        rc = self.lib.orc_highest_hit_prob_per_reference(t, n, _ptr(sizes, _u16p), len(sizes), _ptr(out, _f64p))
        if rc < 0:
            raise ArithmeticError(f"reference would panic (code {rc})")
        return out

    def prob_table(self, t: int, n: int, sizes):
        sizes = np.ascontiguousarray(sizes, dtype=np.uint16)
        table = np.zeros(t + 1, dtype=np.float64)
        z = C.c_double(0.0)
        rc = self.lib.orc_prob_table(t, n, _ptr(sizes, _u16p), len(sizes), _ptr(table, _f64p), C.byref(z))
        if rc < 0:
            raise ArithmeticError(f"reference would panic (code {rc})")
        return table, z.value

    def iterative_pmf_ln(self, t: int, n: int, m: int, ln_total: float) -> np.ndarray:
        out = np.zeros(n + 1, dtype=np.float64)
        self.lib.orc_iterative_pmf_ln(t, n, m, ln_total, _ptr(out, _f64p))
        return out


class OracleTree:
    def __init__(self, orc: Oracle, handle):
        self.orc = orc
        self.h = C.c_void_p(handle)

    def __del__(self):
        try:
            if self.h:
                self.orc.lib.orc_tree_free(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def num_tips(self) -> int:
        return self.orc.lib.orc_tree_num_tips(self.h)

    @property
    def lineages(self):
        return [self.orc.lib.orc_tree_lineage(self.h, i).decode() for i in range(self.num_tips)]

    def lineage(self, i: int) -> str:
        return self.orc.lib.orc_tree_lineage(self.h, i).decode()

    def original_index(self) -> np.ndarray:
        return np.array([self.orc.lib.orc_tree_original_index(self.h, i) for i in range(self.num_tips)],
                        dtype=np.uint64)

    def kmer_list(self, kmer: int) -> np.ndarray:
        p = _u32p()
        n = self.orc.lib.orc_tree_kmer_list(self.h, kmer, C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def csr(self):
        off = np.zeros(65537, dtype=np.uint64)
        tot = self.orc.lib.orc_tree_total_postings(self.h)
        post = np.zeros(max(tot, 1), dtype=np.uint32)
        self.orc.lib.orc_tree_export_csr(self.h, _ptr(off, _u64p), _ptr(post, _u32p))
        return off, post[:tot]

    def exact_matches(self, seq) -> np.ndarray:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        p = _u32p()
        n = self.orc.lib.orc_tree_exact_matches(self.h, _ptr(seq, _u8p), len(seq), C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def nodes(self):
        n = self.orc.lib.orc_tree_num_nodes(self.h)
        lo = np.zeros(n, np.uint64)
        hi = np.zeros(n, np.uint64)
        parent = np.zeros(n, np.int64)
        typ = np.zeros(n, np.uint8)
        nch = np.zeros(n, np.uint32)
        self.orc.lib.orc_tree_export_nodes(self.h, _ptr(lo, _u64p), _ptr(hi, _u64p), _ptr(parent, _i64p),
                                           _ptr(typ, _u8p), _ptr(nch, _u32p))
        labels = [self.orc.lib.orc_tree_node_label(self.h, i).decode() for i in range(n)]
        return dict(lo=lo, hi=hi, parent=parent, type=typ, n_children=nch, label=labels)

    def hit_counts(self, seq, skip_exact: bool = False):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        counts = np.zeros(self.num_tips, dtype=np.uint16)
        t = self.orc.lib.orc_hit_counts(self.h, _ptr(seq, _u8p), len(seq), int(skip_exact), _ptr(counts, _u16p))
        return t, counts

    @staticmethod
    def _rows(rows, n):
        out = []
        for i in range(n):
            r = rows[i]
            out.append(dict(idx=int(r.idx), conf=[r.conf[k] for k in range(r.depth)],
                            expd=[r.expd[k] for k in range(r.depth)],
                            local_signal=r.local_signal, global_signal=r.global_signal))
        return out

    def lineage_evaluate(self, probs, cap: int = 512):
        probs = np.ascontiguousarray(probs, dtype=np.float64)
        assert len(probs) == self.num_tips
        rows = (OrcRow * cap)()
        n = self.orc.lib.orc_lineage_evaluate(self.h, _ptr(probs, _f64p), rows, cap)
        if n < 0:
            raise OverflowError("row capacity exceeded")
        return self._rows(rows, n)

    def classify(self, seq, skip_exact=False, raw_confidence=False, cap: int = 512):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        rows = (OrcRow * cap)()
        n = self.orc.lib.orc_classify(self.h, _ptr(seq, _u8p), len(seq), int(skip_exact), int(raw_confidence),
                                      rows, cap)
        if n < 0:
            raise ArithmeticError(f"reference would panic (code {n})")
        return self._rows(rows, n), (rows, n)

    def format_out(self, label: str, raw_rows) -> str:
        rows, n = raw_rows
        buf = C.create_string_buffer(1 << 20)
        w = self.orc.lib.orc_format_out(self.h, label.encode(), rows, n, buf, len(buf))
        assert w >= 0
        return buf.raw[:w].decode()

    def format_tsv(self, label: str, raw_rows, seq) -> str:
        rows, n = raw_rows
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        buf = C.create_string_buffer(1 << 20)
        w = self.orc.lib.orc_format_tsv(self.h, label.encode(), rows, n, _ptr(seq, _u8p), len(seq), buf, len(buf))
        assert w >= 0
        return buf.raw[:w].decode()

    def classify_batch(self, bases: np.ndarray, base_off: np.ndarray, skip_exact=False, raw_confidence=False,
                       threads: int = 1, cap: int = 0, format_strings: bool = False, pin_cpus=None):
        """Returns (#would-panic, rows or None, n_rows or None).  cap=0: results discarded (timing).
        pin_cpus: one logical CPU per pool thread (`--pin`, utils.rs:139-158)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        base_off = np.ascontiguousarray(base_off, dtype=np.uint64)
        n_q = len(base_off) - 1
        pins = (C.c_int * len(pin_cpus))(*pin_cpus) if pin_cpus else None
        n_pin = len(pin_cpus) if pin_cpus else 0
        if cap:
            rows = (OrcRow * (cap * n_q))()
            n_rows = np.zeros(n_q, dtype=np.int32)
            bad = self.orc.lib.orc_classify_batch_ex(self.h, n_q, _ptr(bases, _u8p), _ptr(base_off, _u64p),
                                                     int(skip_exact), int(raw_confidence), threads, rows, cap,
                                                     _ptr(n_rows, _i32p), int(format_strings), pins, n_pin)
            return bad, rows, n_rows
        bad = self.orc.lib.orc_classify_batch_ex(self.h, n_q, _ptr(bases, _u8p), _ptr(base_off, _u64p),
                                                 int(skip_exact), int(raw_confidence), threads, None, 256, None,
                                                 int(format_strings), pins, n_pin)
        return bad, None, None

    def hit_counts_batch(self, bases: np.ndarray, base_off: np.ndarray, skip_exact=False, threads: int = 1):
        """raxtax.rs:41,55-68 for every query -> (t [n_q], counts [n_q][num_tips])."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        base_off = np.ascontiguousarray(base_off, dtype=np.uint64)
        n_q = len(base_off) - 1
        counts = np.zeros((n_q, self.num_tips), dtype=np.uint16)
        t = np.zeros(n_q, dtype=np.uint32)
        self.orc.lib.orc_hit_counts_batch(self.h, n_q, _ptr(bases, _u8p), _ptr(base_off, _u64p), int(skip_exact), threads,
                                          _ptr(counts, _u16p), _ptr(t, _u32p))
        return t, counts

    def rows_of(self, rows, n_rows, q: int, cap: int):
        """Rows of query q out of classify_batch(cap=cap) as dicts (None where the reference would panic)."""
        n = int(n_rows[q])
        if n < 0:
            return None
        out = []
        for i in range(n):
            r = rows[q * cap + i]
            out.append(dict(idx=int(r.idx), conf=[r.conf[k] for k in range(r.depth)],
                            expd=[r.expd[k] for k in range(r.depth)],
                            local_signal=r.local_signal, global_signal=r.global_signal))
        return out
