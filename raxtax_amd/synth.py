"""Deterministic synthetic inputs for tests and bench.py.

"phylo" model: a root COI-length sequence evolves down a 7-level taxonomy (phylum, class,
order, family, genus, species, individual) with per-site substitution probabilities
mu = (0.06, 0.05, 0.04, 0.03, 0.03, 0.02, 0.005); a substituted site is redrawn from the
base composition p(A,C,G,T) = (0.263, 0.169, 0.143, 0.425) measured on the reference's
example/diptera_queries.fasta.  Length 658, substitutions only.  Queries: a uniformly
chosen reference mutated with mu_q = 0.02 per site; 10 % exact copies; 1 % carry 1-3 `N`.

Sequences use the reference's in-memory encoding (src/parser.rs:11-34): one byte per base,
A=1 C=2 G=4 T=8, N=15.  PRNG: numpy PCG64 seeded per stage (root=1, db=2, queries=3), so
the same seeds give the same data on every box with this image's numpy.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

BASE_P = np.array([0.263, 0.169, 0.143, 0.425])
MU = (0.06, 0.05, 0.04, 0.03, 0.03, 0.02, 0.005)
LEVEL_PREFIX = ("p:P", "c:C", "o:O", "f:F", "g:G", "s:S")
ONE_HOT = np.array([1, 2, 4, 8], dtype=np.uint8)
COI_LEN = 658


def default_fanouts(n_refs: int) -> Tuple[int, ...]:
    if n_refs <= 2_000:
        return (2, 2, 2, 2, 2, 2)          # 64 species (small test databases)
    if n_refs <= 100_000:
        return (3, 3, 3, 4, 4, 3)          # 1 296 species
    if n_refs <= 1_000_000:
        return (3, 4, 5, 6, 6, 6)          # 12 960 species
    return (4, 5, 6, 9, 10, 12)            # 129 600 species


def _draw(rng: np.random.Generator, shape) -> np.ndarray:
    """Bases 0..3 drawn from BASE_P."""
    u = rng.random(shape, dtype=np.float32)
    c = np.cumsum(BASE_P).astype(np.float32)
    return (u >= c[0]).astype(np.uint8) + (u >= c[1]) + (u >= c[2])


def _mutate(rng: np.random.Generator, seqs: np.ndarray, mu: float) -> np.ndarray:
    mask = rng.random(seqs.shape, dtype=np.float32) < mu
    out = seqs.copy()
    n = int(mask.sum())
    if n:
        out[mask] = _draw(rng, n)
    return out


@dataclass
class SynthDB:
    lineages: List[str]        # one per reference, input order
    seq_bytes: np.ndarray      # uint8 one-hot codes, concatenated
    seq_off: np.ndarray        # uint64 [n+1]
    length: int

    @property
    def n(self) -> int:
        return len(self.lineages)

    def seq(self, i: int) -> np.ndarray:
        return self.seq_bytes[int(self.seq_off[i]):int(self.seq_off[i + 1])]

    def fasta(self) -> str:
        dec = np.frombuffer(b"?AC?G???T??????N", dtype=np.uint8)
        lines = []
        for i, lin in enumerate(self.lineages):
            lines.append(f">r{i};tax={lin};")
            lines.append(dec[self.seq(i)].tobytes().decode())
        return "\n".join(lines) + "\n"


def write_fasta(path, headers: Sequence[str], seq_bytes: np.ndarray, seq_off: np.ndarray, block: int = 50_000) -> int:
    """Writes sequences in the reference's in-memory encoding (parser.rs:11-34) as FASTA: one header line (`>` + headers[i]) and one
    sequence line per record -- what `raxtax -d` / `-i` read (parser.rs:46-154).  Returns the bytes written."""
    dec = np.frombuffer(b"?AC?G???T??????N", dtype=np.uint8)
    n = len(headers)
    written = 0
    with open(path, "wb") as fh:
        for a in range(0, n, block):
            b = min(n, a + block)
            parts = []
            for i in range(a, b):
                parts.append(b">" + headers[i].encode() + b"\n")
                parts.append(dec[seq_bytes[int(seq_off[i]):int(seq_off[i + 1])]].tobytes() + b"\n")
            chunk = b"".join(parts)
            fh.write(chunk)
            written += len(chunk)
    return written


def make_db(n_refs: int, length: int = COI_LEN, fanouts: Optional[Sequence[int]] = None, seed_root: int = 1,
            seed_db: int = 2) -> SynthDB:
    fan = tuple(fanouts) if fanouts is not None else default_fanouts(n_refs)
    assert len(fan) == 6
    root = _draw(np.random.default_rng(seed_root), (1, length))
    rng = np.random.default_rng(seed_db)
    level = root
    labels = [""]
    for d in range(6):
        level = _mutate(rng, np.repeat(level, fan[d], axis=0), MU[d])
        new_labels = []
        k = 0
        for lab in labels:
            for _ in range(fan[d]):
                new_labels.append((lab + "," if lab else "") + f"{LEVEL_PREFIX[d]}{k}")
                k += 1
        labels = new_labels
    n_species = level.shape[0]
    per = np.full(n_species, n_refs // n_species, dtype=np.int64)
    per[: n_refs - int(per.sum())] += 1
    seq_bytes = np.empty(n_refs * length, dtype=np.uint8)
    out = seq_bytes.reshape(n_refs, length)
    pos = 0
    chunk = max(1, 4_000_000 // length)
    sp_idx = np.repeat(np.arange(n_species), per)
    for a in range(0, n_refs, chunk):
        b = min(n_refs, a + chunk)
        out[a:b] = ONE_HOT[_mutate(rng, level[sp_idx[a:b]], MU[6])]
        pos = b
    lineages = [labels[s] for s in sp_idx]
    seq_off = (np.arange(n_refs + 1, dtype=np.uint64) * np.uint64(length))
    return SynthDB(lineages, seq_bytes, seq_off, length)


@dataclass
class SynthQueries:
    labels: List[str]
    bases: np.ndarray      # uint8 one-hot codes, concatenated
    base_off: np.ndarray   # uint64 [n+1]
    source: np.ndarray     # reference (input order) each query was derived from

    @property
    def n(self) -> int:
        return len(self.base_off) - 1

    def seq(self, i: int) -> np.ndarray:
        return self.bases[int(self.base_off[i]):int(self.base_off[i + 1])]


def make_queries(db: SynthDB, n_queries: int, seed: int = 3, mu_q: float = 0.02, exact_frac: float = 0.10,
                 n_frac: float = 0.01, first_label: int = 0) -> SynthQueries:
    rng = np.random.default_rng(seed)
    L = db.length
    src = rng.integers(0, db.n, size=n_queries)
    refs = db.seq_bytes.reshape(db.n, L)
    bases = np.empty((n_queries, L), dtype=np.uint8)
    chunk = max(1, 4_000_000 // L)
    to_idx = np.zeros(16, dtype=np.uint8)
    to_idx[[1, 2, 4, 8]] = [0, 1, 2, 3]
    for a in range(0, n_queries, chunk):
        b = min(n_queries, a + chunk)
        cur = to_idx[refs[src[a:b]]]
        exact = rng.random(b - a) < exact_frac
        mut = _mutate(rng, cur, mu_q)
        mut[exact] = cur[exact]
        enc = ONE_HOT[mut]
        with_n = np.nonzero(rng.random(b - a) < n_frac)[0]
        for i in with_n:
            k = int(rng.integers(1, 4))
            enc[i, rng.integers(0, L, size=k)] = 15
        bases[a:b] = enc
    labels = [f"q{first_label + i}" for i in range(n_queries)]
    off = np.arange(n_queries + 1, dtype=np.uint64) * np.uint64(L)
    return SynthQueries(labels, bases.reshape(-1), off, src)


# ------------------------------------------------------------------------------------------------------------
# Real composition: the reference's example barcodes, expanded (bench.py value_real_composition, tests)
# ------------------------------------------------------------------------------------------------------------
def read_fasta_records(path) -> Tuple[List[str], List[str], List[np.ndarray]]:
    """(labels, lineages, sequences in the reference's encoding) of a FASTA file whose headers carry `tax=...;` (parser.rs:46-105:
    the header up to the first ';' is the label, `tax=([^;]+);` the lineage; ACGT only)."""
    code = np.zeros(256, np.uint8)
    for ch, v in zip("ACGT", (1, 2, 4, 8)):
        code[ord(ch)] = code[ord(ch.lower())] = v
    labels, lins, seqs, cur = [], [], [], []
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith(";"):
                continue
            if line.startswith(">"):
                if cur:
                    seqs.append(code[np.frombuffer("".join(cur).encode(), np.uint8)])
                cur = []
                labels.append(line[1:].split(";")[0])
                lins.append(line.split("tax=")[1].split(";")[0])
            else:
                cur.append(line)
    seqs.append(code[np.frombuffer("".join(cur).encode(), np.uint8)])
    assert len(labels) == len(lins) == len(seqs)
    return labels, lins, seqs


@dataclass
class HoldOut:
    lineages: List[str]        # database: one per reference
    seq_bytes: np.ndarray
    seq_off: np.ndarray
    q_bases: np.ndarray        # queries (variable length)
    q_off: np.ndarray
    q_labels: List[str]
    n_records_db: int
    n_records_held_out: int


def real_composition_holdout(path, copies: int = 16, n_queries: int = 131072, held_out: float = 0.10, seed: int = 11) -> HoldOut:
    """The reference's benchmark methodology (scripts/common.py:11-25: sample real sequences, 90 % -> database, 10 % -> queries) on the
    only real data it ships (example/diptera_queries.fasta, ~205 bp, t ~ 195), scaled to a database of some size: every database
    record enters `copies` times -- once as it is, the others with 0.2 ... 4 % substitutions drawn from the record's own bases (what
    individuals of a species look like) -- and the held-out records are the queries, each drawn with 0 / 0.5 / 1 % substitutions so
    that the batch is not 787 sequences repeated.  A query's best hit is a relative at its natural distance, never a copy of itself;
    k-mers common to most references and background counts near 40 % of the best hit are what this composition brings."""
    labels, lins, seqs = read_fasta_records(path)
    rng = np.random.default_rng(seed)
    n = len(seqs)
    perm = rng.permutation(n)
    n_q_rec = max(1, int(round(n * held_out)))
    q_rec, db_rec = np.sort(perm[:n_q_rec]), np.sort(perm[n_q_rec:])
    mus = np.array([0.002, 0.005, 0.01, 0.02, 0.04])
    out_seqs, out_lin = [], []
    for r in db_rec:
        base = seqs[r]
        for c in range(copies):
            s = base.copy()
            if c:
                hit = rng.random(len(s)) < float(mus[int(rng.integers(0, len(mus)))])
                k = int(hit.sum())
                if k:
                    s[hit] = rng.choice(base, k)
            out_seqs.append(s)
            out_lin.append(lins[r])
    off = np.zeros(len(out_seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in out_seqs])
    qs, qlab = [], []
    src = rng.integers(0, n_q_rec, size=n_queries)
    qmu = (0.0, 0.005, 0.01)
    for i in range(n_queries):
        base = seqs[q_rec[src[i]]]
        s = base.copy()
        mu = qmu[i % 3]
        if mu:
            hit = rng.random(len(s)) < mu
            k = int(hit.sum())
            if k:
                s[hit] = rng.choice(base, k)
        qs.append(s)
        qlab.append(f"{labels[q_rec[src[i]]]}#{i}")
    qoff = np.zeros(n_queries + 1, np.uint64)
    qoff[1:] = np.cumsum([len(s) for s in qs])
    return HoldOut(out_lin, np.concatenate(out_seqs), off, np.concatenate(qs), qoff, qlab, len(db_rec), n_q_rec)
