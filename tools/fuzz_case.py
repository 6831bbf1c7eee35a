"""Prints one query of a tests/test_gpu_parity.py::test_randomised_configurations case: oracle rows vs device rows.
Usage: python tools/fuzz_case.py <seed> <query> [skip]   (needs the GPU; imports the test module for its generators)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import raxtax_amd as rx
from raxtax_amd import synth
import test_gpu_parity as T

seed, qsel = int(sys.argv[1]), int(sys.argv[2])
skip = len(sys.argv) > 3 and sys.argv[3] == "skip"
from oracle.oracle_py import Oracle
oracle = Oracle()
rng = np.random.default_rng(seed)
n_refs = int(rng.choice([37, 700, 8192, 9000, 17000, 26000]))
L = int(rng.choice([40, 150, 658]))
phylo = bool(rng.random() < 0.5)
if phylo:
    db = synth.make_db(n_refs, length=L)
    lineages, flat, off = db.lineages, db.seq_bytes, db.seq_off
else:
    lineages, flat, off = T._random_db(n_refs, L, seed + 1, n_taxa=int(rng.choice([3, 50, 900])))
seqs = flat.reshape(n_refs, L)
otree = oracle.tree_new_flat(lineages, flat, off)
tree = rx.Tree.new_flat(lineages, flat, off, kmer_map=bool(rng.random() < 0.5))
sub = int(rng.choice([0, 5, 64])); cl = bool(rng.random() < 0.7)
print(f"n_refs {n_refs} L {L} phylo {phylo} sub_batch {sub} cluster {cl}")
ix = rx.Index(tree, sub_batch=sub, cluster=cl)
qs = []
for i in range(40):
    src = seqs[int(rng.integers(0, n_refs))].copy()
    kind = rng.integers(0, 5)
    if kind == 0: q = src
    elif kind == 1:
        q = src.copy(); pos = rng.integers(0, L, max(1, L // 40)); q[pos] = (1 << rng.integers(0, 4, len(pos))).astype(np.uint8)
    elif kind == 2: q = src[: int(rng.integers(8, L + 1))].copy()
    elif kind == 3:
        q = src.copy(); q[rng.integers(0, L, 3)] = np.uint8(rng.choice([3, 5, 9, 15]))
    else: q = np.concatenate([src, seqs[int(rng.integers(0, n_refs))][: L // 2]])
    qs.append(q)
qoff = np.zeros(len(qs) + 1, np.uint64); qoff[1:] = np.cumsum([len(q) for q in qs])
bases = np.concatenate(qs)
ex_ids, ex_off = ix.exact_matches(bases, qoff)
res = ix.classify(bases, qoff, ex_ids, ex_off, skip_exact_matches=skip)
q = qsel
t, counts = otree.hit_counts(qs[q], skip_exact=skip)
rows, _ = T._oracle_rows(otree, qs[q], skip)
probs = oracle.highest_hit_prob_per_reference(t, t // 2, counts)
print("t", t, "len", len(qs[q]), "max count", counts.max(), "n at max", int((counts == counts.max()).sum()))
print("oracle rows:")
for r in rows: print("  ", r["idx"], lineages[r["idx"]], r["conf"])
print("device rows:")
for g in res.rows(q): print("  ", g.lineage, lineages[g.lineage], g.confidence_values)
on = otree.nodes()
pre = np.concatenate([[0.0], np.cumsum(probs)])
conf = pre[on["hi"].astype(np.int64)] - pre[on["lo"].astype(np.int64)]
frac = conf * 100 - np.floor(conf * 100)
near = np.where(np.abs(frac - 0.5) < 1e-6)[0]
print("nodes at a rounding boundary:", [(int(i), float(conf[i])) for i in near[:10]])
