#!/usr/bin/env python3
"""Does tile pruning pay on databases whose tiles are (nearly) all live?  Step time with pruning forced on and off, and the handle's own verdict
(rtx_index_self_sample: off from a live share of 0.85), on the real-composition hold-out (14 tiles, share 0.945) and on 30 000 near-identical
references (4 tiles, share 1.0; tests/test_gpu_records.py).   python tools/prune_rule_probe.py"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402


def measure(name, tree, bases, off, steps=4):
    auto = rx.Index(tree)
    print(f"{name}: verdict {auto.prune_verdict}", flush=True)
    del auto
    index = rx.Index(tree, prune_self_sample=False)
    n = len(off) - 1
    for prune in (1, 0, 1, 0):
        rx._lib.check(index._lib.rtx_index_set_option(index._h, 13, prune))
        index.upload(bases, off)
        index.run(0); index.download(copy=False)
        t0 = time.perf_counter()
        for _ in range(steps):
            index.run(0)
            index.download(copy=False)
        dt = (time.perf_counter() - t0) / steps
        st = index.debug_prune_stats() if prune else {}
        print(f"{name}: pruning {prune}: {dt * 1e3:7.2f} ms per {n} queries = {n / dt / 1e6:.2f} M/s; live tiles per query {st.get('live_tiles_per_query', float('nan')):.2f}, "
              f"with threshold {st.get('queries_with_threshold', 0)}, on the records path {st.get('record_queries', 0)}", flush=True)


h = synth.real_composition_holdout(ROOT / "tests" / "golden" / "diptera_queries.fasta")
measure("real composition", rx.Tree.new_flat(h.lineages, h.seq_bytes, h.seq_off, kmer_map=False), h.q_bases, h.q_off)

L, n_refs, n_q = 400, 30_000, 65_536
rng = np.random.default_rng(3)
root = synth._draw(rng, (1, L))
seqs = synth._mutate(rng, np.repeat(root, n_refs, axis=0), 0.01)
seq_bytes = synth.ONE_HOT[seqs].reshape(-1)
seq_off = np.arange(n_refs + 1, dtype=np.uint64) * np.uint64(L)
lineages = [f"p:P{i % 3},c:C{i % 30},o:O{i % 300},f:F{i % 3000},g:G{i},s:S{i}" for i in range(n_refs)]
q = synth.ONE_HOT[synth._mutate(rng, seqs[rng.integers(0, n_refs, n_q)], 0.01)].reshape(-1)
qoff = np.arange(n_q + 1, dtype=np.uint64) * np.uint64(L)
measure("near-identical 30k", rx.Tree.new_flat(lineages, seq_bytes, seq_off, kmer_map=False), q, qoff)
