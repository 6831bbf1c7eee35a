#!/usr/bin/env python3
"""The real-composition leg under a few option settings (sub-batches, overlap): ms per step with and without tile pruning.
   python tools/real_composition_options.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

h = synth.real_composition_holdout(ROOT / "tests" / "golden" / "diptera_queries.fasta")
tree = rx.Tree.new_flat(h.lineages, h.seq_bytes, h.seq_off, kmer_map=False)
n_q = len(h.q_off) - 1
for name, opts in (("default", {}), ("min_sub_batches=2", {20: 2}), ("min_sub_batches=1", {20: 1}), ("min_sub_batches=8", {20: 8}), ("overlap=0", {19: 0}),
                   ("sub_batch=65536", {1: 65536}), ("sub_batch=16384", {1: 16384})):
    index = rx.Index(tree, prune_self_sample=False)
    for k, v in opts.items():
        rx._lib.check(index._lib.rtx_index_set_option(index._h, k, v))
    out = []
    for prune in (1, 0):
        rx._lib.check(index._lib.rtx_index_set_option(index._h, 13, prune))
        index.upload(h.q_bases, h.q_off)
        for _ in range(2):
            index.run(0); index.download(copy=False)
        t0 = time.perf_counter()
        for _ in range(4):
            index.run(0)
            index.download(copy=False)
        out.append((time.perf_counter() - t0) / 4 * 1e3)
    print(f"{name:22s} pruned {out[0]:6.2f} ms  unpruned {out[1]:6.2f} ms per {n_q} queries", flush=True)
    del index
