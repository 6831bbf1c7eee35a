#!/usr/bin/env python3
"""The real-composition leg of bench.py on its own (hold-out of the Diptera records, 14 tiles): step time, stage times, rows per query,
host share.  For rocprofv3 --kernel-trace --stats.   python tools/real_composition_probe.py [steps] [RTX_OPT_RECORDS]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h = synth.real_composition_holdout(ROOT / "tests" / "golden" / "diptera_queries.fasta")
tree = rx.Tree.new_flat(h.lineages, h.seq_bytes, h.seq_off, kmer_map=False)
records = int(sys.argv[2]) if len(sys.argv) > 2 else None   # RTX_OPT_RECORDS: live tiles up to which a pruned query takes the records path
index = rx.Index(tree, stage_timing=True, records=records, prune_self_sample=False)   # (prune=1 below means pruning ON: the verdict of the self-sample is set aside)
n_q = len(h.q_off) - 1
index.upload(h.q_bases, h.q_off)
for prune in (1, 0):
    rx._lib.check(index._lib.rtx_index_set_option(index._h, 13, prune))
    index.upload(h.q_bases, h.q_off)
    index.run(0); index.download(copy=False)
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        index.run(0)
        t2 = time.perf_counter()
        index.sync()
        t3 = time.perf_counter()
        v = index.download(copy=False)
        t4 = time.perf_counter()
    dt = (time.perf_counter() - t0) / steps
    st = {s: round(ms, 2) for s, (ms, n) in index.stage_times().items() if n}
    print(f"prune={prune}: {dt * 1e3:.1f} ms per step of {n_q} queries = {n_q / dt / 1e6:.2f} M/s; last step: enqueue {1e3 * (t2 - t1):.1f} ms, "
          f"device until sync {1e3 * (t3 - t2):.1f} ms, download after sync {1e3 * (t4 - t3):.1f} ms; rows per query {v.n_rows / n_q:.2f}; "
          f"stages sum {sum(st.values()):.1f} ms {st}; prune stats {index.debug_prune_stats() if prune else ''}")
