#!/usr/bin/env python3
"""The long-read leg of bench.py on its own (16S-like reads of 1 500 bases against references of 1 500 bases): step and stage times.
   python tools/long_reads_probe.py [refs] [queries] [length]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
db = synth.make_db(n_refs, length=L)
qs = synth.make_queries(db, n_q, seed=5)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
index = rx.Index(tree, stage_timing=True)
index.upload(qs.bases, qs.base_off)
index.run(0); index.download(copy=False)
t0 = time.perf_counter()
for _ in range(2):
    index.run(0)
    index.download(copy=False)
dt = (time.perf_counter() - t0) / 2
st = {s: round(ms, 2) for s, (ms, n) in index.stage_times().items() if n}
print(f"{n_q} reads of {L} bases vs {n_refs} references: {dt * 1e3:.1f} ms per step = {n_q / dt / 1e6:.3f} M/s; stages {st}")
