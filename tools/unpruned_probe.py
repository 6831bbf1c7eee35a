#!/usr/bin/env python3
"""Stage times of a run that counts every tile (RTX_OPT_TILE_PRUNE = 0) on the synthetic workload; results are not looked at (for variant
builds that change them).   python tools/unpruned_probe.py [refs] [queries] [steps]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
db = synth.make_db(n_refs)
qs = synth.make_queries(db, n_q)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
index = rx.Index(tree, stage_timing=True)
rx._lib.check(index._lib.rtx_index_set_option(index._h, 13, 0))
index.upload(qs.bases, qs.base_off)
index.run(0); index.download(copy=False)
t0 = time.perf_counter()
for _ in range(steps):
    index.run(0)
    index.sync()
    index.download(copy=False)
dt = (time.perf_counter() - t0) / steps
st = {s: round(ms, 2) for s, (ms, n) in index.stage_times().items() if n}
print(f"unpruned, {n_refs} references: {dt * 1e3:.1f} ms per step of {n_q} queries = {n_q / dt / 1e6:.3f} M/s; stages {st}")
