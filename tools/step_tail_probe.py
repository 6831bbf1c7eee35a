#!/usr/bin/env python3
"""How far behind the device is the host at the end of a step?  run -> sync (the device is done) -> download (what is left: copies +
finalisation of the sub-batches the host has not taken yet).   python tools/step_tail_probe.py [refs] [queries]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx
from raxtax_amd import synth
n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
db = synth.make_db(n_refs)
qs = synth.make_queries(db, n_q, seed=3)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
index = rx.Index(tree)
index.upload(qs.bases, qs.base_off)
index.run(0); index.download(copy=False)
for rep in range(4):
    t0 = time.perf_counter(); index.run(0)
    t1 = time.perf_counter(); index.sync()
    t2 = time.perf_counter(); index.download(copy=False)
    t3 = time.perf_counter()
    print(f"A (run, sync, download): enqueue {1e3*(t1-t0):.1f} ms, device until sync {1e3*(t2-t1):.1f} ms, download after sync {1e3*(t3-t2):.1f} ms, total {1e3*(t3-t0):.1f}")
for rep in range(4):
    t0 = time.perf_counter(); index.run(0)
    t1 = time.perf_counter(); index.download(copy=False)
    t3 = time.perf_counter()
    print(f"B (run, download): enqueue {1e3*(t1-t0):.1f} ms, download {1e3*(t3-t1):.1f} ms, total {1e3*(t3-t0):.1f}")
