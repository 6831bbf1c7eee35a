#!/bin/bash
# config1 and the real-composition probe for an A/B of index options: tools/ab_legs.sh <two_level option value>
python - "$1" <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
import raxtax_amd as rx
from raxtax_amd import synth
from pathlib import Path
opt = int(sys.argv[1])
def leg(name, tree, bases, off, steps=5):
    index = rx.Index(tree, stage_timing=True, two_level=opt)
    index.upload(bases, off)
    index.run(0); index.download(copy=False)
    t0 = time.perf_counter()
    for _ in range(steps):
        index.run(0); index.download(copy=False)
    dt = (time.perf_counter() - t0) / steps
    st = {s: round(ms, 2) for s, (ms, n) in index.stage_times().items() if n}
    print(f"two_level={opt} {name}: {dt*1e3:.2f} ms = {(len(off)-1)/dt/1e6:.2f} M/s {st}", flush=True)
h = synth.real_composition_holdout(Path('tests/golden/diptera_queries.fasta'))
leg('real', rx.Tree.new_flat(h.lineages, h.seq_bytes, h.seq_off, kmer_map=False), h.q_bases, h.q_off)
db = synth.make_db(50_000); qs = synth.make_queries(db, 100_000, seed=3)
leg('config1', rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False), qs.bases, qs.base_off)
db = synth.make_db(200_000); qs = synth.make_queries(db, 131_072, seed=3)
leg('200k', rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False), qs.bases, qs.base_off)
PY
