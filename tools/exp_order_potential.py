"""Upper bound for the processing order: hit_count time with the library's min-hash order vs the queries sorted by
their TRUE source reference (cluster off), same kernels."""
import sys, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx
from raxtax_amd import synth
db = synth.make_db(50000)
qs = synth.make_queries(db, 100000)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
orig = tree.original_index().astype(np.int64); inv = np.empty(db.n, np.int64); inv[orig] = np.arange(db.n)
B = qs.bases.reshape(-1, db.length)
for name, cluster, order in (("library order (min-hash)", True, np.arange(qs.n)), ("input order", False, np.arange(qs.n)),
                             ("true source order", False, np.argsort(inv[qs.source], kind="stable"))):
    ix = rx.Index(tree, stage_timing=True, cluster=cluster)
    ix.upload(np.ascontiguousarray(B[order]).reshape(-1), qs.base_off)
    for rep in range(3):
        ix.run(0); ix.download(copy=False)
    print(name, {k: round(v[0], 2) for k, v in ix.stage_times().items()}, flush=True)
    del ix
