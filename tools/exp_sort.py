import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import raxtax_amd as rx
from raxtax_amd import synth
db = synth.make_db(50000)
qs = synth.make_queries(db, 100000)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
ix = rx.Index(tree)
orig = tree.original_index().astype(np.int64); inv = np.empty(db.n, np.int64); inv[orig] = np.arange(db.n)
L = db.length
B = qs.bases.reshape(-1, L)
for name, order in (("random", np.arange(qs.n)), ("sorted_by_true_ref", np.argsort(inv[qs.source], kind="stable")),
                    ("sorted_by_genus_noise", np.argsort(inv[qs.source] // 128 * 128 + np.random.default_rng(0).integers(0, 128, qs.n), kind="stable"))):
    bases = np.ascontiguousarray(B[order]).reshape(-1)
    ix.upload(bases, qs.base_off)
    for rep in range(2):
        ix.run(0); ix.download(copy=False)
    print(name, {k: round(v[0], 2) for k, v in ix.stage_times().items()})
