import sys, numpy as np
sys.path.insert(0, '/root/repo')
import raxtax_amd as rx
from raxtax_amd import synth
db = synth.make_db(50000)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
ix = rx.Index(tree, stage_timing=True)
qs = synth.make_queries(db, 40000)
L = db.length
for keep in (658, 330, 170, 90, 16):
    b = qs.bases.reshape(-1, L).copy()
    b[:, keep:] = 15          # N: only the first `keep` bases give k-mers
    ix.upload(b.reshape(-1), qs.base_off)
    for rep in range(2):
        ix.run(0); ix.download(copy=False)
    st = ix.stage_times()
    print('valid bases', keep, {k: round(v[0], 2) for k, v in st.items()})
