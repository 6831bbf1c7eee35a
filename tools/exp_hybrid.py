import sys, numpy as np
sys.path.insert(0, '/root/repo')
import raxtax_amd as rx
from raxtax_amd import synth
db = synth.make_db(50000)
qs = synth.make_queries(db, 40000)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
for name, kw in (("plain", dict(hybrid=False)), ("sm0", dict(hybrid=True, sparse_max=0)), ("sm4", dict(hybrid=True, sparse_max=4)),
                 ("sm12", dict(hybrid=True, sparse_max=12)), ("sm32", dict(hybrid=True, sparse_max=32))):
    ix = rx.Index(tree, **kw)
    ix.upload(qs.bases, qs.base_off)
    for rep in range(2):
        ix.run(0); ix.download(copy=False)
    print(name, 'index MB %.0f' % (ix.device_bytes / 1e6), {k: round(v[0], 2) for k, v in ix.stage_times().items() if k in ('hit_count',)})
    del ix
