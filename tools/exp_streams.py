import sys, numpy as np
sys.path.insert(0, '/root/repo')
import raxtax_amd as rx, time
from raxtax_amd import synth
db = synth.make_db(50000)
qs = synth.make_queries(db, 100000)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
for streams in (1, 2):
    ix = rx.Index(tree, streams=streams)
    ex_ids, ex_off = ix.exact_matches(qs.bases, qs.base_off)
    ix.upload(qs.bases, qs.base_off, ex_ids, ex_off)
    for rep in range(3):
        t0 = time.perf_counter(); ix.run(0); ix.sync(); t1 = time.perf_counter(); ix.download(copy=False); t2 = time.perf_counter()
    print('streams', streams, 'run+sync ms %.2f' % ((t1 - t0) * 1e3), 'download ms %.2f' % ((t2 - t1) * 1e3), {k: round(v[0], 2) for k, v in ix.stage_times().items()})
