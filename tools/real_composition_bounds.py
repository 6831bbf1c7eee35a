import sys
import numpy as np
sys.path.insert(0, '.')
from pathlib import Path
import raxtax_amd as rx
from raxtax_amd import synth, checks
h = synth.real_composition_holdout(Path('tests/golden/diptera_queries.fasta'))
tree = rx.Tree.new_flat(h.lineages, h.seq_bytes, h.seq_off, kmer_map=False)
index = rx.Index(tree, debug_taps=True, prune_self_sample=False)
n_q = len(h.q_off) - 1
index.upload(h.q_bases, h.q_off); index.run(0); res = index.download()
qs = checks.last_sub_batch_queries(index, n_q)[:300]
fr_h, fr_u, fr_u2 = [], [], []
for q in qs:
    ub = index.debug_tile_bounds(int(q)).astype(int); d = index.debug_prune_detail(int(q))
    hm = d["block_counts"]; M = d["M"]; H = hm[hm * 5 >= M * 4]; hmin = int(H.min()) if len(H) else 0
    fr_h.append((ub > hmin - 1).mean()); fr_u.append((ub > d["threshold"]).mean())
print("tiles above h_min - 1: mean fraction %.2f; above the final threshold: %.2f; M %.0f thr %.0f" % (np.mean(fr_h), np.mean(fr_u), np.mean([index.debug_prune_detail(int(q))["M"] for q in qs[:50]]), np.mean([index.debug_prune_detail(int(q))["threshold"] for q in qs[:50]])))
print("share of queries with >= 75 %% of tiles above h_min - 1: %.2f; with >= 75 %% above the final threshold: %.2f" % (np.mean(np.array(fr_h) >= 0.75), np.mean(np.array(fr_u) >= 0.75)))
