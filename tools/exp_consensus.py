#!/usr/bin/env python3
"""Host-side study (no GPU): how much smaller would the index be if hit counts were taken against one consensus column per
species plus per-reference corrections (count_r = |K(q) & C_s| + |K(q) & Add_r| - |K(q) & Rem_r|)?  Reports, for a
sample of species and queries, the postings of the plain index vs consensus + corrections, and the correction updates a
query would trigger.  Usage: tools/exp_consensus.py [refs] [species sample]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.oracle_py import Oracle  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
n_sp = int(sys.argv[2]) if len(sys.argv) > 2 else 120
db = synth.make_db(n_refs)
o = Oracle()
lin = np.array(db.lineages)
order = np.argsort(lin, kind="stable")
# species = runs of equal lineage in sorted order
sl = lin[order]
starts = np.flatnonzero(np.concatenate([[True], sl[1:] != sl[:-1]]))
ends = np.concatenate([starts[1:], [n_refs]])
rng = np.random.default_rng(0)
pick = rng.choice(len(starts), n_sp, replace=False)
plain = cons = add = rem = 0
add_by_k = np.zeros(65536, np.int64)
rem_by_k = np.zeros(65536, np.int64)
cons_by_k = np.zeros(65536, np.int64)
plain_by_k = np.zeros(65536, np.int64)
nref_s = 0
for s in pick:
    members = order[starts[s]:ends[s]]
    sets = [o.sequence_to_kmers(db.seq(int(r))).astype(np.int64) for r in members]
    cnt = np.zeros(65536, np.int32)
    for k in sets:
        cnt[k] += 1
    c = cnt * 2 > len(members)
    cons += int(c.sum())
    cons_by_k[c] += 1
    for k in sets:
        plain += len(k)
        plain_by_k[k] += 1
        has = np.zeros(65536, bool)
        has[k] = True
        a = has & ~c
        r = c & ~has
        add += int(a.sum())
        rem += int(r.sum())
        add_by_k[a] += 1
        rem_by_k[r] += 1
    nref_s += len(members)
print(f"{n_sp} species, {nref_s} references: plain postings {plain / nref_s:.0f} per reference; consensus {cons / n_sp:.0f} per species "
      f"({cons / nref_s:.1f} per reference) + corrections add {add / nref_s:.1f} + remove {rem / nref_s:.1f} per reference")
# a query derived from a reference of the sample: what it touches (scaled to the whole database by nref_s / n_refs)
qs = synth.make_queries(db, 2000)
scale = n_refs / nref_s
tot_plain = tot_cons = tot_add = tot_rem = 0
for q in range(200):
    km = o.sequence_to_kmers(qs.seq(q)).astype(np.int64)
    tot_plain += plain_by_k[km].sum()
    tot_cons += cons_by_k[km].sum()
    tot_add += add_by_k[km].sum()
    tot_rem += rem_by_k[km].sum()
print(f"per query, scaled to {n_refs} references: plain postings touched {tot_plain / 200 * scale / 1e6:.2f} M (H_q); consensus postings "
      f"{tot_cons / 200 * scale / 1e6:.3f} M; correction updates add {tot_add / 200 * scale / 1e6:.3f} M + remove {tot_rem / 200 * scale / 1e6:.3f} M")
