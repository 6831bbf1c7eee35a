#!/usr/bin/env python3
"""Soak of RTX_OPT_RUN_AHEAD: rtx_raxtax over the same queries again and again with chunk sizes drawn at random (the next chunk enqueued ahead, now and then the
test aid that abandons every second run-ahead): every call must print the lines of the one-chunk call.   python tools/run_ahead_soak.py [refs] [queries] [calls]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

refs = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 24
db = synth.make_db(refs)
qs = synth.make_queries(db, nq, seed=21)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
index = rx.Index(tree, device=0)
queries = [(qs.labels[i], qs.bases[qs.base_off[i]:qs.base_off[i + 1]]) for i in range(nq)]


def lines(chunk, skip):
    got = []
    rx.raxtax(queries, index, skip, False, chunk, lambda l, o, t: got.append((l, o)), False)
    return got


want = {skip: lines(0, skip) for skip in (False, True)}
rng = np.random.default_rng(5)
bad = 0
for c in range(calls):
    chunk = int(rng.integers(32_768, 140_000))
    skip = bool(c & 1)
    aid = 2 if c in (4, 9, 14) else 0   # (every abandoned run-ahead under the aid grows the arena by half, as a real overflow would: a few of them)
    rx._lib.check(index._lib.rtx_index_set_option(index._h, 23, aid))
    before = index.run_ahead_stats
    t0 = time.perf_counter()
    got = lines(chunk, skip)
    dt = time.perf_counter() - t0
    ahead, abandoned = (a - b for a, b in zip(index.run_ahead_stats, before))
    ok = got == want[skip]
    bad += not ok
    print(f"call {c:2d}: chunk {chunk:6d} skip {int(skip)} aid {aid}: {dt * 1e3:7.1f} ms, enqueued ahead {ahead}, abandoned {abandoned}, lines {'identical' if ok else 'DIFFER'}", flush=True)
rx._lib.check(index._lib.rtx_index_set_option(index._h, 23, 0))
print("soak", "ok" if not bad else f"FAILED ({bad} calls)")
sys.exit(1 if bad else 0)
