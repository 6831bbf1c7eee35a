#!/usr/bin/env python3
"""Where a step of the bench spends its wall time: rtx_batch_run (ordering + enqueue), rtx_batch_download (waits for the device,
copies, host finalisation) and the device time of the stages.  Usage: tools/exp_step_timeline.py [queries] [refs]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_refs = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
db = synth.make_db(n_refs)
qs = synth.make_queries(db, n_q)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
for timing in (False, True):
    ix = rx.Index(tree, stage_timing=timing)
    ex = ix.exact_matches(qs.bases, qs.base_off)
    ix.upload(qs.bases, qs.base_off, *ex)
    for it in range(4):
        t0 = time.perf_counter()
        ix.run(0)
        t1 = time.perf_counter()
        ix.download(copy=False)
        t2 = time.perf_counter()
        st = ix.stage_times()
        print(f"stage_timing={timing} step {it}: run {1e3 * (t1 - t0):7.1f} ms, download {1e3 * (t2 - t1):7.1f} ms, total {1e3 * (t2 - t0):7.1f} ms; "
              f"device stages {sum(v[0] for v in st.values()):7.1f} ms {({k: round(v[0], 1) for k, v in st.items()})}", flush=True)
    del ix
