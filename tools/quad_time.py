#!/usr/bin/env python3
"""Times hit_count of one library build (RTX_LIB_PATH) at BASELINE configs[2] size.  Usage: quad_time.py quad|single [queries]"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

quad = sys.argv[1] == "quad"
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 40960
db = synth.make_db(500_000)
qs = synth.make_queries(db, n_q)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
import os
ix = rx.Index(tree, hit_quad=quad, stage_timing=not os.environ.get("RTX_EXP_COUNT_ONLY"), segment_classes=int(os.environ.get("RTX_SEG_CLASSES", "1")),
              streams=int(os.environ.get("RTX_STREAMS", "0")), hit_pair=int(os.environ.get("RTX_HIT_PAIR", "0")))
ex = ix.exact_matches(qs.bases, qs.base_off)
ix.upload(qs.bases, qs.base_off, *ex)
for _ in range(3):
    t0 = time.time()
    ix.run(0)
    if os.environ.get("RTX_EXP_COUNT_ONLY"):     # builds with wrong counts on purpose: kmer_extract + hit_count only
        ix.sync()
    else:
        ix.download(copy=False)
    dt = time.time() - t0
st = ix.stage_times()
w = ix.work()["bitmap_bytes_read"]
row_bytes = ((500_000 + 7) // 8 + 61) // 62
gsz = 2 if os.environ.get("RTX_HIT_PAIR", "0") != "0" else 4
print(f"  [stamp builds: {w / row_bytes * 64 / ((n_q + gsz - 1) // gsz * 62):.0f} cycles per (group, tile) of the stamped phase]")
print(f"{sys.argv[1]} {n_q} queries: {dt * 1e3:.1f} ms, hit_count {st['hit_count'][0]:.1f} ms over {st['hit_count'][1]} launches; "
      f"requested MB/query {ix.work()['bitmap_bytes_read'] / n_q / 1e6:.2f}", flush=True)
print(f"  [kmer stamp builds: {ix.work()['sum_hits'] / n_q:.0f} cycles per query of the stamped phase; kmer_extract {st.get('kmer_extract', (0, 0))[0]:.2f} ms]")
