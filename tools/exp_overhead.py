"""Where the non-kernel time of a bench step goes: enqueue, device time, D2H + host finalisation."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import raxtax_amd as rx
from raxtax_amd import synth

db = synth.make_db(50000)
qs = synth.make_queries(db, 100000)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
index = rx.Index(tree, device=0, stage_timing=("--all" in sys.argv), cluster=("--no-cluster" not in sys.argv))
SYNC = "--no-sync" not in sys.argv
ex_ids, ex_off = index.exact_matches(qs.bases, qs.base_off)
index.upload(qs.bases, qs.base_off, ex_ids, ex_off)
for it in range(6):
    t0 = time.perf_counter(); index.run(0)
    t1 = time.perf_counter()
    if SYNC: index.sync()
    t2 = time.perf_counter(); index.download(copy=False)
    t3 = time.perf_counter()
    st = index.stage_times()
    print(f"enqueue {1e3*(t1-t0):.2f}  wait {1e3*(t2-t1):.2f}  download {1e3*(t3-t2):.2f}  total {1e3*(t3-t0):.2f}  kernels {sum(v[0] for v in st.values()):.2f} {dict((k, round(v[0],2)) for k,v in st.items())}")
