#!/usr/bin/env python3
"""The step at a rank's CPU share: configs[2] and the real-composition leg (ten rows per query) with the library's host
pools sized for K ranks on this host (rtx_set_host_share), K = 1 / 8 / 16 -- on a 16-CPU grant 16 / 2 / 1 threads.  Per share: run + streamed
download per step, and the same with the download taken after a sync (what the host side costs when nothing hides it).
   python tools/host_share_probe.py [refs] [queries] [steps]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lib = rx._lib.load()


def measure(index, n, label):
    for share in (1, 8, 16):
        rx._lib.check(lib.rtx_set_host_share(share))
        index.run(0)
        index.download(copy=False)
        t0 = time.perf_counter()
        for _ in range(steps):
            index.run(0)
            v = index.download(copy=False)
        dt = (time.perf_counter() - t0) / steps
        index.run(0)
        index.sync()
        t1 = time.perf_counter()
        v = index.download(copy=False)
        dl = time.perf_counter() - t1
        print(f"{label}: share {share:2d} ({lib.rtx_host_threads()} threads): {dt * 1e3:7.2f} ms per step = {n / dt / 1e6:6.2f} M queries/s; "
              f"download after a sync {dl * 1e3:6.2f} ms; rows per query {v.n_rows / n:.2f}", flush=True)
    rx._lib.check(lib.rtx_set_host_share(1))
    import numpy as np
    rc = np.ctypeslib.as_array(v.row_count, shape=(n,))
    print(f"{label}: rows per query: median {int(np.median(rc))}, 90 % {int(np.percentile(rc, 90))}, 99 % {int(np.percentile(rc, 99))}, max {int(rc.max())}; "
          f"queries with 100 rows or more: {int((rc >= 100).sum())}", flush=True)


db = synth.make_db(refs)
qs = synth.make_queries(db, nq, seed=3)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
index = rx.Index(tree, device=0)
index.upload(qs.bases, qs.base_off)
measure(index, nq, f"configs[2]-like {nq} x {refs}")
del index, tree

h = synth.real_composition_holdout(ROOT / "tests" / "golden" / "diptera_queries.fasta")
tree = rx.Tree.new_flat(h.lineages, h.seq_bytes, h.seq_off, kmer_map=False)
index = rx.Index(tree, device=0)
n = len(h.q_off) - 1
index.upload(h.q_bases, h.q_off)
measure(index, n, f"real composition {n} x {len(h.lineages)}")
