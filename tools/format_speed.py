import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, '.')
import raxtax_amd as rx
from raxtax_amd import dist_util, synth
n_q = 1_000_000
rng = np.random.default_rng(0)
count = np.where(rng.random(n_q) < 0.15, 2, 1).astype(np.uint32)
n_rows = int(count.sum())
order = rng.permutation(n_q)   # rows in a processing order that is not the query order
begin = np.zeros(n_q, np.int64); begin[order] = np.concatenate([[0], np.cumsum(count[order])[:-1]])
rec = dist_util.pack_records(None, rng.integers(0, 500_000, n_rows), np.full(n_rows, 6), rng.integers(0, 101, (n_rows, 32)) / 100.0,
                             rng.random(n_rows), global_signal=rng.random(n_q), row_begin=begin, row_count=count, t=np.full(n_q, 640), status=np.zeros(n_q))
db = synth.make_db(500_000)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
labels = (ctypes.c_char_p * n_q)(*[f"q{i}".encode() for i in range(n_q)])
keep = np.empty(4 * len(rec) + (1 << 20), np.uint8)   # the writer's buffer, touched once
for nt in (1, 4, 8, 16):
    for reuse in (False, True):
        dist_util.format_records(tree, rec, labels, threads=nt, out=keep if reuse else None)
        t0 = time.perf_counter()
        for _ in range(3):
            text, off = dist_util.format_records(tree, rec, labels, threads=nt, out=keep if reuse else None)
        dt = (time.perf_counter() - t0) / 3
        print(f"{nt} threads, {'the same output buffer every call' if reuse else 'a fresh output array per call'}: {1e3*dt:.0f} ms per 1 M queries "
              f"({len(text)/1e6:.0f} MB), {1e9*dt*nt/n_q:.0f} ns of one thread per query", flush=True)
