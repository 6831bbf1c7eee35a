#!/usr/bin/env python3
"""End to end through rtx_raxtax / rtx_raxtax_multi (host buffers -> formatted strings, a sender that discards) at configs[2]:
one handle against two handles ON THE SAME GPU (a second index; chunk c + 1 is enqueued on the other handle while chunk c is
finalised), several chunk sizes.   python tools/e2e_probe.py [queries] [refs]"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_q = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_refs = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
db = synth.make_db(n_refs)
qs = synth.make_queries(db, n_q, seed=3)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
lib = rx._lib.load()
handles = [rx.Index(tree, device=0), rx.Index(tree, device=0)]
labels = (ctypes.c_char_p * n_q)(*[l.encode() for l in qs.labels])
SENDER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p)
lib.rtx_raxtax_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_char_p),
                                 rx._lib.u8p, rx._lib.u64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, SENDER, ctypes.c_void_p, ctypes.c_int]
discard = ctypes.cast(lib.rtx_sender_discard, SENDER)
bases, off = np.ascontiguousarray(qs.bases), np.ascontiguousarray(qs.base_off)
for n_h in (1, 2):
    arr = (ctypes.c_void_p * n_h)(*[h._h for h in handles[:n_h]])
    for chunk in (65536, 131072, 262144):
        counted = (ctypes.c_uint64 * 2)()

        def run():
            rx._lib.check(lib.rtx_raxtax_multi(arr, n_h, tree._h, n_q, labels, rx._lib.ptr(bases, rx._lib.u8p), rx._lib.ptr(off, rx._lib.u64p), 0, 0, chunk,
                                               discard, ctypes.cast(counted, ctypes.c_void_p), 0))
        run()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        dt = (time.perf_counter() - t0) / 3
        busy = (ctypes.c_double * 4)()
        nch = ctypes.c_uint64()
        lib.rtx_raxtax_last_timing(busy, ctypes.byref(nch))
        print(f"{n_h} handle(s), chunks of {chunk}: {dt * 1e3:.1f} ms per {n_q} queries = {n_q / dt / 1e6:.2f} M/s; busy ms: device {busy[1] * 1e3:.0f} (busiest handle), "
              f"format {busy[2] * 1e3:.0f}, sender {busy[3] * 1e3:.0f}", flush=True)
