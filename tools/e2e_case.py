import ctypes, sys, time
import numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx
from raxtax_amd import synth
n_q = int(sys.argv[1]); chunk = int(sys.argv[2]); n_h = int(sys.argv[3]); opts = sys.argv[4:]
db = synth.make_db(500_000)
qs = synth.make_queries(db, n_q, seed=3)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
lib = rx._lib.load()
kw = {}
for o in opts:
    k, v = o.split('='); kw[k] = int(v)
handles = [rx.Index(tree, device=0, **kw) for _ in range(n_h)]
labels = (ctypes.c_char_p * n_q)(*[l.encode() for l in qs.labels])
SENDER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p)
lib.rtx_raxtax_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_char_p),
                                 rx._lib.u8p, rx._lib.u64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, SENDER, ctypes.c_void_p, ctypes.c_int]
discard = ctypes.cast(lib.rtx_sender_discard, SENDER)
bases, off = np.ascontiguousarray(qs.bases), np.ascontiguousarray(qs.base_off)
arr = (ctypes.c_void_p * n_h)(*[h._h for h in handles])
counted = (ctypes.c_uint64 * 2)()
for rep in range(4):
    t0 = time.perf_counter()
    rx._lib.check(lib.rtx_raxtax_multi(arr, n_h, tree._h, n_q, labels, rx._lib.ptr(bases, rx._lib.u8p), rx._lib.ptr(off, rx._lib.u64p), 0, 0, chunk, discard, ctypes.cast(counted, ctypes.c_void_p), 0))
    dt = time.perf_counter() - t0
    busy = (ctypes.c_double * 4)(); nch = ctypes.c_uint64()
    lib.rtx_raxtax_last_timing(busy, ctypes.byref(nch))
    info = []
    for h in handles:
        sb, ns = ctypes.c_uint32(), ctypes.c_uint32()
        lib.rtx_batch_sub_batch(h._h, ctypes.byref(sb), ctypes.byref(ns))
        info.append((sb.value, ns.value, round(h.device_bytes / 1e9, 2)))
    print("   (sub-batch, sub-batches, index GB) per handle:", info)
    print("   run-ahead (enqueued ahead, abandoned) per handle:", [h.run_ahead_stats for h in handles])
    print(f"n_q={n_q} chunk={chunk} handles={n_h} {opts} rep {rep}: {dt*1e3:.1f} ms; device {busy[1]*1e3:.0f} format {busy[2]*1e3:.0f} sender {busy[3]*1e3:.0f}", flush=True)
