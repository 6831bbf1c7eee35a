"""How much hit_count gains from the order of the queries (L2 reuse of bitmap rows): random input order,
sorted by the true source reference, the same with an XCD-aware interleave (consecutive sorted queries on the
same XCD: workgroup id % 8), and orders derived from min-hash sketches that need no knowledge of the taxonomy."""
import sys, time, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raxtax_amd as rx
from raxtax_amd import synth
db = synth.make_db(50000)
qs = synth.make_queries(db, 100000)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
ix = rx.Index(tree, stage_timing=True)
orig = tree.original_index().astype(np.int64); inv = np.empty(db.n, np.int64); inv[orig] = np.arange(db.n)
L = db.length
B = qs.bases.reshape(-1, L)
SUB = 4096

def xcd_interleave(order):
    """within every sub-batch of 4096: sorted position s -> launch position (s % 512) * 8 + s // 512"""
    out = order.copy()
    for a in range(0, len(order), SUB):
        blk = order[a:a + SUB]
        n = len(blk)
        if n < SUB:
            continue
        s = np.arange(n)
        pos = (s % 512) * 8 + s // 512
        tmp = np.empty(n, order.dtype); tmp[pos] = blk
        out[a:a + SUB] = tmp
    return out

def minhash_keys(nh):
    code = np.full(256, 255, np.uint8); code[1] = 0; code[2] = 1; code[4] = 2; code[8] = 3
    c = code[B].astype(np.uint32)                      # [Q][L]
    valid = (c < 4)
    k = np.zeros((B.shape[0], L - 7), np.uint32)
    ok = np.ones((B.shape[0], L - 7), bool)
    for j in range(8):
        k |= (c[:, j:L - 7 + j] & 3) << (14 - 2 * j)
        ok &= valid[:, j:L - 7 + j]
    keys = []
    for h in range(nh):
        mul = np.uint32([0x9E3779B1, 0x85EBCA6B, 0xC2B2AE35, 0x27D4EB2F][h])
        hv = ((k * mul) >> np.uint32(16)) & np.uint32(0xFFFF)
        hv = np.where(ok, hv, 0xFFFF)
        keys.append(hv.min(axis=1).astype(np.uint64))
    key = np.zeros(B.shape[0], np.uint64)
    for kk in keys:
        key = (key << np.uint64(16)) | kk
    return key

true_sorted = np.argsort(inv[qs.source], kind="stable")
orders = [("random", np.arange(qs.n)), ("true_sorted", true_sorted), ("true_sorted+xcd", xcd_interleave(true_sorted))]
for nh in (1, 2, 4):
    o = np.argsort(minhash_keys(nh), kind="stable")
    orders += [(f"minhash{nh}", o), (f"minhash{nh}+xcd", xcd_interleave(o))]
for name, order in orders:
    bases = np.ascontiguousarray(B[order]).reshape(-1)
    ix.upload(bases, qs.base_off)
    for rep in range(2):
        ix.run(0); ix.download(copy=False)
    print(name, {k: round(v[0], 2) for k, v in ix.stage_times().items()}, flush=True)
