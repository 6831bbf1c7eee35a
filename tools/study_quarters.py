#!/usr/bin/env python3
"""Host-side study (no GPU) for the two-level bounds pass (bounds2_kernel): which B-tiles (4 tiles of the database) does a refine rule pick, and how many live tiles are left.
Derived from study_toplevel.py: how loose do union bounds over LARGER blocks of references get?  For sampled queries at several divergences:
the threshold of the pruning (criterion (3) from the best block of 64), the union bound per block of B references for B in
64 .. 2048, the live tiles each B leaves, and the popcount classes of the coarse (blocks of 64) bitmap rows a query reads.
    python tools/study_toplevel.py [n_refs] [queries per divergence]"""
import ctypes as C
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import raxtax_amd as rx  # noqa: E402
from oracle.oracle_py import Oracle  # noqa: E402
from raxtax_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 12
BS = (64, 256)
DELTAS = (0.36,)


def emul_lib():
    out = ROOT / "tests" / "_build" / "librtx_emul.so"
    src = ROOT / "raxtax_amd" / "csrc" / "rtx_emul.cpp"
    out.parent.mkdir(exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", f"-I{src.parent}", "-o", str(out), str(src)])
    return C.CDLL(str(out))


def main():
    t0 = time.time()
    db = synth.make_db(N)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    off, post = tree.csr()
    orig = tree.original_index().astype(np.int64)
    print(f"db + tree: {time.time() - t0:.1f} s; postings {len(post)}", flush=True)
    orc = Oracle()
    emul = emul_lib()
    lf = np.array([orc.lib.orc_ln_factorial(i) for i in range(2 * 660 + 8)], dtype=np.float64)
    ntiles = (N + 8191) // 8192
    seqs_sorted = db.seq_bytes.reshape(N, db.length)[orig]
    rng = np.random.default_rng(1)
    for mu in (0.02, 0.05):
        rows = []
        for qi in range(NQ):
            src = int(rng.integers(0, N))
            s = seqs_sorted[src].copy()
            hit = rng.random(len(s)) < mu
            s[hit] = (1 << rng.integers(0, 4, int(hit.sum()))).astype(np.uint8)
            km = orc.sequence_to_kmers(s)
            t = len(km)
            M = np.zeros((N, t), dtype=bool)
            for j, k in enumerate(km):
                M[post[off[k]:off[k + 1]], j] = True
            counts = M.sum(axis=1)

            def ub_fixed(sz):
                pad = (-N) % sz
                X = np.concatenate([M, np.zeros((pad, t), bool)]) if pad else M
                return X.reshape(-1, sz, t).any(axis=1).sum(axis=1)
            ub = {b: ub_fixed(b) for b in BS}
            best = int(np.argmax(ub[64]))
            blk = np.zeros(64, np.uint32)
            seg = counts[best * 64:(best + 1) * 64]
            blk[:len(seg)] = seg
            r = dict(t=t, M=int(counts.max()))
            tmax = np.concatenate([counts, np.zeros((-N) % 8192, counts.dtype)]).reshape(-1, 8192).max(axis=1)
            def thr_of(tub):
                tub = np.ascontiguousarray(tub.astype(np.uint16))
                u4, i14 = C.c_uint32(), C.c_uint32()
                emul.emul_prune_threshold_tiles(C.c_uint32(t), C.c_uint64(N), blk.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p), C.c_uint32(1023),
                                                C.c_uint32(len(tub)), tub.ctypes.data_as(C.c_void_p), C.byref(u4), C.byref(i14))
                return int(u4.value)
            def per_tile_max(v, b):
                per_tile = 8192 // b
                return np.concatenate([v, np.zeros((-len(v)) % per_tile, v.dtype)]).reshape(-1, per_tile).max(axis=1)
            x64, xA = per_tile_max(ub[64], 64), per_tile_max(ub[256], 256)
            r["thr64"] = thr_of(x64)
            r["live64"] = int((x64 > r["thr64"]).sum())
            r["maxA"] = int(xA.max())
            r["max64"] = int(x64.max())
            nbt = (len(xA) + 3) // 4
            btA = np.concatenate([xA, np.zeros(nbt * 4 - len(xA), xA.dtype)]).reshape(nbt, 4).max(axis=1)
            for d in DELTAS:
                ref_bt = btA + int(d * t) > xA.max()
                mixed = np.where(np.repeat(ref_bt, 4)[:len(xA)], x64, xA)
                th = thr_of(mixed)
                r[f"d{d}:nbt"] = int(ref_bt.sum())
                r[f"d{d}:thr"] = th
                r[f"d{d}:live"] = int((mixed > th).sum())
            r["need"] = int((tmax > r["thr64"]).sum())
            # quarters of a tile (2048 references = 32 blocks of 64) and sub-tiles (512 = 8 blocks) with a block bound above the threshold, and
            # with a COUNT above it
            u = r["thr64"]
            qb = per_tile_max(ub[64], 64 * 1)  # placeholder, replaced below
            b64 = np.concatenate([ub[64], np.zeros((-len(ub[64])) % 128, ub[64].dtype)])
            r["quarters_bound"] = int((b64.reshape(-1, 32).max(axis=1) > u).sum())
            r["subtiles_bound"] = int((b64.reshape(-1, 8).max(axis=1) > u).sum())
            r["blocks_bound"] = int((b64 > u).sum())
            cpad = np.concatenate([counts, np.zeros((-N) % 8192, counts.dtype)])
            r["quarters_count"] = int((cpad.reshape(-1, 2048).max(axis=1) > u).sum())
            r["refs_above"] = int((counts > u).sum())
            rows.append(r)
        keys = [k for k in rows[0]]
        print(f"mu_q = {mu}: means over {NQ} queries ({ntiles} tiles)")
        print("   " + "  ".join(f"{k}={np.mean([r[k] for r in rows]):.1f}" for k in keys), flush=True)
    print(f"total {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
