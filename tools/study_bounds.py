#!/usr/bin/env python3
"""Host-side study of the tile pruning's bounds (no GPU): for sampled queries at several divergences, the exact hit counts, the union
bounds over blocks of references of several shapes, the threshold, and how many tiles of 8192 references stay live under each choice.
    python tools/study_bounds.py [n_refs] [queries per divergence]
Blocks: fixed runs of 64 / 32 / 16 references (what rtx_prune.hip does with 64), and runs cut at species boundaries (at most 64 long)."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import raxtax_amd as rx  # noqa: E402
from oracle.oracle_py import Oracle  # noqa: E402
from raxtax_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 24


def emul_lib():
    import subprocess
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
    lin = tree.lineages
    orig = tree.original_index().astype(np.int64)
    print(f"db + tree: {time.time() - t0:.1f} s; postings {len(post)}")
    orc = Oracle()
    emul = emul_lib()
    lf = np.array([orc.lib.orc_ln_factorial(i) for i in range(2 * 660 + 8)], dtype=np.float64)
    species_start = np.array([i for i in range(N) if i == 0 or lin[i] != lin[i - 1]], dtype=np.int64)
    # blocks cut at species boundaries, at most 64 long
    cuts = []
    for a, b in zip(species_start, np.append(species_start[1:], N)):
        cuts.extend(range(a, b, 64))
    cuts = np.array(sorted(set(cuts) | set(range(0, N, 8192))), dtype=np.int64)   # never across a tile
    print(f"{len(species_start)} species, {len(cuts)} species-aligned blocks (fixed 64: {(N + 63) // 64})")
    # data-driven blocks, no taxonomy: a block is closed when it has 56 references, at a tile boundary, or when the next reference
    # would bring more than `grow` k-mers the block does not have yet (its union would widen: the bound of every member loosens)
    seqs_sorted0 = db.seq_bytes.reshape(N, db.length)[orig]
    code = np.zeros(16, np.int64); code[[1, 2, 4, 8]] = [0, 1, 2, 3]
    c2 = code[seqs_sorted0]
    km_all = np.zeros((N, db.length - 7), np.int64)
    for j in range(8):
        km_all = km_all * 4 + c2[:, j:db.length - 7 + j]
    km_all = km_all.astype(np.uint16)
    def greedy(grow, cap=56):
        starts = [0]
        have = np.zeros(65536, bool)
        have[km_all[0]] = True
        n_in = 1
        for r in range(1, N):
            k = km_all[r]
            new = int((~have[k]).sum())        # (duplicates inside a reference count twice: a slight over-estimate)
            if n_in >= cap or r % 8192 == 0 or new > grow:
                starts.append(r)
                have[:] = False
                n_in = 0
            have[k] = True
            n_in += 1
        return np.array(starts, dtype=np.int64)
    greedy_cuts = {}
    for grow in (40, 80, 160):
        tg = time.time()
        greedy_cuts[grow] = greedy(grow)
        print(f"greedy blocks, at most {grow} new k-mers per added reference: {len(greedy_cuts[grow])} blocks ({time.time() - tg:.0f} s)")
    ntiles = (N + 8191) // 8192
    seqs_sorted = db.seq_bytes.reshape(N, db.length)[orig]
    rng = np.random.default_rng(1)
    for mu in (0.02, 0.05, 0.10, 0.15):
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
            ub64, ub32, ub16, ub8 = ub_fixed(64), ub_fixed(32), ub_fixed(16), ub_fixed(8)
            ubsp = np.logical_or.reduceat(M, cuts, axis=0).sum(axis=1)
            ubg = {g: np.logical_or.reduceat(M, c, axis=0).sum(axis=1) for g, c in greedy_cuts.items()}
            best = int(np.argmax(ub64))
            blk = np.zeros(64, np.uint32)
            seg = counts[best * 64:(best + 1) * 64]
            blk[:len(seg)] = seg
            u, i1 = C.c_uint32(), C.c_uint32()
            emul.emul_prune_threshold(C.c_uint32(t), C.c_uint64(N), blk.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p), C.c_uint32(1023), C.byref(u), C.byref(i1))
            thr = int(u.value)
            # round 4: the tile-aware threshold (criterion (4)) from the largest bound of every tile of the blocks of 64
            pad64 = (-len(ub64)) % 128
            tub = np.concatenate([ub64, np.zeros(pad64, ub64.dtype)]).reshape(-1, 128).max(axis=1).astype(np.uint16)
            u4, i14 = C.c_uint32(), C.c_uint32()
            emul.emul_prune_threshold_tiles(C.c_uint32(t), C.c_uint64(N), blk.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p), C.c_uint32(1023),
                                            C.c_uint32(len(tub)), tub.ctypes.data_as(C.c_void_p), C.byref(u4), C.byref(i14))
            thr4 = int(u4.value)

            def live_fixed(ub, sz, th):
                per_tile = 8192 // sz
                pad = (-len(ub)) % per_tile
                x = np.concatenate([ub, np.zeros(pad, ub.dtype)]).reshape(-1, per_tile).max(axis=1)
                return int((x > th).sum())

            def live_cuts(ub, th, cuts=cuts):
                tile_of = cuts // 8192
                mx = np.zeros(ntiles, np.int64)
                np.maximum.at(mx, tile_of, ub)
                return int((mx > th).sum())
            def tmax_of(c):
                return np.concatenate([c, np.zeros((-N) % 8192, c.dtype)]).reshape(-1, 8192).max(axis=1)
            tmax = tmax_of(counts)
            r = dict(t=t, M=int(counts.max()), thr=thr, thr4=thr4, need4=int((tmax_of(counts) > thr4).sum()), l64_4=live_fixed(ub64, 64, thr4),
                     l16_4=live_fixed(ub16, 16, thr4), l8_4=live_fixed(ub8, 8, thr4))
            for d in (0, 30, 60):
                r[f"need+{d}"] = int((tmax > thr + d).sum())
                r[f"l64+{d}"] = live_fixed(ub64, 64, thr + d)
                r[f"l32+{d}"] = live_fixed(ub32, 32, thr + d)
                r[f"l16+{d}"] = live_fixed(ub16, 16, thr + d)
                r[f"lsp+{d}"] = live_cuts(ubsp, thr + d)
                for g, c in greedy_cuts.items():
                    r[f"lg{g}+{d}"] = live_cuts(ubg[g], thr + d, c)
            # slack of the bound where it matters: blocks whose bound exceeds the threshold although no member does
            mx64 = np.concatenate([counts, np.zeros((-N) % 64, counts.dtype)]).reshape(-1, 64).max(axis=1)
            false64 = (ub64 > thr) & (mx64 <= thr)
            r["false_blocks64"] = int(false64.sum())
            r["slack64_med"] = float(np.median((ub64 - mx64)[false64])) if false64.any() else 0.0
            rows.append(r)
        keys = [k for k in rows[0] if k not in ("t",)]
        print(f"mu_q = {mu}: means over {NQ} queries ({ntiles} tiles)")
        print("   " + "  ".join(f"{k}={np.mean([r[k] for r in rows]):.1f}" for k in keys))
    print(f"total {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
