#!/usr/bin/env python3
"""Host-side study (no GPU): inside the dense (row, tile) segments a query asks for, which of the eight 128-byte lines
hold any reference at all?  span = last - first + 1 non-empty line (a contiguous range can be had for free with the
buffer descriptor), nz = number of non-empty lines.  Usage: tools/exp_line_span.py [refs] [sample queries]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.oracle_py import Oracle  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
n_s = int(sys.argv[2]) if len(sys.argv) > 2 else 128
db = synth.make_db(n_refs)
o = Oracle()
ot = o.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
off, post = ot.csr()
nt = (n_refs + 8191) // 8192
pop = np.zeros((65536, nt), np.uint32)
lmask = np.zeros((65536, nt), np.uint8)       # bit l: line l (1024 references) of the segment is not empty
for k0 in range(0, 65536, 1024):
    a, b = int(off[k0]), int(off[min(k0 + 1024, 65536)])
    if a == b:
        continue
    lens = np.diff(off[k0:k0 + 1025].astype(np.int64))
    rows = np.repeat(np.arange(len(lens)), lens)
    pp = post[a:b].astype(np.int64)
    key = rows * nt + (pp >> 13)
    pop[k0:k0 + len(lens)] += np.bincount(key, minlength=len(lens) * nt).reshape(len(lens), nt).astype(np.uint32)
    # NOTE: the device permutes references inside a tile (ref_slot): line = 16-byte chunk index / 8, chunk = lane;
    # reference r of a full tile sits in lane ((r & 8191) >> 3) % 64 -- the lanes of 512 consecutive references
    if len(sys.argv) > 3 and sys.argv[3] == "natural":     # lane l <-> 128 consecutive references: line = 1024 consecutive references
        line = (pp & 8191) >> 10
    else:
        lane = ((pp & 8191) >> 3) & 63
        line = lane >> 3
    np.bitwise_or.at(lmask.reshape(-1), key + k0 * nt, (1 << line).astype(np.uint8))
qs = synth.make_queries(db, 20000)
tot_dense = 0
span_lines = 0
nz_lines = 0
by_class = {}
edges = [17, 33, 65, 129, 257, 513, 1025, 2049, 4097, 8193]
for q in range(n_s):
    km = o.sequence_to_kmers(qs.seq(q)).astype(np.int64)
    p = pop[km].reshape(-1)
    m = lmask[km].reshape(-1)
    d = p > 16
    mm = m[d].astype(np.uint32)
    pd = p[d]
    nz = np.array([bin(x).count("1") for x in range(256)])[mm]
    first = np.array([(x & -x).bit_length() - 1 if x else 0 for x in range(256)])[mm]
    last = np.array([x.bit_length() - 1 if x else 0 for x in range(256)])[mm]
    span = last - first + 1
    tot_dense += d.sum()
    span_lines += span.sum()
    nz_lines += nz.sum()
    for i in range(len(edges) - 1):
        sel = (pd >= edges[i]) & (pd < edges[i + 1])
        c = by_class.setdefault(i, [0, 0, 0])
        c[0] += sel.sum(); c[1] += span[sel].sum(); c[2] += nz[sel].sum()
print(f"dense segments per query {tot_dense / n_s:.0f}; lines per segment: span {span_lines / tot_dense:.2f}, non-empty {nz_lines / tot_dense:.2f} of 8")
for i, c in by_class.items():
    if c[0]:
        print(f"  [{edges[i]:5d}, {edges[i + 1] - 1:5d}]: {100 * c[0] / tot_dense:5.1f} % of the dense segments, span {c[1] / c[0]:.2f}, non-empty {c[2] / c[0]:.2f}")
