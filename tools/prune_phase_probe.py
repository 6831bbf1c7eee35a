#!/usr/bin/env python3
"""Where prune_kernel spends its cycles (a -DRTX_PRUNE_PROFILE build of the library: tools/build_variant.py prof -DRTX_PRUNE_PROFILE, run with
RTX_LIB_PATH=gpurun_scratch/lib_prof.so): mean shader-clock cycles per query of gather / i* search / criterion (2) / window + groups / bisection."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402
from raxtax_amd._lib import u64p, ptr  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 262_144
db = synth.make_db(N)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
qs = synth.make_queries(db, Q)
idx = rx.Index(tree)
for rep in range(2):
    idx.upload(qs.bases, qs.base_off)
    idx.run()
out = np.zeros(16, np.uint64)
idx._lib.rtx_debug_prune_stats(idx._h, ptr(out, u64p))
nq = float(out[5])
names = {2: "ub + exact counts of the best block", 3: "i* search", 4: "criterion (2)", 6: "window, groups", 7: "bisection of u"}
tot = sum(float(out[k]) for k in names)
for k, nm in names.items():
    print(f"{nm:40s} {float(out[k]) / nq:10.0f} cycles per query  {100 * float(out[k]) / tot:5.1f} %")
print(f"queries {nq:.0f}, total {tot / nq:.0f} cycles per query")
