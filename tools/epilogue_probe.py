#!/usr/bin/env python3
"""Stage times of the real-composition hold-out with every tile counted, for variant builds that break the results (tools/build_variant.py with
experimental macros: parts of the dense epilogue left out): nothing is downloaded or looked at.   python tools/epilogue_probe.py [steps]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
h = synth.real_composition_holdout(ROOT / "tests" / "golden" / "diptera_queries.fasta")
tree = rx.Tree.new_flat(h.lineages, h.seq_bytes, h.seq_off, kmer_map=False)
index = rx.Index(tree, stage_timing=True, prune_self_sample=False)
rx._lib.check(index._lib.rtx_index_set_option(index._h, 13, 0))
n_q = len(h.q_off) - 1
index.upload(h.q_bases, h.q_off)
index.run(0)
index.sync()
t0 = time.perf_counter()
for _ in range(steps):
    index.run(0)
    index.sync()
dt = (time.perf_counter() - t0) / steps
st = {s: round(ms, 2) for s, (ms, n) in index.stage_times().items() if n}
print(f"{dt * 1e3:.2f} ms per step of {n_q} queries (no download); stages {st}", flush=True)
