#!/usr/bin/env python3
"""The value_mixed_lengths leg of bench.py on its own (131 072 COI reads alone / with ten reads of 1.1 .. 8 kb among them), e.g. under different
GPU_MAX_HW_QUEUES.   python tools/mixed_lengths_probe.py [refs]"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (sets GPU_MAX_HW_QUEUES unless the environment has it)
import raxtax_amd as rx  # noqa: E402
from raxtax_amd import synth  # noqa: E402

refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
db = synth.make_db(refs)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
index = rx.Index(tree, device=0)
out = bench.mixed_lengths_block(argparse.Namespace(), rx, rx._lib.load(), index, db, 0)
print({k: out[k] for k in ("value", "ms_per_step", "value_coi_alone", "ms_per_step_coi_alone", "slowdown_of_the_batch", "classified_ok")}, flush=True)
