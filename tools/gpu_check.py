#!/usr/bin/env python3
"""Stage-by-stage diagnostic of the HIP path against the oracle (development aid; the judged
parity tests are tests/test_gpu_parity.py).  Prints a summary per stage instead of stopping at
the first mismatch."""
import sys
import time
import traceback
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import raxtax_amd as rx  # noqa: E402
from oracle.oracle_py import Oracle  # noqa: E402
from raxtax_amd import synth  # noqa: E402


def main():
    n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 5184
    n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    orc = Oracle()
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, n_q, exact_frac=0.2, n_frac=0.05)
    t0 = time.time()
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    otree = orc.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    print(f"trees built in {time.time() - t0:.2f}s; devices: {rx._lib.load().rtx_device_count()}")
    ix = rx.Index(tree)
    print("index bytes", ix.device_bytes)
    ex_ids, ex_off = ix.exact_matches(qs.bases, qs.base_off)
    for skip in (False, True):
        print(f"==== skip_exact_matches={skip}")
        try:
            res = ix.classify(qs.bases, qs.base_off, ex_ids, ex_off, skip_exact_matches=skip)
        except Exception:
            traceback.print_exc()
            continue
        print("stage times", ix.stage_times())
        bad = dict(kmers=0, counts=0, probs=0, rows=0, conf=0, signal=0, status=0)
        worst_p = 0.0
        for q in range(qs.n):
            seq = qs.seq(q)
            try:
                km = ix.debug_kmers(q)
                if not np.array_equal(km, orc.sequence_to_kmers(seq)):
                    bad["kmers"] += 1
                    if bad["kmers"] <= 2:
                        print(" kmers differ q", q, len(km), len(orc.sequence_to_kmers(seq)), km[:8], orc.sequence_to_kmers(seq)[:8])
                t, counts = otree.hit_counts(seq, skip_exact=skip)
                got = ix.debug_hit_counts(q)
                if not np.array_equal(got, counts):
                    bad["counts"] += 1
                    if bad["counts"] <= 3:
                        d = np.nonzero(got != counts)[0]
                        print(f" counts differ q{q}: {len(d)} refs, first {d[:6]} got {got[d[:6]]} want {counts[d[:6]]}")
                if res.status[q] != 0:
                    bad["status"] += 1
                    continue
                pref = orc.highest_hit_prob_per_reference(t, t // 2, counts)
                p = ix.debug_probs(q)
                err = float(np.max(np.abs(p - pref)))
                worst_p = max(worst_p, err)
                if not err < 1e-9:
                    bad["probs"] += 1
                    if bad["probs"] <= 3:
                        print(f" probs differ q{q}: max err {err}, sum {p.sum()}")
                rows, _ = otree.classify(seq, skip_exact=skip, raw_confidence=True)
                g = res.rows(q)
                if [r.lineage for r in g] != [r["idx"] for r in rows]:
                    bad["rows"] += 1
                    if bad["rows"] <= 3:
                        print(f" rows differ q{q}: got {[(r.lineage, r.confidence_values) for r in g][:3]} want {[(r['idx'], r['conf']) for r in rows][:3]}")
                    continue
                for a, b in zip(g, rows):
                    if a.confidence_values != b["conf"]:
                        bad["conf"] += 1
                    if abs(a.local_signal - b["local_signal"]) > 1e-9 or abs(a.global_signal - b["global_signal"]) > 1e-9:
                        bad["signal"] += 1
                        if bad["signal"] <= 3:
                            print(f" signal differ q{q}: {a.local_signal} {b['local_signal']} {a.global_signal} {b['global_signal']}")
            except Exception:
                traceback.print_exc()
                break
        print("mismatches:", bad, "worst prob err", worst_p)
    print("work", ix.work())


if __name__ == "__main__":
    main()
