"""BASELINE.json configs[4] at its database size on ONE GPU: 5 000 000 references (611 tiles; 77 tiles of bounds for the tile
pruning), unsharded through the default -- pruned -- handle (the 41 GB index) and cut into two reference shards that live beside it
in HBM (2 x 20 GB of bitmaps), the exchange of raxtax_amd/sharded.py emulated in-process.  Every query is checked against the CPU
oracle: the pruned run as it was (visited counts bit-exact, unvisited tiles below the threshold, probabilities), hit counts of a
recount, result rows of both the unsharded and the sharded run (exact ties counted in the Excuses ledger)."""
import os

import numpy as np
import pytest

import raxtax_amd as rx
from gpu_common import Excuses, check_run_as_left
from raxtax_amd import sharded, synth

pytestmark = pytest.mark.gpu

# The default -m gpu run takes the same path at 1 500 000 references (184 tiles, 23 tiles of bounds, two shards of 750 000): the suite has
# to stay within the driver's time limit (135 of its 470 s were this one test).  RTX_FULL_SIZE=1 runs it at configs[4]'s
# 5 000 000 (opt-in; tools/scale_check.py 5000000 is the timed run at that size, profiles/r6*_scale_check_5m.txt).
FULL = os.environ.get("RTX_FULL_SIZE") == "1"
N_REFS, N_Q = (5_000_000 if FULL else 1_500_000), 256


@pytest.mark.skipif(os.environ.get("RTX_SKIP_5M") == "1", reason="RTX_SKIP_5M=1")
def test_5m_references_pruned_and_in_two_shards(oracle, emul):
    from test_gpu_parity import assert_rows_equivalent

    db = synth.make_db(N_REFS)
    qs = synth.make_queries(db, N_Q)
    L = db.length
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)                 # with Tree.k_mer_map: the shards are cut out of it
    whole = rx.Index(rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False), debug_taps=True)   # default options: pruned
    assert whole.device_bytes > (40e9 if FULL else 10e9)
    ex = whole.exact_matches(qs.bases, qs.base_off)
    ref = whole.classify(qs.bases, qs.base_off, *ex)
    assert (ref.status == 0).all()
    st = whole.debug_prune_stats()
    print("tile pruning at 5 M references:", st)
    assert st["pairs"] == N_Q // 2 and st["bound_violations"] == 0 and st["queries_with_threshold"] > 0.9 * N_Q
    assert st["live_tiles_per_pair"] < 60                                          # of 611 (184)

    # ---- the oracle on every query, in chunks (a count vector is 10 MB)
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    threads = os.cpu_count() or 1
    lf = np.array([oracle.lib.orc_ln_factorial(i) for i in range(2 * L + 8)], dtype=np.float64)
    lineages = otree.lineages
    want_rows, tables, counts_keep = {}, {}, {}
    seen = []
    for a in range(0, N_Q, 32):
        b = min(N_Q, a + 32)
        sub, off = qs.bases[a * L:b * L], qs.base_off[a:b + 1] - qs.base_off[a]
        t_o, counts_o = otree.hit_counts_batch(sub, off, threads=threads)
        tables_o, z_o, rc = oracle.prob_tables_batch(t_o, counts_o, threads=threads)
        assert (rc == 0).all()
        bad, rows_o, nrows_o = otree.classify_batch(sub, off, raw_confidence=True, threads=threads, cap=64)
        assert bad == 0
        for q in range(a, b):
            # the pruned run of the whole batch (one sub-batch) exactly as it was
            seen.append(check_run_as_left(whole, q, int(t_o[q - a]), counts_o[q - a], tables_o[q - a], N_REFS, emul, lf, f"query {q}"))
            want_rows[q] = otree.rows_of(rows_o, nrows_o, q - a, 64)
            if q % 16 == 0:
                counts_keep[q] = counts_o[q - a].copy()
            tables[q] = tables_o[q - a].copy()          # table / Z per count; the count vector is recomputed where rows differ
    print(f"{len(seen)} queries read back as the pruned run left them: mean threshold {np.mean([o['threshold'] for o in seen]):.1f}, "
          f"{np.mean([o['live'] for o in seen]):.2f} of {(N_REFS + 8191) // 8192} tiles visited per query ({np.mean([o['needed'] for o in seen]):.2f} hold a count above the "
          f"threshold), max |p - p_oracle| {max(o['dp'] for o in seen):.2e}, dropped mass {max(o['dropped'] for o in seen):.2e}")

    def rows_against_oracle(res, name):
        exc = Excuses(f"config4/5M/{name}")
        for q in range(N_Q):
            got, want = res.rows(q), want_rows[q]
            exc.checked += 1
            if [r.lineage for r in got] != [r["idx"] for r in want] or [r.confidence_values for r in got] != [r["conf"] for r in want]:
                t, counts = otree.hit_counts(qs.seq(q))
                exc.tie(assert_rows_equivalent(got, want, tables[q][counts], lineages, f"{name}: query {q}"))
            else:
                for g, r in zip(got, want):
                    assert abs(g.local_signal - r["local_signal"]) < 1e-6 and abs(g.global_signal - r["global_signal"]) < 1e-9
        exc.check()

    rows_against_oracle(ref, "whole")
    for q, c in counts_keep.items():                                               # the recounting tap: every tile, bit-exact
        assert np.array_equal(whole.debug_hit_counts(q), c), q

    # ---- two reference shards next to it
    cuts = sharded.shard_cuts(tree.num_tips, 2)
    shards = [sharded.ShardIndex(tree, r, cuts, sub_batch=256) for r in range(2)]
    got = sharded.ShardedClassifier(shards, sharded.LocalComm()).classify(qs.bases, qs.base_off, *ex)
    assert np.array_equal(got.t, ref.t) and np.array_equal(got.status, ref.status)
    assert np.max(np.abs(got.global_signal - ref.global_signal)) < 1e-9
    rows_against_oracle(got, "two-shards")
    q = N_Q - 1                                                                    # per-shard hit counts = slices of the oracle's
    t, counts = otree.hit_counts(qs.seq(q))
    for s in shards:
        assert np.array_equal(s.debug_hit_counts(q), counts[s.ref_lo:s.ref_hi])
