"""BASELINE.json configs[4] at its database size on ONE GPU: 5 000 000 references cut into two reference shards that
live side by side in HBM (2 x 20 GB of bitmaps; the unsharded 41 GB index next to them), the exchange of
raxtax_amd/sharded.py emulated in-process.  The sharded result must equal the unsharded one, and a seeded sample of
the queries is checked against the CPU oracle (hit counts bit-exact, rows identical)."""
import os

import numpy as np
import pytest

import raxtax_amd as rx
from gpu_common import Excuses
from raxtax_amd import sharded, synth

pytestmark = pytest.mark.gpu

N_REFS, N_Q, N_ORACLE = 5_000_000, 512, 24


@pytest.mark.skipif(os.environ.get("RTX_SKIP_5M") == "1", reason="RTX_SKIP_5M=1")
def test_two_reference_shards_at_5m(oracle):
    from test_gpu_parity import assert_rows_equivalent

    db = synth.make_db(N_REFS)
    qs = synth.make_queries(db, N_Q)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)                 # with Tree.k_mer_map: the shards are cut out of it
    whole = rx.Index(rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False), cluster=False)
    assert whole.device_bytes > 40e9
    ex = whole.exact_matches(qs.bases, qs.base_off)
    ref = whole.classify(qs.bases, qs.base_off, *ex)
    assert (ref.status == 0).all()
    cuts = sharded.shard_cuts(tree.num_tips, 2)
    shards = [sharded.ShardIndex(tree, r, cuts, sub_batch=256) for r in range(2)]
    got = sharded.ShardedClassifier(shards, sharded.LocalComm()).classify(qs.bases, qs.base_off, *ex)
    for f in ("row_off", "row_conf", "t", "status"):
        assert np.array_equal(getattr(got, f), getattr(ref, f)), f
    assert np.max(np.abs(got.global_signal - ref.global_signal)) < 1e-12
    assert np.mean(got.row_lineage != ref.row_lineage) < 0.01          # exact ties only (offset-added prefix sums)
    # per-shard hit counts are the slices of the unsharded ones (the last sub-batch is still resident)
    q = N_Q - 1
    full = whole.debug_hit_counts(q)
    for s in shards:
        assert np.array_equal(s.debug_hit_counts(q), full[s.ref_lo:s.ref_hi])
    # the oracle on a seeded sample
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    sample = np.sort(np.random.default_rng(5).choice(N_Q, N_ORACLE, replace=False))
    L = db.length
    sub = np.ascontiguousarray(qs.bases.reshape(-1, L)[sample]).reshape(-1)
    off = (np.arange(N_ORACLE + 1) * L).astype(np.uint64)
    threads = os.cpu_count() or 1
    t_o, counts_o = otree.hit_counts_batch(sub, off, threads=threads)
    bad, rows_o, nrows_o = otree.classify_batch(sub, off, raw_confidence=True, threads=threads, cap=64)
    assert bad == 0
    whole.classify(sub, off, *whole.exact_matches(sub, off))
    exc = Excuses("config4/5M/two-shards")
    lineages = None
    for j, qi in enumerate(sample):
        assert np.array_equal(whole.debug_hit_counts(j), counts_o[j]), int(qi)
        want = otree.rows_of(rows_o, nrows_o, j, 64)
        g = got.rows(int(qi))
        exc.checked += 1
        if [r.lineage for r in g] != [r["idx"] for r in want] or [r.confidence_values for r in g] != [r["conf"] for r in want]:
            lineages = lineages or otree.lineages
            tables, z, rc = oracle.prob_tables_batch(t_o[j:j + 1], counts_o[j:j + 1])
            exc.tie(assert_rows_equivalent(g, want, tables[0][counts_o[j]], lineages, f"query {int(qi)}"))
    exc.check()
