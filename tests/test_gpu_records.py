"""The RECORDS path of the tile pruning (RTX_OPT_RECORDS, rtx_records.hip) and the two-stream overlap (RTX_OPT_OVERLAP) under the oracle.

A pruned query with few live tiles leaves (reference, count) records of the counts above its threshold instead of the counts of every
reference of its tiles; one wave per query turns them into the prefix sums of lineage.rs:61-77 and walks the lineage.  What must hold:
  * the rows are those of the dense path (taxon_prefix sweeps + walk) and of the oracle -- on the bench workload, on queries far from
    their best hit (both paths in one batch), and when the boundary entries of a query do not fit LDS (the written-out prefix row);
  * the records themselves are the oracle's counts above the threshold, in reference order (checks.check_run_as_left through the
    as-run tap, which rebuilds the counts from the record segments);
  * running the back half of a sub-batch on a second stream changes nothing."""
import numpy as np
import pytest

import raxtax_amd as rx
from gpu_common import Excuses, check_properties, oracle_sample_parity
from raxtax_amd import synth

pytestmark = pytest.mark.gpu

FIELDS = ("row_off", "row_lineage", "row_conf", "row_local_signal", "global_signal", "t", "status")


def same_results(a, b, what):
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f"{what}: {f} differs"


def test_records_path_equals_dense_path_and_oracle(oracle, emul):
    """configs[1]-sized database (7 tiles): the same batch through records + overlap (the default), records on one stream, three stages,
    and the dense epilogues; then the sample against the oracle as the records run left it."""
    n_refs, n_q = 50_000, 40_000
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, n_q, seed=21)
    # a quarter of the batch far from its source (10 % substitutions): thresholds near the background, many live tiles -> dense epilogues
    far = synth.make_queries(db, n_q // 4, seed=22, mu_q=0.10, exact_frac=0.0)
    bases = np.concatenate([qs.bases, far.bases])
    off = np.concatenate([qs.base_off, far.base_off[1:] + qs.base_off[-1]])
    n_all = n_q + n_q // 4
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    index = rx.Index(tree, debug_taps=True, sub_batch=8192)
    res = index.classify(bases, off)
    check_properties(res, db, n_all)
    st = index.debug_prune_stats()
    print("records + overlap:", st)
    assert st["record_queries"] > 0.7 * n_q and st["records_per_record_query"] >= 1 and st["bound_violations"] == 0
    assert st["record_queries"] < n_all, "every query took the records path: the dense epilogues were not exercised"
    for opt, val, what in ((19, 0, "one stream"), (19, 2, "three stages"), (18, 0, "dense epilogues"), (18, 16, "records up to 16 live tiles")):
        rx._lib.check(index._lib.rtx_index_set_option(index._h, opt, val))
        other = index.classify(bases, off)
        st2 = index.debug_prune_stats()
        print(what, st2)
        if opt == 18:
            assert (st2["record_queries"] == 0) == (val == 0)
        same_results(res, other, what)
        rx._lib.check(index._lib.rtx_index_set_option(index._h, opt, 1 if opt == 19 else 4))
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)

    class Both:  # oracle_sample_parity classifies a sample of fixed-length queries of `qs`
        pass
    both = Both()
    both.bases, both.base_off = bases, off
    sample = np.sort(np.random.default_rng(5).choice(n_all, 300, replace=False))
    for skip in (False, True):
        ex = Excuses(f"records/50k/skip={int(skip)}")
        oracle_sample_parity(index, oracle, otree, db, both, sample, skip, ex, full_res=None if skip else res, chunk=100, emul=emul)
        ex.check()


def test_records_slow_path_when_entries_do_not_fit_lds(oracle):
    """30 000 near-identical references, every one a species of its own: all four tiles stay live, thousands of references lie above any
    threshold and every one of them sits between two taxonomy boundaries of its own -- more boundary entries than records_tail_kernel
    keeps in LDS, so it writes the query's prefix row out and walks it from memory.  Rows = the dense path's = the oracle's."""
    L, n_refs, n_q = 400, 30_000, 600
    rng = np.random.default_rng(3)
    root = synth._draw(rng, (1, L))
    seqs = synth._mutate(rng, np.repeat(root, n_refs, axis=0), 0.01)
    seq_bytes = synth.ONE_HOT[seqs].reshape(-1)
    seq_off = np.arange(n_refs + 1, dtype=np.uint64) * np.uint64(L)
    lineages = [f"p:P{i % 3},c:C{i % 30},o:O{i % 300},f:F{i % 3000},g:G{i},s:S{i}" for i in range(n_refs)]
    qsrc = rng.integers(0, n_refs, n_q)
    q = synth.ONE_HOT[synth._mutate(rng, seqs[qsrc], 0.01)].reshape(-1)
    qoff = np.arange(n_q + 1, dtype=np.uint64) * np.uint64(L)
    tree = rx.Tree.new_flat(lineages, seq_bytes, seq_off, kmer_map=False)
    index = rx.Index(tree, prune_self_sample=False)   # (near-identical references: the handle's self-sample leaves tile pruning off here -- the test is about the pruned path)
    assert rx.Index(tree).prune_verdict[0] is False
    res = index.classify(q, qoff)
    st = index.debug_prune_stats()
    print("near-identical references:", st)
    assert (res.status == 0).all()
    if st["pairs"] == 0 or st["record_queries"] == 0:
        pytest.skip("this database did not prune / no query took the records path")
    assert st["record_slow_path_queries"] > 0, "no query overflowed the LDS entries: the slow path was not exercised"
    rx._lib.check(index._lib.rtx_index_set_option(index._h, 18, 0))
    dense = index.classify(q, qoff)
    assert index.debug_prune_stats()["record_queries"] == 0
    # Equal references give sibling taxa EXACTLY equal probabilities here: which of them a tie picks (lineage.rs:158-166: the last maximum)
    # depends on the last bits of the prefix sums, and the two paths add in different orders (a wave scan per 64 records / a block scan
    # per 1024 references; the reference adds one by one).  What must agree: the number of rows and their confidence vectors.
    assert np.array_equal(res.row_off, dense.row_off) and np.array_equal(res.t, dense.t) and np.array_equal(res.global_signal, dense.global_signal)
    n_id = 0
    for j in range(n_q):
        a, b = int(res.row_off[j]), int(res.row_off[j + 1])
        ca, cb = res.row_conf[a:b], dense.row_conf[a:b]
        assert np.array_equal(ca, cb), f"query {j}: confidence vectors of the records path and of the dense path differ"
        n_id += np.array_equal(res.row_lineage[a:b], dense.row_lineage[a:b])
    print(f"{n_id} of {n_q} queries with the same lineages on both paths")
    otree = oracle.tree_new_flat(lineages, seq_bytes, seq_off)
    bad, rows_o, nrows_o = otree.classify_batch(q[: 100 * L], qoff[:101], skip_exact=False, raw_confidence=True, threads=8, cap=64)
    assert bad == 0
    n_same = 0
    for j in range(100):
        want = otree.rows_of(rows_o, nrows_o, j, 64)
        got = res.rows(j)
        assert [g.confidence_values for g in got] == [r["conf"] for r in want], f"query {j}: confidence vectors differ from the oracle's"
        n_same += [g.lineage for g in got] == [r["idx"] for r in want]
    print(f"{n_same} of 100 queries with the oracle's lineages as well")


def test_counts_rows_are_handed_out_and_grow_when_they_run_out(oracle, emul):
    """The diet of the counts buffer (round 6): behind tile pruning with the records path only the queries that take the dense epilogues get
    a row of counts (prune_kernel hands them out: HitParams::cnt_row); the buffer starts with an eighth of a sub-batch's rows.  A batch
    of barcodes fits; a batch of reads 10 % from their source (tens of live tiles each: nearly every query dense) runs out of rows -- the
    download doubles them and repeats the run until it fits.  Either way: the rows of a handle that keeps every row (RTX_OPT_RECORDS = 0),
    and the run as it was left against the oracle (the dense queries' counts through their rows)."""
    from gpu_common import as_run_oracle_sample

    n_refs, n_q = 140_000, 40_000
    db = synth.make_db(n_refs)
    near = synth.make_queries(db, n_q, seed=41)
    far = synth.make_queries(db, n_q, seed=42, mu_q=0.10, exact_frac=0.0)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    diet = rx.Index(tree, debug_taps=True)
    full = rx.Index(tree, records=0)
    full.upload(near.bases, near.base_off)
    full_counts = full.workspace_parts()["counts"]
    for qs, what in ((near, "barcodes 2 % from their source"), (far, "reads 10 % from their source")):
        want = full.classify(qs.bases, qs.base_off)
        got = diet.classify(qs.bases, qs.base_off)
        st = diet.debug_prune_stats()
        parts = diet.workspace_parts()
        print(f"{what}: counts buffer {parts['counts'] / 1e9:.2f} GB (every row: {full_counts / 1e9:.2f} GB); record queries {st['record_queries']} of {n_q}")
        same_results(got, want, what)
        seen = as_run_oracle_sample(diet, got, oracle, otree, qs.bases, qs.base_off, 150, False, threads=8, emul=emul)
        print("as the run left them:", seen)
        assert seen["n"] == 150 and seen["max_dp"] < 1e-9
        if qs is near:
            assert parts["counts"] < 0.3 * full_counts and st["record_queries"] > 0.9 * n_q
            assert seen["on_records_path"] < seen["n"] or True
        else:
            assert st["record_queries"] < 0.5 * n_q and parts["counts"] > 0.4 * full_counts     # the rows have grown
            assert seen["on_records_path"] < 100                                                  # dense queries read back through their rows
    again = diet.classify(near.bases, near.base_off)                                                # (the larger buffer stays; the results do not move)
    same_results(again, full.classify(near.bases, near.base_off), "barcodes again")
