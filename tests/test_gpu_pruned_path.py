"""The pruned path itself under the oracle: what the timed run computed -- the counts of the tiles it visited, the
tiles it left out, the histogram, the probabilities -- is read back WITHOUT a recount (rtx_debug_run_counts,
rtx_debug_pruned_prob_table, rtx_debug_prune_detail) and held against the oracle's full computation (gpu_common.check_run_as_left), at
the sizes and on the kind of data the other suites do not reach:
  * a database of more than 524 288 references: the bounds of the tile pruning span more than one tile of the union bitmap;
  * real barcode composition: the 600 Diptera records of the reference's example data (tests/golden/diptera_subset.fasta, 195-208 bp,
    t ~ 195, k-mers that occur in most references) expanded by per-copy substitutions to 15 tiles.
BASELINE configs[2] gets the same checks in tests/test_gpu_config2.py, configs[4]'s database size in tests/test_gpu_config4_shards.py."""
from pathlib import Path

import numpy as np
import pytest

import raxtax_amd as rx
from gpu_common import Excuses, check_properties, oracle_sample_parity
from raxtax_amd import synth

pytestmark = pytest.mark.gpu

FASTA = Path(__file__).resolve().parent / "golden" / "diptera_subset.fasta"


def test_more_than_one_tile_of_bounds(oracle, emul):
    """N = 620 000: 9 688 blocks of 64 references = two tiles of bounds (u_ntiles = 2), 76 tiles of references."""
    n_refs, n_q, n_sample = 620_000, 20_000, 300
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, n_q, seed=11)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    index = rx.Index(tree, debug_taps=True)
    res = index.classify(qs.bases, qs.base_off, *index.exact_matches(qs.bases, qs.base_off))
    check_properties(res, db, n_q)
    st = index.debug_prune_stats()
    print("tile pruning of the full batch:", st)
    assert st["pairs"] == (n_q + 1) // 2 and st["bound_violations"] == 0 and st["queries_with_threshold"] > 0.9 * n_q
    assert st["live_tiles_per_pair"] < 10
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    sample = np.sort(np.random.default_rng(62).choice(n_q, n_sample, replace=False))
    for skip in (False, True):
        ex = Excuses(f"pruned/620k/skip={int(skip)}")
        oracle_sample_parity(index, oracle, otree, db, qs, sample, skip, ex, full_res=None if skip else res, chunk=100, emul=emul)
        ex.check()
    # Queries far from their best hit (10 % substitutions) keep tens of tiles each: the counting pass walks its list of live (pair, tile)
    # blocks in several passes of the grid (rtx_hit_pair.hip: more blocks than workgroups), in the full batch and in the sample's own.
    # First with the first stage of the bounds alone (RTX_OPT_FINE_BOUNDS = 0: the long lists), then with the second stage: the pairs with
    # four live tiles or more are counted against the union bitmap over blocks of 8 references, which takes most of their tiles off
    # the lists -- same rows, every sampled query as the run left it against the oracle, far fewer blocks counted.
    n_q2 = 3000
    qs2 = synth.make_queries(db, n_q2, seed=12, mu_q=0.10, exact_frac=0.0)
    ex2 = index.exact_matches(qs2.bases, qs2.base_off)
    rx._lib.check(index._lib.rtx_index_set_option(index._h, 17, 0))
    res2 = index.classify(qs2.bases, qs2.base_off, *ex2)
    check_properties(res2, db, n_q2)
    st2 = index.debug_prune_stats()
    print("divergent queries, first stage only:", st2)
    assert st2["bound_violations"] == 0 and st2["live_tiles_per_pair"] * st2["pairs"] > 3 * n_q2      # grid: 2 workgroups per pair and pass
    assert st2["fine_blocks_per_pair"] == 0
    sample2 = np.sort(np.random.default_rng(63).choice(n_q2, 200, replace=False))
    ex = Excuses("pruned/620k/divergent")
    oracle_sample_parity(index, oracle, otree, db, qs2, sample2, False, ex, full_res=res2, chunk=100, emul=emul)
    assert index.debug_prune_stats()["live_tiles_per_pair"] * 100 > 2048
    ex.check()
    rx._lib.check(index._lib.rtx_index_set_option(index._h, 17, 1))
    res3 = index.classify(qs2.bases, qs2.base_off, *ex2)
    st3 = index.debug_prune_stats()
    print("divergent queries, both stages:", st3)
    assert st3["bound_violations"] == 0 and st3["fine_blocks_per_pair"] > 1 and st3["fine_cleared_per_query"] > 1
    assert st3["live_tiles_per_pair_first_stage"] == st2["live_tiles_per_pair"] and st3["live_tiles_per_pair"] < 0.5 * st2["live_tiles_per_pair"]
    for f in ("row_off", "row_lineage", "row_conf", "row_local_signal", "global_signal", "t", "status"):
        assert np.array_equal(getattr(res3, f), getattr(res2, f)), f     # a tile taken off the list held nothing above the threshold
    ex = Excuses("pruned/620k/divergent/two-stage")
    oracle_sample_parity(index, oracle, otree, db, qs2, sample2, False, ex, full_res=res3, chunk=100, emul=emul)
    ex.check()


def _diptera_expanded(copies, seed=7):
    """The 600 Diptera records, `copies` copies each: copy 0 is the record itself, the others carry substitutions at a per-copy
    rate drawn from {0.2, 0.5, 1, 2, 4} % (bases redrawn from the record's own composition).  Lineage = the record's."""
    code = np.zeros(256, np.uint8)
    for ch, v in zip("ACGT", (1, 2, 4, 8)):
        code[ord(ch)] = code[ord(ch.lower())] = v
    recs, lin, cur = [], [], []
    for line in FASTA.read_text().splitlines():
        if line.startswith(">"):
            if cur:
                recs.append("".join(cur))
            cur = []
            lin.append(line.split("tax=")[1].split(";")[0])
        elif line.strip():
            cur.append(line.strip())
    recs.append("".join(cur))
    assert len(recs) == len(lin) == 600
    rng = np.random.default_rng(seed)
    seqs, lineages = [], []
    for r, l in zip(recs, lin):
        base = code[np.frombuffer(r.encode(), np.uint8)]
        assert (base > 0).all()
        for c in range(copies):
            s = base.copy()
            if c:
                mu = float(rng.choice([0.002, 0.005, 0.01, 0.02, 0.04]))
                hit = rng.random(len(s)) < mu
                s[hit] = rng.choice(base, int(hit.sum()))
            seqs.append(s)
            lineages.append(l)
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return lineages, np.concatenate(seqs), off, seqs


def test_real_composition_short_reads(oracle, emul):
    """Diptera COI at ~205 bp (t ~ 195, n ~ 97): 115 200 references = 15 tiles; queries = references with 0 .. 8 % substitutions,
    truncated reads and the raw records (exact matches of copy 0).  Variable lengths: the sample is compared query by query."""
    from test_gpu_parity import assert_rows_equivalent

    lineages, flat, off, seqs = _diptera_expanded(192)
    n_refs = len(seqs)
    assert (n_refs + 8191) // 8192 == 15
    rng = np.random.default_rng(8)
    qs = []
    for i in range(3000):
        s = seqs[int(rng.integers(0, n_refs))].copy()
        kind = i % 6
        mu = (0.0, 0.01, 0.03, 0.08, 0.02, 0.0)[kind]
        hit = rng.random(len(s)) < mu
        s[hit] = rng.choice(s, int(hit.sum()))
        if kind == 4:
            s = s[: int(rng.integers(120, len(s)))]           # a truncated read
        if kind == 5:
            s = seqs[int(rng.integers(0, 600)) * 192].copy()    # a raw record
        qs.append(s)
    qoff = np.zeros(len(qs) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(q) for q in qs])
    bases = np.concatenate(qs)
    otree = oracle.tree_new_flat(lineages, flat, off)
    tree = rx.Tree.new_flat(lineages, flat, off, kmer_map=False)
    index = rx.Index(tree, debug_taps=True, prune_self_sample=False)   # (a database of near-copies: its self-sample would leave tile pruning off)
    ex_ids, ex_off = index.exact_matches(bases, qoff)
    lf = np.array([oracle.lib.orc_ln_factorial(i) for i in range(2 * 210 + 8)], dtype=np.float64)
    olin = None
    import os
    from gpu_common import check_run_as_left

    threads = os.cpu_count() or 1
    for skip in (False, True):
        res = index.classify(bases, qoff, ex_ids, ex_off, skip_exact_matches=skip)
        assert (res.status == 0).all()
        st = index.debug_prune_stats()
        print(f"skip={skip}: tile pruning on real composition:", st)
        assert st["pairs"] == len(qs) // 2 and st["bound_violations"] == 0
        # the whole batch is one sub-batch: every query can be read back as the run left it
        sample = np.sort(np.random.default_rng(9 + skip).choice(len(qs), 400, replace=False))
        sub = np.concatenate([qs[j] for j in sample])
        soff = np.zeros(len(sample) + 1, np.uint64)
        soff[1:] = np.cumsum([len(qs[j]) for j in sample])
        t_o, counts_o = otree.hit_counts_batch(sub, soff, skip_exact=skip, threads=threads)
        tables_o, z_o, rc = oracle.prob_tables_batch(t_o, counts_o, threads=threads)
        assert (rc == 0).all()
        # tol 1e-7: with 192 near-identical copies of every record hundreds of references share the top counts, the sums
        # sum_m hist[m] ln cmf_m(i) of prob.rs:62-73 reach 1e4 and more, and two correct f64 evaluations of them (the oracle's order of
        # summation, the device's) differ by up to 1e-9 relative (measured: 8.8e-10) -- the reference's own order is random (ahash,
        # SURVEY.md 8c).  north_star: 1e-6.
        seen = [check_run_as_left(index, int(j), int(t_o[k]), counts_o[k], tables_o[k], n_refs, emul, lf, f"query {int(j)} skip {skip}", tol=1e-7)
                for k, j in enumerate(sample)]
        with_thr = sum(o["threshold"] > 0 for o in seen)
        print(f"skip={skip}: {len(seen)} queries read back as the run left them: {with_thr} with a threshold (mean {np.mean([o['threshold'] for o in seen]):.1f} "
              f"of t ~ {np.mean(t_o):.0f}), {np.mean([o['live'] for o in seen]):.2f} of 15 tiles visited per query, "
              f"{np.mean([o['needed'] for o in seen]):.2f} hold a count above the threshold; max |p - p_oracle| {max(o['dp'] for o in seen):.2e}, "
              f"dropped mass {max(o['dropped'] for o in seen):.2e}")
        # then the rows, and the recounting taps (hit counts of every tile)
        bad, rows_o, nrows_o = otree.classify_batch(sub, soff, skip_exact=skip, raw_confidence=True, threads=threads, cap=64)
        assert bad == 0
        exc = Excuses(f"pruned/diptera115k/skip={int(skip)}")
        for k, j in enumerate(sample):
            j = int(j)
            assert res.t[j] == t_o[k]
            if k % 8 == 0:
                assert np.array_equal(index.debug_hit_counts(j), counts_o[k]), (skip, j)
            want = otree.rows_of(rows_o, nrows_o, k, 64)
            got = res.rows(j)
            exc.checked += 1
            if [g.lineage for g in got] != [r["idx"] for r in want] or [g.confidence_values for g in got] != [r["conf"] for r in want]:
                olin = olin or otree.lineages
                exc.tie(assert_rows_equivalent(got, want, tables_o[k][counts_o[k]], olin, f"query {j} skip {skip}"))
        exc.check()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_reference_shards_prune_with_the_global_threshold(oracle, n_shards):
    """BASELINE configs[4] mode B with tile pruning: every shard (9 tiles or more) counts the queries against its own union bitmap, the
    shards exchange their candidates for the best block (RTX_BUF_BEST: 264 bytes per query), every shard derives the threshold from the
    best block of the WHOLE database and counts its live tiles only.  Against the oracle (rows, every query) and against the unsharded
    pruned run (same rows; the thresholds may differ by a few counts where a shard's blocks of 64 are cut elsewhere); cuts inside taxa,
    --skip-exact-matches (an exact match of the best block may live on another shard), several sub-batches."""
    from raxtax_amd import sharded
    from test_gpu_parity import assert_rows_equivalent

    n_refs = 8192 * 9 * n_shards + 4321
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, 700, seed=21 + n_shards, exact_frac=0.15)
    L = db.length
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    whole = rx.Index(tree)
    ex = whole.exact_matches(qs.bases, qs.base_off)
    cuts = sharded.shard_cuts(tree.num_tips, n_shards)
    cuts[1] += 37                                         # not a multiple of 64, inside a species
    shards = [sharded.ShardIndex(tree, r, cuts, sub_batch=256) for r in range(n_shards)]
    clf = sharded.ShardedClassifier(shards, sharded.LocalComm())
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    import os
    threads = os.cpu_count() or 1
    lineages = otree.lineages
    for skip in (False, True):
        ref = whole.classify(qs.bases, qs.base_off, *ex, skip_exact_matches=skip)
        assert whole.debug_prune_stats()["pairs"] > 0
        got = clf.classify(qs.bases, qs.base_off, *ex, skip_exact_matches=skip)
        assert all(s.prunes for s in shards)
        live = [s.debug_prune_stats() for s in shards]
        print(f"{n_shards} shards, skip={skip}: live tiles per pair and shard {[round(x['live_tiles_per_pair'], 2) for x in live]} of {[(s.n_refs + 8191) // 8192 for s in shards]}, "
              f"unsharded {whole.debug_prune_stats()['live_tiles_per_pair']:.2f}; bound violations {[x['bound_violations'] for x in live]}")
        assert all(x["bound_violations"] == 0 and x["pairs"] > 0 for x in live)
        assert sum(x["live_tiles_per_pair"] for x in live) < 0.5 * sum((s.n_refs + 8191) // 8192 for s in shards)
        assert np.array_equal(got.t, ref.t) and np.array_equal(got.status, ref.status)
        assert np.max(np.abs(got.global_signal - ref.global_signal)) < 1e-9
        bad, rows_o, nrows_o = otree.classify_batch(qs.bases, qs.base_off, skip_exact=skip, raw_confidence=True, threads=threads, cap=64)
        assert bad == 0
        exc = Excuses(f"pruned/shards{n_shards}/skip={int(skip)}")
        for q in range(qs.n):
            want = otree.rows_of(rows_o, nrows_o, q, 64)
            for res, name in ((got, "sharded"), (ref, "whole")):
                g = res.rows(q)
                if name == "sharded":
                    exc.checked += 1
                if [r.lineage for r in g] != [r["idx"] for r in want] or [r.confidence_values for r in g] != [r["conf"] for r in want]:
                    t, counts = otree.hit_counts(qs.seq(q), skip_exact=skip)
                    tables, z, rc = oracle.prob_tables_batch(np.array([t], np.uint32), counts[None, :])
                    ties = assert_rows_equivalent(g, want, tables[0][counts], lineages, f"{name}: query {q} skip {skip}")
                    if name == "sharded":
                        exc.tie(ties)
        exc.check()
        # the counts every shard wrote for the tiles it visited are the oracle's (last sub-batch: still resident, as the run left it)
        perm = shards[0].debug_order(qs.n)
        assert all(np.array_equal(s.debug_order(qs.n), perm) for s in shards)     # the shards agree on the processing order
        q = int(perm[-1])
        t, counts = otree.hit_counts(qs.seq(q), skip_exact=skip)
        for s in shards:
            rc = s.debug_run_counts(q, t)
            lv = np.repeat(rc["tile_live"], 8192)[: s.n_refs]
            assert np.array_equal(rc["counts"][lv], counts[s.ref_lo:s.ref_hi][lv]) and rc["threshold"] > 0
            assert (counts[s.ref_lo:s.ref_hi][~lv] <= rc["threshold"]).all()


def test_shards_that_disagree_on_pruning_fall_back_together(oracle):
    """whether a reference shard CAN prune is local state (a union bitmap exists from RTX_PRUNE_MIN_TILES = 4 local tiles on).
    49 153 references over two shards are 24 577 (4 tiles) and 24 576 (3 tiles): left to themselves one shard would process the queries
    in min-hash order and exchange best blocks, the other in input order without -- the histogram all-reduce would add rows of different
    queries.  ShardedClassifier.upload takes the minimum of the shards' verdicts: nobody prunes, and the result is the unsharded one."""
    from raxtax_amd import sharded

    n_refs = 49153
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, 300, seed=31, exact_frac=0.1)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    cuts = sharded.shard_cuts(tree.num_tips, 2)
    assert [(b - a + 8191) // 8192 for a, b in zip(cuts, cuts[1:])] == [4, 3]
    shards = [sharded.ShardIndex(tree, r, cuts, sub_batch=128) for r in range(2)]
    alone = []
    for s in shards:          # each shard on its own: the first could prune, the second cannot
        s.upload(qs.bases, qs.base_off)
        s.begin()
        alone.append(s.prunes)
    assert alone == [True, False]
    whole = rx.Index(tree, tile_prune=False)
    ex = whole.exact_matches(qs.bases, qs.base_off)
    ref = whole.classify(qs.bases, qs.base_off, *ex)
    clf = sharded.ShardedClassifier(shards, sharded.LocalComm())
    got = clf.classify(qs.bases, qs.base_off, *ex)
    assert not any(s.prunes for s in shards) and clf._prunes is False
    assert np.array_equal(got.t, ref.t) and np.array_equal(got.row_off, ref.row_off) and np.array_equal(got.row_lineage, ref.row_lineage)
    assert np.max(np.abs(got.row_conf - ref.row_conf)) < 1e-9 and np.max(np.abs(got.global_signal - ref.global_signal)) < 1e-9


def test_self_sample_decides_per_database():
    """rtx_index_self_sample (RTX_OPT_PRUNE_SELF_SAMPLE): a database of clades that have nothing to do with each other keeps tile pruning,
    a database whose every tile holds relatives of every query (the Diptera records, expanded) leaves it off -- and the rows of a batch are
    the same either way (what pruning drops never reaches a row; test_real_composition_short_reads holds the pruned run against the oracle)."""
    db = synth.make_db(60_000)
    ix = rx.Index(rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False))
    on, frac = ix.prune_verdict
    assert on and 0.0 < frac < 0.5, (on, frac)
    qs = synth.make_queries(db, 512, seed=2)
    ix.classify(qs.bases, qs.base_off)
    assert ix.debug_prune_stats()["pairs"] == 256

    # the database of bench.py's value_real_composition: the reference's hold-out methodology on its Diptera records, 14 tiles
    h = synth.real_composition_holdout(FASTA.parent / "diptera_queries.fasta", n_queries=2048)
    tree = rx.Tree.new_flat(h.lineages, h.seq_bytes, h.seq_off, kmer_map=False)
    auto, forced = rx.Index(tree), rx.Index(tree, prune_self_sample=False)
    on, frac = auto.prune_verdict
    assert not on and frac >= 0.85, (on, frac)
    assert forced.prune_verdict[0] and forced.prune_verdict[1] == frac
    bases, qoff = h.q_bases, h.q_off
    n_queries = len(qoff) - 1
    a = auto.classify(bases, qoff)
    assert auto.debug_prune_stats()["pairs"] == 0             # every tile counted
    b = forced.classify(bases, qoff)
    assert forced.debug_prune_stats()["pairs"] == n_queries // 2
    for f in ("row_off", "row_lineage", "t", "status"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.abs(a.row_conf - b.row_conf).max() < 1e-9
