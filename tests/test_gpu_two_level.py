"""The bounds pass of the tile pruning in two levels (rtx_bounds2.hip, RTX_OPT_TWO_LEVEL_BOUNDS) against the one-level pass over blocks
of 64 references (hit_count_pair_kernel<.., 1, ..>) and against the oracle:
  * every tile's bound is an upper bound of every count of the tile (oracle counts), whatever level it comes from;
  * a bound of level A (blocks of 256) is never below the one-level bound of the same tile, a refined tile's bound EQUALS it, and the
    best block of 64 is the one-level pass's whenever its tile was refined (it always is on these workloads); a query far from its best hit
    ("heavy": its rule asks for many B-tiles) carries the one-level bound on EVERY tile -- the one-level pass took it;
  * the results are those of the one-level run (probabilities within the budget of the pruning, rows identical);
  * the rule that picks the refined groups of tiles decides nothing but time: with a rule that refines NOTHING beyond the group of the
    largest bound, and with one that refines everything, the same rows come back.
The pruned runs as they were left (counts of the visited tiles, thresholds against the CPU restatement) are held against the oracle with
the two-level pass on -- the default -- in every other pruned test of the suite (tests/test_gpu_pruned_path.py has two A-tiles)."""
import numpy as np
import pytest

import raxtax_amd as rx
from gpu_common import check_properties, last_sub_batch_queries
from raxtax_amd import synth

pytestmark = pytest.mark.gpu


def _rule(ct, cm, lo, hi):
    return ct | (cm << 16) | (lo << 32) | (hi << 48)


def _taps(index, n_q, n_take):
    qs_last = last_sub_batch_queries(index, n_q)[:n_take]
    return qs_last, {int(q): (index.debug_tile_bounds(int(q)), index.debug_prune_detail(int(q))) for q in qs_last}


@pytest.mark.parametrize("n_refs,mu", [(140_000, 0.02), (140_000, 0.08), (33_000, 0.03)])   # (33 000: 5 tiles -- below kTwoLevelMinTiles the option changes nothing)
def test_two_level_bounds_against_one_level_and_oracle(oracle, n_refs, mu):
    n_q, n_take = 6000, 160
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, n_q, seed=5, mu_q=mu, exact_frac=0.05)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    ntiles = (n_refs + 8191) // 8192
    runs = {}
    for name, opt in (("one", 0), ("two", 1), ("two/least", _rule(2, 0, 0, 0)), ("two/all", _rule(0, 0, 1024, 1024))):
        index = rx.Index(tree, debug_taps=True, two_level=opt)
        res = index.classify(qs.bases, qs.base_off, *index.exact_matches(qs.bases, qs.base_off))
        check_properties(res, db, n_q)
        st = index.debug_prune_stats()
        assert st["pairs"] == (n_q + 1) // 2 and st["bound_violations"] == 0, (name, st)
        last, taps = _taps(index, n_q, n_take)
        runs[name] = (res, st, last, taps)
        print(name, {k: st[k] for k in ("live_tiles_per_pair", "live_tiles_per_pair_first_stage", "queries_with_threshold", "mean_threshold") if k in st})
    res1, st1, last1, taps1 = runs["one"]
    L = db.length
    for name in ("two", "two/least", "two/all"):
        res2, st2, last2, taps2 = runs[name]
        assert np.array_equal(last1, last2)     # the processing order does not depend on the bounds
        sub = np.ascontiguousarray(qs.bases.reshape(-1, L)[last1]).reshape(-1)
        off = (np.arange(len(last1) + 1) * L).astype(np.uint64)
        t_o, counts_o = otree.hit_counts_batch(sub, off, skip_exact=False, threads=8)
        n_equal_best = n_all_equal = 0
        for k, q in enumerate(last1):
            ub1, det1 = taps1[int(q)]
            ub2, det2 = taps2[int(q)]
            tile_max = np.concatenate([counts_o[k], np.zeros((-n_refs) % 8192, counts_o.dtype)]).reshape(-1, 8192).max(axis=1)
            assert (ub2.astype(np.int64) >= tile_max).all(), (name, int(q))             # bounds are bounds
            assert (ub2 >= ub1).all(), (name, int(q))                                   # blocks of 256 are unions of blocks of 64
            T = det2["block"] // 128
            assert ub2[T] == ub1[T] and det2["largest_bound"] == ub2[T], (name, int(q))  # the best block's tile was refined: the bound of level B
            blk = counts_o[k][det2["block"] * 64:(det2["block"] + 1) * 64]
            assert det2["largest_bound"] >= det2["M"] == int(blk.max())
            # the exact counts of the best block, reference by reference (with the two-level pass they come from the block-major copy of the
            # database, with the one-level pass from the tile-major bitmap: rtx_prune.hip)
            assert np.array_equal(det2["block_counts"][:len(blk)], blk) and not det2["block_counts"][len(blk):].any(), (name, int(q))
            blk1 = counts_o[k][det1["block"] * 64:(det1["block"] + 1) * 64]
            assert np.array_equal(det1["block_counts"][:len(blk1)], blk1), int(q)
            n_equal_best += det1["block"] == det2["block"]
            n_all_equal += bool(np.array_equal(ub1, ub2))   # every tile at the one-level bound: a heavy query (the one-level pass took it), or one whose every B-tile was refined
            if name == "two/all":
                assert np.array_equal(ub1, ub2) and det1["block"] == det2["block"] and det1["threshold"] == det2["threshold"], int(q)
        assert n_equal_best >= 0.98 * len(last1), (name, n_equal_best)
        print(name, f"{n_all_equal} of {len(last1)} sampled queries carry the one-level bound on every tile")
        if name == "two" and ntiles >= 16:
            # queries 8 % from their best hit: most of them ask for many B-tiles and are handed to the one-level pass; at 2 % hardly any is
            assert (n_all_equal >= 0.3 * len(last1)) if mu >= 0.08 else (n_all_equal <= 0.5 * len(last1)), (mu, n_all_equal)
        # same rows as the one-level run (what is proven dead before counting never reaches the output either way)
        for f in ("row_off", "row_lineage", "t", "status"):
            assert np.array_equal(getattr(res2, f), getattr(res1, f)), (name, f)
        assert np.abs(res2.row_conf - res1.row_conf).max() < 1e-9 and np.abs(res2.global_signal - res1.global_signal).max() < 1e-9
    # the default rule refines a few groups and leaves the counting pass about what the one-level pass leaves it
    # (before the fine stage -- which only takes the pairs with four live tiles or more: the final numbers are not monotone in the bounds)
    fs = "live_tiles_per_pair_first_stage"
    assert runs["two/all"][1][fs] <= runs["two"][1][fs] + 1e-9 and runs["two"][1][fs] <= runs["two/least"][1][fs] + 1e-9
    assert runs["two/all"][1][fs] == st1[fs] and runs["two/all"][1]["live_tiles_per_pair"] == st1["live_tiles_per_pair"]
    if ntiles >= 8 and mu <= 0.03:
        assert runs["two"][1]["live_tiles_per_pair"] <= 1.25 * st1["live_tiles_per_pair"], (runs["two"][1], st1)


def test_best_block_at_the_ragged_end_of_the_database(oracle):
    """A database that ends 37 references into a tile of its own (16 tiles + 37: the last block of 64, the last B-tile, the last A-tile and the last
    tile are all partial) and queries drawn from those last references: their best block is the partial one.  Exact counts of the block
    (block-major copy of the database, rtx_prune.hip), bounds and rows against the oracle and against the one-level run."""
    n_refs, n_q = 16 * 8192 + 37, 1200
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, n_q, seed=9, mu_q=0.02, exact_frac=0.0)
    L = db.length
    refs = db.seq_bytes.reshape(n_refs, L)
    rng = np.random.default_rng(4)
    B = qs.bases.reshape(n_q, L)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    orig = tree.original_index()                           # position in the (lineage-sorted) database -> record of `db`
    for i in range(300):                                   # a quarter of the batch: relatives of the last 37 references of the database
        s = refs[int(orig[n_refs - 1 - (i % 37)])].copy()
        pos = rng.integers(0, L, size=12)
        s[pos] = s[rng.integers(0, L, size=12)]
        B[i] = s
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    last_block = (n_refs - 1) // 64
    runs = {}
    for name, opt in (("one", 0), ("two", 1)):
        index = rx.Index(tree, debug_taps=True, two_level=opt)
        res = index.classify(qs.bases, qs.base_off, *index.exact_matches(qs.bases, qs.base_off))
        check_properties(res, db, n_q)
        st = index.debug_prune_stats()
        assert st["pairs"] == n_q // 2 and st["bound_violations"] == 0 and st["queries_with_threshold"] > 0.9 * n_q, (name, st)
        sample = np.arange(0, 300, 3)
        taps = {}
        sub_last = set(int(q) for q in last_sub_batch_queries(index, n_q))
        for q in sample:
            if int(q) in sub_last:
                taps[int(q)] = (index.debug_tile_bounds(int(q)), index.debug_prune_detail(int(q)))
        runs[name] = (res, taps)
    res1, taps1 = runs["one"]
    res2, taps2 = runs["two"]
    assert len(taps2) >= 20 and set(taps1) == set(taps2)
    qlist = sorted(taps2)
    sub = np.ascontiguousarray(B[qlist]).reshape(-1)
    off = (np.arange(len(qlist) + 1) * L).astype(np.uint64)
    t_o, counts_o = otree.hit_counts_batch(sub, off, skip_exact=False, threads=8)
    n_last = 0
    for k, q in enumerate(qlist):
        for ub, det in (taps1[q], taps2[q]):
            blk = counts_o[k][det["block"] * 64:(det["block"] + 1) * 64]
            assert np.array_equal(det["block_counts"][:len(blk)], blk) and not det["block_counts"][len(blk):].any(), q
            tile_max = np.concatenate([counts_o[k], np.zeros((-n_refs) % 8192, counts_o.dtype)]).reshape(-1, 8192).max(axis=1)
            assert (ub.astype(np.int64) >= tile_max).all(), q
        n_last += taps2[q][1]["block"] == last_block
        assert taps1[q][1]["threshold"] > 0 and taps2[q][1]["threshold"] > 0
    assert n_last >= 10, n_last                             # the partial block is the best block of the queries made from it
    for f in ("row_off", "row_lineage", "t", "status"):
        assert np.array_equal(getattr(res2, f), getattr(res1, f)), f
    assert np.abs(res2.row_conf - res1.row_conf).max() < 1e-9
