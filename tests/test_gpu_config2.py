"""BASELINE.json configs[2] on the GPU: 500 000 references (the database size of the headline and of the 8-GPU
configs[3]), 100 000 of the synthetic queries of the bench through the size-independent properties, and a seeded
1 000-query sample against the CPU oracle in both exact-match modes -- hit counts bit-exact, probabilities within
1e-6, result rows identical (the reference's own methodology samples databases of this size,
scripts/runtime_memory.py:42-43)."""
import numpy as np
import pytest

import raxtax_amd as rx
from gpu_common import Excuses, as_run_oracle_sample, check_properties, oracle_sample_parity, rows_of
from raxtax_amd import synth

pytestmark = pytest.mark.gpu

N_REFS, N_Q, N_SAMPLE = 500_000, 100_000, 1_000   # (2 000 until round 6: the suite has to stay well inside the driver's time limit; every bench line checks 2 000 of its own)


@pytest.fixture(scope="module")
def cfg2(oracle):
    db = synth.make_db(N_REFS)
    qs = synth.make_queries(db, N_Q)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)   # bitmaps built on the GPU
    index = rx.Index(tree, debug_taps=True)      # prune_kernel keeps its view of every query for rtx_debug_prune_detail
    ex_ids, ex_off = index.exact_matches(qs.bases, qs.base_off)
    res = index.classify(qs.bases, qs.base_off, ex_ids, ex_off)
    prune_stats = index.debug_prune_stats()
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    sample = np.sort(np.random.default_rng(20260).choice(N_Q, N_SAMPLE, replace=False))
    return dict(db=db, qs=qs, tree=tree, index=index, res=res, prune_stats=prune_stats, ex=(ex_ids, ex_off), otree=otree, sample=sample)


def test_every_query_is_classified(cfg2):
    check_properties(cfg2["res"], cfg2["db"], N_Q)
    assert cfg2["index"].n_refs == N_REFS


def test_batch_order_does_not_matter(cfg2):
    qs, index, res = cfg2["qs"], cfg2["index"], cfg2["res"]
    L = cfg2["db"].length
    rev = np.ascontiguousarray(qs.bases.reshape(-1, L)[::-1]).reshape(-1)
    r2 = index.classify(rev, qs.base_off, *index.exact_matches(rev, qs.base_off))
    for q in list(range(0, N_Q, 331)) + [N_Q - 1]:
        a, b = rows_of(res, q), rows_of(r2, N_Q - 1 - q)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), q
        assert res.global_signal[q] == r2.global_signal[N_Q - 1 - q] and res.t[q] == r2.t[N_Q - 1 - q]


def test_exact_copies_find_their_reference(cfg2):
    qs, tree, res = cfg2["qs"], cfg2["tree"], cfg2["res"]
    ex_ids, ex_off = cfg2["ex"]
    orig = tree.original_index()
    inv = np.empty(len(orig), np.int64)
    inv[orig] = np.arange(len(orig))
    n_exact = np.diff(ex_off.astype(np.int64))
    copies = np.nonzero(n_exact > 0)[0]
    assert len(copies) > N_Q // 20          # 10 % of the synthetic queries are exact copies
    for q in copies[:300]:
        ids = ex_ids[int(ex_off[q]):int(ex_off[q + 1])]
        assert inv[qs.source[q]] in ids
        assert res.t[q] >= 2


def test_exact_matches_looked_up_on_the_device(cfg2):
    """a3 (Tree.sequences.get, raxtax.rs:42) on the device for all 100 000 queries: the ids are those of the host map, and the
    result of the batch is the one the host's ids gave (both exact-match modes: the zeroing of raxtax.rs:65-68 reads them)."""
    qs, index, res = cfg2["qs"], cfg2["index"], cfg2["res"]
    ex_ids, ex_off = cfg2["ex"]
    assert index.has_exact_lookup
    r2 = index.classify(qs.bases, qs.base_off)                       # no ids: the device looks them up
    ids_d, off_d = index.device_exact_matches()
    assert np.array_equal(off_d, ex_off) and np.array_equal(ids_d, ex_ids)
    for f in ("row_off", "row_lineage", "row_conf", "row_local_signal", "global_signal", "t", "status"):
        assert np.array_equal(getattr(r2, f), getattr(res, f)), f
    a = index.classify(qs.bases, qs.base_off, skip_exact_matches=True)
    b = index.classify(qs.bases, qs.base_off, ex_ids, ex_off, skip_exact_matches=True)
    for f in ("row_off", "row_lineage", "row_conf", "global_signal"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


@pytest.mark.parametrize("skip", [False, True])
def test_seeded_oracle_sample(cfg2, oracle, emul, skip):
    """The sample as a batch of its own through the default (pruned) handle: first the run exactly as it was -- the counts of the
    visited tiles, the unvisited tiles against the threshold, the histogram, the probabilities of the pruned run, prune_kernel's
    threshold against its CPU restatement (gpu_common.check_run_as_left) -- then the recounting taps and the result rows."""
    c = cfg2
    ex = Excuses(f"config2/sample{N_SAMPLE}/skip={int(skip)}")
    oracle_sample_parity(c["index"], oracle, c["otree"], c["db"], c["qs"], c["sample"], skip, ex,
                         full_res=None if skip else c["res"], emul=emul)
    ex.check()


def test_tile_pruning_is_at_work_and_changes_no_row(cfg2, oracle):
    """The fixture's handle prunes (RTX_OPT_TILE_PRUNE is the default: hit_count visits only the tiles that can matter,
    rtx_prune.hip) and no bound was violated; the same sample through a handle that counts EVERY tile: rows identical to the
    oracle's, and the rows of the pruned big batch equal them (confidences within 1e-9)."""
    c = cfg2
    st = c["prune_stats"]
    print("tile pruning of the full batch:", st)
    assert st["pairs"] > 0 and st["bound_violations"] == 0 and st["live_tiles_per_pair"] < 0.5 * ((N_REFS + 8191) // 8192), st
    index = rx.Index(c["tree"], tile_prune=False)
    ex = Excuses(f"config2/sample{N_SAMPLE}/no_tile_prune")
    res = oracle_sample_parity(index, oracle, c["otree"], c["db"], c["qs"], c["sample"], False, ex, full_res=None)
    assert index.debug_prune_stats()["pairs"] == 0
    for j in range(0, N_SAMPLE, 3):
        x, y = rows_of(res, j), rows_of(c["res"], int(c["sample"][j]))
        assert np.array_equal(x[0], y[0]) and np.allclose(x[1], y[1], rtol=0, atol=1e-9), int(c["sample"][j])
    ex.check()


def test_one_million_queries_as_the_bench_runs_them(cfg2, oracle, emul):
    """BASELINE.json configs[2] AT ITS SIZE: the 1 M queries of bench.py's rank 0 against the 500 000 references in one
    batch -- 31 sub-batches, exact matches looked up on the device -- through the size-independent properties, and a seeded 2 000-query
    sample taken from INSIDE that batch's last sub-batch held against the oracle exactly as the run left it (no recount, nothing run
    again: counts of the visited tiles bit-exact, unvisited tiles below the threshold, histogram, probabilities, prune_kernel's
    threshold against its CPU restatement), then the rows the batch returned for those queries against the oracle's rows."""
    import os

    c = cfg2
    qs = synth.make_queries(c["db"], 1_000_000, seed=3)
    index = c["index"]
    res = index.classify(qs.bases, qs.base_off)
    check_properties(res, c["db"], qs.n)
    assert index.sub_batch_size() * 2 < qs.n                      # really many sub-batches
    st = index.debug_prune_stats()
    assert st["pairs"] == qs.n // 2 and st["bound_violations"] == 0, st
    seen = as_run_oracle_sample(index, res, oracle, c["otree"], qs.bases, qs.base_off, 2000, False, threads=os.cpu_count() or 1, emul=emul)
    print("1 M queries, sample of the last sub-batch as the run left it:", seen)
    assert seen["n"] == 2000 and seen["with_threshold"] == 2000 and seen["max_dp"] < 1e-9
    ex = Excuses("config2/1M/last_sub_batch")
    ex.checked = seen["n"]
    ex.n["ties"] = seen["ties"]
    ex.check()
