"""BASELINE.json configs[1] at full database size (50 000 references) on the GPU: properties that do not depend on
the size (every query classified, invariance under the order and the composition of the batch, exact copies found)
plus oracle parity on a handful of queries."""
import numpy as np
import pytest

import raxtax_amd as rx
from raxtax_amd import synth

pytestmark = pytest.mark.gpu

N_REFS, N_Q = 50_000, 20_000


@pytest.fixture(scope="module")
def full():
    db = synth.make_db(N_REFS)
    qs = synth.make_queries(db, N_Q)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)   # bitmaps built on the GPU
    index = rx.Index(tree)
    ex_ids, ex_off = index.exact_matches(qs.bases, qs.base_off)
    res = index.classify(qs.bases, qs.base_off, ex_ids, ex_off)
    return dict(db=db, qs=qs, tree=tree, index=index, res=res, ex=(ex_ids, ex_off))


def _rows(res, q):
    a, b = int(res.row_off[q]), int(res.row_off[q + 1])
    return res.row_lineage[a:b], res.row_conf[a:b], res.row_local_signal[a:b]


def test_every_query_is_classified(full):
    res, qs = full["res"], full["qs"]
    assert res.n_queries == N_Q and (res.status == 0).all()
    assert (np.diff(res.row_off.astype(np.int64)) >= 1).all()
    L = full["db"].length
    assert (res.t <= L - 7).all() and (res.t >= 2).all()
    assert np.isfinite(res.global_signal).all() and (res.global_signal > 0).all()
    conf = res.row_conf
    assert (conf >= 0).all() and (conf <= 1.0 + 1e-12).all()
    depth = res.row_depth
    # confidences never increase from one level to the next (a child's range is inside its parent's)
    for d in range(1, 6):
        sel = depth > d
        assert (conf[sel, d] <= conf[sel, d - 1] + 1e-12).all()
    # rows of a query are sorted by descending confidence vectors (lineage.rs:91-93)
    first = res.row_off[:-1].astype(np.int64)
    nxt = first + 1
    two = nxt < res.row_off[1:].astype(np.int64)
    assert (conf[first[two], 0] >= conf[nxt[two], 0]).all()


def test_order_and_composition_of_the_batch_do_not_matter(full):
    qs, index, res = full["qs"], full["index"], full["res"]
    L = full["db"].length
    B = qs.bases.reshape(-1, L)
    # reversed input order
    rev = np.ascontiguousarray(B[::-1]).reshape(-1)
    ex_ids, ex_off = index.exact_matches(rev, qs.base_off)
    r2 = index.classify(rev, qs.base_off, ex_ids, ex_off)
    # a prefix of the batch on its own
    n1 = 3000
    off1 = qs.base_off[: n1 + 1]
    e1 = index.exact_matches(qs.bases[: n1 * L], off1)
    r3 = index.classify(qs.bases[: n1 * L], off1, *e1)
    for q in list(range(0, N_Q, 97)) + [N_Q - 1]:
        a, b = _rows(res, q), _rows(r2, N_Q - 1 - q)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), q
        assert res.global_signal[q] == r2.global_signal[N_Q - 1 - q] and res.t[q] == r2.t[N_Q - 1 - q]
    for q in range(0, n1, 53):
        a, b = _rows(res, q), _rows(r3, q)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), q


def test_exact_copies_find_their_reference(full):
    """A query that is a byte-identical copy of a reference has that reference among its exact matches
    (tree.sequences.get, raxtax.rs:42), a full-overlap count (t hits, prob.rs:24-41) and its best row in the
    lineage of a reference with the same sequence."""
    qs, tree, res, index = full["qs"], full["tree"], full["res"], full["index"]
    ex_ids, ex_off = full["ex"]
    orig = tree.original_index()
    inv = np.empty(len(orig), np.int64)
    inv[orig] = np.arange(len(orig))
    n_exact = np.diff(ex_off.astype(np.int64))
    copies = np.nonzero(n_exact > 0)[0]
    assert len(copies) > N_Q // 20          # 10 % of the synthetic queries are exact copies
    lineages = tree.lineages
    checked = 0
    for q in copies[:200]:
        ids = ex_ids[int(ex_off[q]):int(ex_off[q + 1])]
        assert inv[qs.source[q]] in ids
        top = int(res.row_lineage[int(res.row_off[q])])
        assert lineages[top] in {lineages[i] for i in ids} or n_exact[q] > 1
        checked += 1
    assert checked == min(200, len(copies))


def test_oracle_parity_on_a_sample(full, oracle):
    db, qs, res, index = full["db"], full["qs"], full["res"], full["index"]
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    sample = [0, 1, 4999, 12345, N_Q - 1]
    for q in sample:
        rows, _ = otree.classify(qs.seq(q), raw_confidence=True)
        got = res.rows(q)
        assert [r.lineage for r in got] == [r["idx"] for r in rows], q
        assert [r.confidence_values for r in got] == [r["conf"] for r in rows], q
        for g, r in zip(got, rows):
            assert abs(g.local_signal - r["local_signal"]) < 1e-6 and abs(g.global_signal - r["global_signal"]) < 1e-9
    # hit counts bit-exact (debug taps address the last sub-batch: classify the sample as a batch of its own)
    L = db.length
    sub = np.concatenate([qs.seq(q) for q in sample])
    off = (np.arange(len(sample) + 1) * L).astype(np.uint64)
    index.classify(sub, off, *index.exact_matches(sub, off))
    for j, q in enumerate(sample):
        t, counts = otree.hit_counts(qs.seq(q))
        assert np.array_equal(index.debug_hit_counts(j), counts) and index.debug_kmers(j).size == t
