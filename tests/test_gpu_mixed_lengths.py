"""Batches of mixed read length ("what's missing" 1): the reference classifies every query on one code path up to 65 535
k-mers (raxtax.rs:55-57); the library cuts a batch into LENGTH CLASSES (t <= 255 / t <= 1023 / longer reads whose probability arrays fit
LDS / up to t = 65 535) so that one long read does not move a file of barcodes off the pair kernel, the memoised tables and the tile
pruning -- and no read below the reference's own limit is refused.

  * 100 k COI reads + 10 reads of 1 100 .. 8 000 bases + a few hundred short reads in ONE batch: every long read and a sample of the others
    against the oracle (counts bit-exact through the recounting tap where the tap can reach, rows for all), the rows of the COI reads
    identical to those of a pure COI batch, and the COI class still on the fast path;
  * reads of tens of kilobases up to the reference's limit (t <= 65 535: the forms of hit_count and prob_table that keep their arrays in
    global memory), and the refusal one base beyond it;
  * the processing order switched off (input order inside every class)."""
import numpy as np
import pytest

import raxtax_amd as rx
from gpu_common import check_properties
from raxtax_amd import synth
from raxtax_amd.checks import assert_rows_equivalent

pytestmark = pytest.mark.gpu


def _long_read(rng, db, n_bases):
    """A chimera of references with substitutions: hits all over the database, every one of its k-mers a real one or a neighbour's."""
    L = db.length
    refs = db.seq_bytes.reshape(db.n, L)
    parts = []
    while sum(len(p) for p in parts) < n_bases:
        parts.append(refs[int(rng.integers(0, db.n))].copy())
    s = np.concatenate(parts)[:n_bases]
    hit = rng.random(n_bases) < 0.02
    s[hit] = (1 << rng.integers(0, 4, int(hit.sum()))).astype(np.uint8)
    return s


def _concat(seqs):
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return np.concatenate(seqs), off


def _check_against_oracle(res, otree, bases, off, ids, what):
    sub, soff = _concat([bases[int(off[q]):int(off[q + 1])] for q in ids])
    t_o, counts_o = otree.hit_counts_batch(sub, soff, threads=8)
    from oracle.oracle_py import Oracle
    tables_o, z_o, rc = Oracle().prob_tables_batch(t_o, counts_o, threads=8)
    bad, rows_o, nrows_o = otree.classify_batch(sub, soff, raw_confidence=True, threads=8, cap=512)   # (a chimera of a hundred references returns a hundred rows)
    assert bad == 0
    ties = 0
    for j, q in enumerate(ids):
        q = int(q)
        assert int(res.t[q]) == int(t_o[j]), f"{what}: query {q}: t = {int(res.t[q])}, oracle {int(t_o[j])}"
        want = otree.rows_of(rows_o, nrows_o, j, 512)
        got = res.rows(q)
        if [g.lineage for g in got] == [r["idx"] for r in want] and [g.confidence_values for g in got] == [r["conf"] for r in want]:
            for g, r in zip(got, want):
                assert abs(g.local_signal - r["local_signal"]) < 1e-6 and abs(g.global_signal - r["global_signal"]) < 1e-9, (what, q)
        else:
            ties += assert_rows_equivalent(got, want, tables_o[j][counts_o[j]], otree.lineages, f"{what}: query {q}") > 0
    return ties


def test_coi_reads_with_long_and_short_reads_in_one_batch(oracle):
    n_refs, n_coi = 50_000, 100_000
    db = synth.make_db(n_refs)
    qs = synth.make_queries(db, n_coi, seed=31)
    L = db.length
    rng = np.random.default_rng(32)
    coi = [qs.bases[i * L:(i + 1) * L] for i in range(n_coi)]
    long_lens = [1100, 1500, 1501, 2200, 3000, 4103, 4500, 6000, 7000, 8000]
    longs = [_long_read(rng, db, n) for n in long_lens]
    shorts = [coi[int(rng.integers(0, n_coi))][a:a + int(rng.integers(60, 262))] for a in rng.integers(0, 300, 300)]
    # the long and the short reads scattered through the barcodes
    seqs = list(coi)
    for k, s in enumerate(longs + shorts):
        at = int(rng.integers(0, len(seqs) + 1))
        seqs.insert(at, s)
    bases, off = _concat(seqs)
    lens = np.diff(off.astype(np.int64))
    n_all = len(seqs)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    index = rx.Index(tree)
    res = index.classify(bases, off)
    assert res.n_queries == n_all and (res.status == 0).all()
    classes = index.batch_classes()
    print("length classes of the mixed batch:", classes)
    # the barcodes with the few short reads | the ten long reads: a handful of reads of a few kilobases rides with the longer ones
    assert len(classes) == 2
    c_coi, c_huge = classes
    assert c_coi["queries"] == n_coi + len(shorts) and c_coi["planes"] == 10 and c_coi["tables"] and c_coi["pair"] and c_coi["prune"] and c_coi["records"]
    assert c_huge["queries"] == 10 and c_huge["planes"] == 16 and not c_huge["pair"] and c_huge["global_memory_forms"]
    assert index.debug_prune_stats()["pairs"] > 0
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    ids_long = np.nonzero(lens > 1030)[0]
    ids_short = np.nonzero(lens < 300)[0][:100]
    ids_coi = np.sort(rng.choice(np.nonzero(lens == L)[0], 300, replace=False))
    assert len(ids_long) == len(longs)
    ties = 0
    for ids, what in ((ids_long, "long reads"), (ids_short, "short reads"), (ids_coi, "COI reads")):
        ties += _check_against_oracle(res, otree, bases, off, ids, what)
    print(f"mixed batch: {len(ids_long)} long, {len(ids_short)} short, {len(ids_coi)} COI reads equal the oracle's rows ({ties} with a tie)")
    assert ties <= 3
    # the barcodes get what a batch of barcodes alone gives them -- rows, signals, everything
    pure = index.classify(qs.bases, qs.base_off)
    pure_classes = index.batch_classes()
    assert len(pure_classes) == 1 and pure_classes[0]["prune"]
    pos_coi = np.nonzero(lens == L)[0]
    assert len(pos_coi) >= n_coi
    # (a short read cut from a barcode can have COI length only if it is the whole barcode: none does)
    for j in rng.choice(n_coi, 5000, replace=False):
        # query j of the pure batch is the j-th COI-length read of the mixed one
        a, b = res.rows(int(pos_coi[j])), pure.rows(int(j))
        assert [x.lineage for x in a] == [x.lineage for x in b] and [x.confidence_values for x in a] == [x.confidence_values for x in b]
        assert a[0].global_signal == b[0].global_signal and [x.local_signal for x in a] == [x.local_signal for x in b]
    # the processing order switched off: input order inside every class, the same rows
    rx._lib.check(index._lib.rtx_index_set_option(index._h, 7, 0))
    plain = index.classify(bases, off)
    for q in list(ids_long) + list(ids_coi[:100]) + list(ids_short[:30]):
        a, b = plain.rows(int(q)), res.rows(int(q))
        assert [x.lineage for x in a] == [x.lineage for x in b] and [x.confidence_values for x in a] == [x.confidence_values for x in b], int(q)


def _de_bruijn_4_8():
    """B(4, 8) as encoded bases (1, 2, 4, 8), linearised: 4^8 + 7 bases that hold every 8-mer once (the standard Lyndon-word construction)."""
    k, n = 4, 8
    a = [0] * (k * n)
    seq = []

    def db(t, p):
        if t > n:
            if n % p == 0:
                seq.extend(a[1:p + 1])
        else:
            a[t] = a[t - p]
            db(t + 1, p)
            for j in range(a[t - p] + 1, k):
                a[t] = j
                db(t + 1, t)
    import sys
    sys.setrecursionlimit(10000)
    db(1, 1)
    seq = seq + seq[:n - 1]
    return (1 << np.array(seq, np.uint8)).astype(np.uint8)


def test_reads_up_to_the_reference_limit(oracle):
    """t <= 65 535 DISTINCT k-mers is the reference's limit (raxtax.rs:56 on utils.rs:27-40): reads of 20 kb .. 90 kb are served (global-memory
    forms of hit_count's histogram and of prob_table); only a read that holds every one of the 65 536 8-mers trips the reference's assert --
    the library reports that query (RTX_Q_ALL_KMERS) and classifies the rest of the batch."""
    db = synth.make_db(6000, fanouts=(2, 2, 3, 3, 3, 2))
    rng = np.random.default_rng(41)
    # three chimeras of references (many hits; their k-mers repeat, so t stays near 10 000) and one read of random bases with a
    # reference spliced in: ~ 41 000 distinct k-mers, most of them in no reference
    rnd = (1 << rng.integers(0, 4, 65_542)).astype(np.uint8)
    rnd[1000:1000 + db.length] = db.seq(123)
    seqs = [_long_read(rng, db, n) for n in (20_000, 33_001, 90_000)] + [rnd, db.seq(5).copy(), db.seq(77).copy()]
    bases, off = _concat(seqs)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    index = rx.Index(tree)
    res = index.classify(bases, off)
    classes = index.batch_classes()
    print("classes:", classes, "t:", res.t)
    assert (res.status == 0).all() and classes[-1]["global_memory_forms"] and classes[-1]["queries"] == 4 and int(res.t.max()) > 35_000
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    ties = _check_against_oracle(res, otree, bases, off, np.arange(len(seqs)), "reads up to the limit")
    assert ties <= 1
    # the full vectors of the last sub-batch (the longest class) through the recounting taps
    t_o, counts_o = otree.hit_counts_batch(bases, off, threads=8)
    tables_o, z_o, rc = oracle.prob_tables_batch(t_o, counts_o, threads=8)
    for q in range(4):
        assert np.array_equal(index.debug_hit_counts(q), counts_o[q]), f"hit counts of the {len(seqs[q])}-base read"
        tz, z = index.debug_prob_table(q, int(t_o[q]))
        present = np.bincount(counts_o[q], minlength=int(t_o[q]) + 1)[: int(t_o[q]) + 1] > 0
        d = float(np.max(np.abs(tz[present] - tables_o[q][: int(t_o[q]) + 1][present])))
        assert d < 1e-9, f"probabilities of the {len(seqs[q])}-base read differ by {d}"
    # a de Bruijn sequence B(4, 8): every 8-mer exactly once in 65 543 bases -- t = 65 536, the one case the reference's assert is about
    every = _de_bruijn_4_8()
    assert len(every) == 65_543
    b2, o2 = _concat([db.seq(1).copy(), every, db.seq(9).copy()])
    r2 = index.classify(b2, o2)
    assert list(r2.status) == [0, 2, 0] and int(r2.t[1]) == 65_536 and len(r2.rows(1)) == 0
    _check_against_oracle(r2, otree, b2, o2, np.array([0, 2]), "beside a read with every 8-mer")


def test_full_length_16s_reads_take_the_pruned_class(oracle, emul):
    """Reads of 1 031 .. 2 054 bases (t <= 2047: full-length 16S, the use SINTAX was written for) in numbers have a class of their own since
    round 6: eleven bit planes on the pair kernel (u16 counts), tile pruning, the memoised tables -- until then they fell to the
    one-query-per-wave kernel without pruning ("what's missing" 1).  The run AS IT WAS LEFT against the oracle: counts of the
    visited tiles bit-exact, no unvisited tile above the threshold, probabilities, rows -- with the records path, with the dense epilogues
    (RTX_OPT_RECORDS = 0), for reads 8 % from their source (tens of live tiles: the second stage of the bounds); and the rows of the
    same reads through the unpruned class they take when there are only a few of them."""
    from gpu_common import as_run_oracle_sample

    n_refs, n_q, L = 140_000, 6_000, 1_500          # 18 tiles: the fine union bitmap and the block-major copy exist
    db = synth.make_db(n_refs, length=L)
    qs = synth.make_queries(db, n_q, seed=21)
    qd = synth.make_queries(db, 1_500, seed=23, mu_q=0.08, exact_frac=0.0)
    rng = np.random.default_rng(22)
    seqs = []
    for i in range(n_q):
        s = qs.seq(i)
        seqs.append(s[: int(rng.integers(1040, L))].copy() if i % 3 == 0 else s.copy())   # a third of them cut to 1 040 .. 1 499 bases
    seqs += [_long_read(rng, db, 2054) for _ in range(24)]                                   # the longest reads of the class: t up to 2047
    seqs += [qd.seq(i).copy() for i in range(qd.n)]
    order = rng.permutation(len(seqs))
    seqs = [seqs[int(i)] for i in order]
    bases, off = _concat(seqs)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    index = rx.Index(tree, debug_taps=True)
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    runs = {}
    for records in (4, 0):
        rx._lib.check(index._lib.rtx_index_set_option(index._h, 18, records))
        res = index.classify(bases, off)
        assert (res.status == 0).all()
        classes = index.batch_classes()
        st = index.debug_prune_stats()
        print(f"RTX_OPT_RECORDS = {records}: classes:", classes, "pruning:", st)
        assert len(classes) == 1 and classes[0]["planes"] == 11 and classes[0]["tables"] and classes[0]["pair"] and classes[0]["prune"]
        assert int(res.t.max()) > 1500 and st["pairs"] > 0 and st["bound_violations"] == 0 and st["queries_with_threshold"] > 0.8 * len(seqs)
        assert st["live_tiles_per_pair"] < 9 and st["fine_blocks_per_pair"] > 0             # of 18; the divergent reads reach the second stage
        assert (st["record_queries"] > 0) == (records > 0)
        seen = as_run_oracle_sample(index, res, oracle, otree, bases, off, 300, False, threads=8, emul=emul)
        print("as the run left them:", seen)
        assert seen["n"] == 300 and seen["ties"] <= 3 and seen["with_threshold"] > 240 and seen["max_dp"] < 1e-9
        assert (seen["on_records_path"] > 0) == (records > 0)
        runs[records] = res
    for f in ("row_off", "row_lineage", "row_conf", "row_local_signal", "global_signal", "t", "status"):
        assert np.array_equal(getattr(runs[4], f), getattr(runs[0], f)), f                 # records or counts: the same rows
    res = runs[4]
    # a few hundred of the same reads alone: too few for the class -- one query per wave, no pruning, the recurrence kernel (round 5's path)
    ids = np.sort(rng.choice(len(seqs), 400, replace=False))
    b2, o2 = _concat([seqs[int(i)] for i in ids])
    few = index.classify(b2, o2)
    c2 = index.batch_classes()
    assert len(c2) == 1 and not c2[0]["pair"] and not c2[0]["prune"] and c2[0]["planes"] >= 12
    differ = 0
    for j, q in enumerate(ids):
        a, b = res.rows(int(q)), few.rows(j)
        same = [x.lineage for x in a] == [x.lineage for x in b] and [x.confidence_values for x in a] == [x.confidence_values for x in b]
        differ += not same
    print(f"{differ} of {len(ids)} reads print another row through the unpruned class")
    assert differ <= 4          # (exact ties / a confidence on a rounding boundary: pruned and unpruned probabilities differ by ~1e-11)
