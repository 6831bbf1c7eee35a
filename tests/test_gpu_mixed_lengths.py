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


def test_reads_up_to_the_reference_limit(oracle):
    """t <= 65 535 is the reference's limit (raxtax.rs:56): reads of 20 kb .. 65 542 bases are served (global-memory forms of hit_count's
    histogram and of prob_table), one base more is refused with RTX_ERR_TOO_LONG -- for the whole batch, as the reference asserts."""
    db = synth.make_db(6000, fanouts=(2, 2, 3, 3, 3, 2))
    rng = np.random.default_rng(41)
    # three chimeras of references (many hits; their k-mers repeat, so t stays near 10 000) and one read of random bases with a
    # reference spliced in: ~ 41 000 distinct k-mers, most of them in no reference
    rnd = (1 << rng.integers(0, 4, 65_542)).astype(np.uint8)
    rnd[1000:1000 + db.length] = db.seq(123)
    seqs = [_long_read(rng, db, n) for n in (20_000, 33_001, 65_542)] + [rnd, db.seq(5).copy(), db.seq(77).copy()]
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
    too_long = _long_read(rng, db, 65_543)
    b2, o2 = _concat([db.seq(1).copy(), too_long])
    with pytest.raises(rx.RtxError) as e:
        index.classify(b2, o2)
    assert e.value.code == -8, e.value
