"""a3 on the device (`tree.sequences.get(seq)`, raxtax.rs:42; consumers raxtax.rs:65-68 zeroing, :73-84 override): a
batch uploaded without exact-match ids has them looked up by rtx_exact.hip -- a hash table of the distinct reference sequences, every
candidate verified byte by byte.  The ids must be those of the host map (rtx_tree_exact_matches_batch, itself checked against the
oracle in tests/test_host_logic.py), and every result must equal the one of a run that was handed the host's ids."""
import numpy as np
import pytest

import raxtax_amd as rx
from raxtax_amd import synth

pytestmark = pytest.mark.gpu


def _cases(db, rng, n):
    """Queries around the edges of `equal as byte strings`: copies, copies of duplicated references, one base changed, one base
    more / fewer, an ambiguity code in place of a base, a reference that itself carries ambiguity codes, unrelated, too short."""
    L = db.length
    qs = []
    for i in range(n):
        r = int(rng.integers(0, db.n)) if i % 16 else (60, 7, 75, 2005)[(i // 16) % 4]   # every sixteenth: a duplicated reference
        s = db.seq(r).copy()
        k = i % 8 if i % 16 else 0
        if k == 1:
            s[int(rng.integers(0, L))] ^= 3                      # A<->C / ... : another valid code or an ambiguity code
        elif k == 2:
            s = s[:-1]
        elif k == 3:
            s = np.concatenate([s, s[:1]])
        elif k == 4:
            s[int(rng.integers(0, L))] = 15                      # N
        elif k == 5:
            s = (1 << rng.integers(0, 4, L)).astype(np.uint8)    # unrelated
        elif k == 6:
            s = s[:5]                                            # no k-mer at all
        qs.append(s)
    off = np.zeros(len(qs) + 1, np.uint64)
    off[1:] = np.cumsum([len(q) for q in qs])
    return np.concatenate(qs), off


@pytest.mark.parametrize("weak_hash", [False, True])
def test_device_lookup_equals_the_host_map(oracle, weak_hash):
    db = synth.make_db(3000, fanouts=(2, 2, 3, 3, 3, 2))
    # duplicates: identical sequences under the same and under other lineages; one reference with ambiguity codes (and a copy of it)
    flat = db.seq_bytes.reshape(db.n, db.length).copy()
    flat[100:140] = flat[60:100]
    flat[2000:2010] = flat[60]
    flat[7, 10:13] = 15
    flat[2500] = flat[7]
    flat = flat.reshape(-1)
    tree = rx.Tree.new_flat(db.lineages, flat, db.seq_off, kmer_map=False)
    lib = rx._lib.load()
    if weak_hash:        # 8 x 8 values of (slot, tag): chains of dozens of groups, every probe ends in the byte compare
        rx._lib.check(lib.rtx_set_default_option(2, 0xE000000000000007))
    try:
        index = rx.Index(tree)
    finally:
        rx._lib.check(lib.rtx_set_default_option(2, 0))
    assert index.has_exact_lookup
    db2 = synth.SynthDB(db.lineages, flat, db.seq_off, db.length)
    bases, off = _cases(db2, np.random.default_rng(5), 480)
    # the references in the tree's (lineage-sorted) order are what the ids index: copies of sorted references as well
    ids_h, off_h = tree.exact_matches_batch(bases, off)
    n_ex = np.diff(off_h.astype(np.int64))
    assert n_ex.max() == 12 and (n_ex == 2).sum() >= 8 and (n_ex == 1).sum() >= 40 and (n_ex == 0).sum() >= 300   # 1 + 1 + 10 copies of reference 60
    for skip in (False, True):
        want = index.classify(bases, off, ids_h, off_h, skip_exact_matches=skip)       # host ids handed in
        got = index.classify(bases, off, skip_exact_matches=skip)                        # looked up on the device
        ids_d, off_d = index.device_exact_matches()
        assert np.array_equal(off_d, off_h) and np.array_equal(ids_d, ids_h), skip
        for f in ("row_off", "row_lineage", "row_conf", "row_local_signal", "global_signal", "t", "status"):
            assert np.array_equal(getattr(got, f), getattr(want, f)), (skip, f)
    # the oracle's map agrees (Tree::new, tree.rs:109-112)
    otree = oracle.tree_new_flat(db.lineages, flat, db.seq_off)
    for q in range(0, 480, 7):
        s = bases[int(off[q]):int(off[q + 1])]
        assert np.array_equal(np.sort(otree.exact_matches(s)), ids_h[int(off_h[q]):int(off_h[q + 1])])
    # a handle told not to look up: a batch without ids has no exact matches (ABI version 2 behaviour)
    plain = rx.Index(tree, device_exact=False)
    assert not plain.has_exact_lookup
    none = plain.classify(bases, off, skip_exact_matches=True)
    with_ids = plain.classify(bases, off, ids_h, off_h, skip_exact_matches=True)
    assert not np.array_equal(none.row_conf, with_ids.row_conf) or not np.array_equal(none.row_lineage, with_ids.row_lineage)


def test_host_mirror_uses_the_device_lookup(oracle):
    """rtx_raxtax (raxtax.rs:14-97) with a handle that looks exact matches up itself: text identical to the oracle's, including the
    single-exact-match override (raxtax.rs:73-84) and --skip-exact-matches."""
    db = synth.make_db(1500, fanouts=(2, 2, 3, 3, 3, 2))
    qs = synth.make_queries(db, 120, exact_frac=0.4)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    index = rx.Index(tree)
    assert index.has_exact_lookup
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    queries = [(qs.labels[q], qs.seq(q).copy()) for q in range(qs.n)]
    for skip, raw in ((False, False), (True, False), (False, True)):
        got = {}
        rx.raxtax(queries, index, skip, raw, 50, lambda l, o, t: got.__setitem__(l, o), False)
        n_diff = 0
        for label, seq in queries:
            rows, rawrows = otree.classify(seq, skip_exact=skip, raw_confidence=raw)
            n_diff += got[label] != otree.format_out(label, rawrows)
        assert n_diff <= 2, (skip, raw, n_diff)      # exact ties between sibling taxa at most


@pytest.mark.parametrize("chunk", [0, 37, 64])
def test_several_handles_in_one_call(oracle, chunk):
    """rtx_raxtax_multi: one call, several device handles, each driven by a thread of its own inside the library (the header's promise
    that distinct handles may be driven from distinct host threads, exercised) -- here three handles on the one GPU of the box, one of
    them without the device lookup (its exact matches come from the host map).  The messages arrive in input order and are those of a
    single handle."""
    db = synth.make_db(3000, fanouts=(2, 2, 3, 3, 3, 2))
    qs = synth.make_queries(db, 300, exact_frac=0.3)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    handles = [rx.Index(tree), rx.Index(tree, sub_batch=50), rx.Index(tree, device_exact=False)]
    queries = [(qs.labels[q], qs.seq(q).copy()) for q in range(qs.n)]
    for skip, tsv in ((False, True), (True, False)):
        one, many = [], []
        rx.raxtax(queries, handles[0], skip, False, chunk, lambda l, o, t: one.append((l, o, t)), tsv)
        rx.raxtax(queries, handles, skip, False, chunk, lambda l, o, t: many.append((l, o, t)), tsv)
        assert [m[0] for m in many] == qs.labels and many == one, (skip, chunk)
    # a closed channel stops the run (raxtax.rs:87)
    seen = []

    def closing(l, o, t):
        seen.append(l)
        if len(seen) == 70:
            raise BrokenPipeError("sink closed")
    with pytest.raises(BrokenPipeError):
        rx.raxtax(queries, handles, False, False, 32, closing, False)
    assert seen == qs.labels[:70]
