"""Runs the device arithmetic of raxtax_amd/csrc/rtx_math.hpp (the inline functions the HIP
kernels call) on the CPU through a small emulation harness and checks it against the oracle:
bit-sliced counters must be bit-exact, the linear-domain pmf recurrence must reproduce the
reference's log-space probability table."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from raxtax_amd import synth

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("planes,n_rows,density", [(10, 648, 0.3), (10, 1016, 0.97), (12, 4088, 0.5), (16, 8000, 0.9)])
def test_bit_planes_exact(emul, planes, n_rows, density):
    rng = np.random.default_rng(planes * 1000 + n_rows)
    bits = rng.random((n_rows, 32)) < density
    rows = (bits.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    out = np.zeros(32, dtype=np.uint32)
    emul.emul_planes_count(rows.ctypes.data_as(C.c_void_p), n_rows, planes, out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, bits.sum(axis=0).astype(np.uint32))


@pytest.mark.parametrize("groups,n_rows,density", [(4, 1024, 0.3), (16, 1024, 0.9), (4, 128, 0.02), (16, 896, 0.5)])
def test_grouped_bit_planes_of_the_two_level_bounds(emul, groups, n_rows, density):
    """rtx_bounds2.hip: R lane groups fold different rows of the same columns, their bit-sliced partial counters are added pairwise
    (planes_add), the largest counter is read off the planes (planes_max): totals and maximum exact, the lowest counter among equals."""
    rng = np.random.default_rng(groups * 10000 + n_rows)
    bits = rng.random((n_rows, 32)) < density
    rows = (bits.astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    out = np.zeros(32, dtype=np.uint32)
    emul.emul_planes_grouped.restype = C.c_uint32
    key = emul.emul_planes_grouped(rows.ctypes.data_as(C.c_void_p), C.c_uint32(n_rows), C.c_uint32(groups), out.ctypes.data_as(C.c_void_p))
    want = bits.sum(axis=0).astype(np.uint32)
    assert np.array_equal(out, want)
    assert key >> 8 == int(want.max()) and key & 0xFF == int(np.argmax(want))


def _lnfact(oracle, n):
    return np.array([oracle.lib.orc_ln_factorial(i) for i in range(n)], dtype=np.float64)


def _emul_table(emul, lf, t, sizes, fn="emul_prob_table"):
    hist = np.bincount(sizes, minlength=t + 1).astype(np.uint32)
    tz = np.zeros(t + 1)
    z, gs = C.c_double(), C.c_double()
    stats = np.zeros(4, dtype=np.uint64)
    f = getattr(emul, fn)
    f.restype = C.c_int
    rc = f(C.c_uint32(t), hist.ctypes.data_as(C.c_void_p), C.c_uint64(len(sizes)),
                              lf.ctypes.data_as(C.c_void_p), tz.ctypes.data_as(C.c_void_p), C.byref(z), C.byref(gs),
                              stats.ctypes.data_as(C.c_void_p))
    _emul_table.last_stats = stats
    return rc, tz, z.value, gs.value


def test_prob_recurrence_matches_reference_p2(emul, oracle, kats):
    k = kats["P2_hit_prob"]
    t = k["t"]
    sizes = np.arange(0, t + 1, dtype=np.uint16)
    lf = _lnfact(oracle, 2 * t + 8)
    rc, tz, z, gs = _emul_table(emul, lf, t, sizes)
    assert rc == 0
    ref = oracle.highest_hit_prob_per_reference(t, t // 2, sizes)   # any count == t -> only_last branch
    assert np.max(np.abs(tz[sizes] - ref)) < 1e-12
    # general branch: drop the full-overlap reference
    sizes2 = np.arange(0, t, dtype=np.uint16)
    rc, tz, z, gs = _emul_table(emul, lf, t, sizes2)
    ref = oracle.highest_hit_prob_per_reference(t, t // 2, sizes2)
    assert rc == 0 and z >= 1.0
    assert np.max(np.abs(tz[sizes2] - ref)) < 1e-12
    assert abs(ref.sum() - 1.0) < 1e-9


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_prob_recurrence_on_realistic_counts(emul, oracle, seed):
    db = synth.make_db(1200, fanouts=(2, 2, 3, 3, 3, 2))
    ot = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    qs = synth.make_queries(db, 6, seed=10 + seed, exact_frac=0.3)
    lf = _lnfact(oracle, 2048)
    for i in range(qs.n):
        for skip in (False, True):
            t, counts = ot.hit_counts(qs.seq(i), skip_exact=skip)
            rc, tz, z, gs = _emul_table(emul, lf, t, counts)
            ref = oracle.highest_hit_prob_per_reference(t, t // 2, counts)
            assert rc == 0
            assert np.max(np.abs(tz[counts] - ref)) < 1e-11, (i, skip)
            gs_ref = np.sqrt(((ref - 1.0 / len(ref)) ** 2).sum())
            assert abs(gs - gs_ref) < 1e-11


def test_prob_recurrence_long_sequences_need_scaling(emul, oracle):
    # t = 3000: ln pmf_m(0) is far below ln(DBL_MIN); exercises the 2^-512 rescaling path
    t = 3000
    rng = np.random.default_rng(5)
    sizes = np.concatenate([rng.integers(200, 700, 400), rng.integers(2500, 2950, 5), [0, 1, 2990]]).astype(np.uint16)
    lf = _lnfact(oracle, 2 * t + 8)
    rc, tz, z, gs = _emul_table(emul, lf, t, sizes)
    ref = oracle.highest_hit_prob_per_reference(t, t // 2, sizes)
    assert rc == 0
    assert np.max(np.abs(tz[sizes] - ref)) < 1e-11


def test_prob_status_cases(emul, oracle):
    lf = _lnfact(oracle, 64)
    assert _emul_table(emul, lf, 0, np.zeros(4, np.uint16))[0] == 1          # t == 0
    assert _emul_table(emul, lf, 1, np.zeros(4, np.uint16))[0] == 1          # t == 1, no full overlap
    rc, tz, z, gs = _emul_table(emul, lf, 1, np.array([0, 1, 1, 0], np.uint16))  # t == 1 with full overlap
    assert rc == 0 and z == 2.0 and tz[1] == 0.5 and tz[0] == 0.0


@pytest.mark.parametrize("seed", [0, 1])
def test_prob_lookup_tables_match_reference(emul, oracle, seed):
    """The memoised-table formulation (rtx_prob_tables.hip) against the reference's log-space table."""
    db = synth.make_db(1200, fanouts=(2, 2, 3, 3, 3, 2))
    ot = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    qs = synth.make_queries(db, 6, seed=20 + seed, exact_frac=0.3)
    lf = _lnfact(oracle, 2048)
    for i in range(qs.n):
        for skip in (False, True):
            t, counts = ot.hit_counts(qs.seq(i), skip_exact=skip)
            rc, tz, z, gs = _emul_table(emul, lf, t, counts, fn="emul_prob_lookup")
            ref = oracle.highest_hit_prob_per_reference(t, t // 2, counts)
            assert rc == 0
            assert np.max(np.abs(tz[counts] - ref)) < 1e-11, (i, skip)
    # P2 of the reference's tests, general branch
    t = 400
    sizes = np.arange(0, t, dtype=np.uint16)
    rc, tz, z, gs = _emul_table(emul, _lnfact(oracle, 2 * t + 8), t, sizes, fn="emul_prob_lookup")
    ref = oracle.highest_hit_prob_per_reference(t, t // 2, sizes)
    assert rc == 0 and np.max(np.abs(tz[sizes] - ref)) < 1e-12
    # degenerate inputs
    assert _emul_table(emul, lf, 0, np.zeros(4, np.uint16), fn="emul_prob_lookup")[0] == 1
    rc, tz, z, gs = _emul_table(emul, lf, 9, np.zeros(7, np.uint16), fn="emul_prob_lookup")   # no hits at all
    assert rc == 0 and abs(tz[0] - 1.0 / 7) < 1e-15


@pytest.mark.parametrize("n_refs", [1, 100, 8192, 8193, 50000, 3 * 8192])
def test_bitmap_bit_layout_is_a_bijection(emul, n_refs):
    """ref_slot (where a reference sits in its row) is a bijection onto the bits hit_count's lanes own, and the
    reference of (tile, lane, group, j) -- the order of a lane's count stores -- is its inverse."""
    stride = ((n_refs + 7) // 8 + 127) // 128 * 128
    seen = set()
    w, b = C.c_uint32(), C.c_uint32()
    emul.emul_slot_ref.restype = C.c_uint32
    for r in list(range(min(n_refs, 3000))) + list(range(max(0, n_refs - 3000), n_refs)):
        emul.emul_ref_slot(r, stride, C.byref(w), C.byref(b))
        assert w.value * 4 < stride and b.value < 32
        pos = (w.value, b.value)
        assert pos not in seen or r < 3000 and r >= n_refs - 3000
        seen.add(pos)
        tile, in_tile = divmod(w.value, 256)
        lane, word = divmod(in_tile, 4)
        g, j = word * 4 + b.value // 8, b.value % 8
        assert emul.emul_slot_ref(tile, lane, g, j, stride) == r


def test_epilogue_byte_merge_and_transpose(emul):
    rng = np.random.default_rng(9)
    for _ in range(50):
        sb = rng.integers(0, 256, 8).astype(np.uint8)
        cnt = rng.integers(0, 60000, 8).astype(np.uint16)
        st = cnt.view(np.uint32).copy()
        emul.emul_merge_bytes(sb.view(np.uint32).ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p))
        assert np.array_equal(st.view(np.uint16), cnt + sb)
    m = rng.integers(0, 2, (64, 64)).astype(np.uint64)
    rows = (m << np.arange(64, dtype=np.uint64)[None, :]).sum(axis=1).astype(np.uint64)
    cols = np.zeros(64, np.uint64)
    emul.emul_transpose64(rows.ctypes.data_as(C.c_void_p), cols.ctypes.data_as(C.c_void_p))
    want = (m.T << np.arange(64, dtype=np.uint64)[None, :]).sum(axis=1).astype(np.uint64)
    assert np.array_equal(cols, want)


def test_row_finalisation_matches_the_oracle(emul, oracle):
    """finalise_kernel's arithmetic (rtx_math.hpp: fin_row_before, fin_local_signal) on the CPU: the rows the oracle returns for real
    barcodes (lineage.rs:91-110: sorted, with their local signal), shuffled, must come back in the oracle's order with the oracle's local
    signals; rows handed over in the oracle's order must keep it (the sort of lineage.rs:91-93 is stable)."""
    from pathlib import Path

    text = (Path(__file__).resolve().parent / "golden" / "diptera_subset.fasta").read_text()
    otree = oracle.parse_reference_fasta_str(text)
    queries = oracle.parse_query_fasta_str(text)
    N = otree.num_tips
    rng = np.random.default_rng(5)
    emul.emul_finalise_rows.restype = C.c_uint32
    multi = 0
    for label, seq in queries[:150]:
        rows, _ = otree.classify(seq, skip_exact=True, raw_confidence=True)
        n = len(rows)
        D = max(len(r["conf"]) for r in rows)
        k = np.zeros((n, D), np.uint8)
        size = np.zeros((n, D), np.uint32)
        depth = np.zeros(n, np.uint32)
        for i, r in enumerate(rows):
            depth[i] = len(r["conf"])
            k[i, :depth[i]] = np.rint(np.array(r["conf"]) * 100).astype(np.uint8)
            size[i, :depth[i]] = np.rint(np.array(r["expd"]) * N).astype(np.uint32)
        for order in (np.arange(n), rng.permutation(n)):
            ks, ss, ds = np.ascontiguousarray(k[order]), np.ascontiguousarray(size[order]), np.ascontiguousarray(depth[order])
            rank = np.zeros(n, np.uint32)
            local = np.zeros(n, np.float64)
            bad = emul.emul_finalise_rows(C.c_uint32(n), C.c_uint32(D), ks.ctypes.data_as(C.c_void_p), ds.ctypes.data_as(C.c_void_p),
                                          ss.ctypes.data_as(C.c_void_p), C.c_double(N), rank.ctypes.data_as(C.c_void_p), local.ctypes.data_as(C.c_void_p))
            assert bad == 0                                           # the kernel's word-wise compare = the byte-wise statement of lineage.rs:91-93
            assert sorted(rank.tolist()) == list(range(n))            # a permutation: no two rows share a place
            back = np.empty(n, np.int64)
            back[rank] = np.arange(n)
            assert np.array_equal(ks[back], k) and np.array_equal(ds[back], depth)      # the oracle's order of confidence vectors
            want = np.array([r["local_signal"] for r in rows])
            if np.array_equal(order, np.arange(n)):
                assert np.array_equal(rank, np.arange(n))             # stable: equal vectors keep the order of the walk
                assert np.allclose(local, want, rtol=0, atol=1e-12)
            else:   # (rows with equal confidence vectors may have changed places among themselves: compare them as a set)
                key = lambda kk, dd, ll: sorted((tuple(a), int(b), round(float(c), 11)) for a, b, c in zip(kk.tolist(), dd, ll))
                assert key(ks, ds, local) == key(k, depth, want)
        multi += n > 1
    assert multi > 20
