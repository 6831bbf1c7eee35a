"""The threshold of the tile pruning (raxtax_amd/csrc/rtx_prune.hip, step 3 of prune_kernel) attacked on the CPU.

`emul_prune_threshold` (rtx_emul.cpp) restates the kernel's inequalities with the same tables; the GPU tests hold the kernel's
(u, i* + 1) against it query by query (tests/test_gpu_pruned_path.py).  Here the threshold it yields is held against the oracle
(prob.rs:8-103 restated) on the histograms that stress its proof: the bound is N x tail, so the adversary puts the best block H
at the top and EVERY other reference of the database exactly at the threshold u (or spreads them just below it), for databases
of up to 5 M references.  Treating the counts up to u as references without a hit -- what prob_lookup does for a pruned query,
emulated by `emul_prob_lookup_pruned` -- must leave every probability, and every sum of probabilities over any set of
references (the prefix sums of lineage.rs:61-66), within 1e-9 of the oracle's full computation; north_star allows 1e-6.

The tile-aware criterion of a whole-database handle ("(4)" in rtx_prune.hip, `emul_prune_threshold_tiles`) knows the largest bound of
every tile of 8192 references; there the adversary fills every dead tile with references exactly AT its bound and every live tile with
references at the threshold (or just above it)."""
import ctypes as C

import numpy as np
import pytest

TOL = 1e-9           # proved: a few eps, eps = 1e-10 (rtx_math.hpp: kPruneEpsHD; round 3 ran with 1e-12 and asserted 1e-11)
TAB_TMAX = 1023


def _lnfact(oracle, n):
    return np.array([oracle.lib.orc_ln_factorial(i) for i in range(n)], dtype=np.float64)


def threshold(emul, lf, t, n_refs, block_counts):
    hm = np.zeros(64, np.uint32)
    hm[: len(block_counts)] = block_counts
    u, i1 = C.c_uint32(), C.c_uint32()
    emul.emul_prune_threshold(C.c_uint32(t), C.c_uint64(n_refs), hm.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p),
                              C.c_uint32(TAB_TMAX), C.byref(u), C.byref(i1))
    return int(u.value), int(i1.value)


def threshold_tiles(emul, lf, t, n_refs, block_counts, tile_ub):
    hm = np.zeros(64, np.uint32)
    hm[: len(block_counts)] = block_counts
    ub = np.ascontiguousarray(tile_ub, dtype=np.uint16)
    u, i1 = C.c_uint32(), C.c_uint32()
    emul.emul_prune_threshold_tiles(C.c_uint32(t), C.c_uint64(n_refs), hm.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p),
                                    C.c_uint32(TAB_TMAX), C.c_uint32(len(ub)), ub.ctypes.data_as(C.c_void_p), C.byref(u), C.byref(i1))
    return int(u.value), int(i1.value)


def pruned_table(emul, lf, t, n_refs, hist, u, i1):
    tz = np.zeros(t + 1)
    z, gs = C.c_double(), C.c_double()
    emul.emul_prob_lookup_pruned.restype = C.c_int
    rc = emul.emul_prob_lookup_pruned(C.c_uint32(t), hist.ctypes.data_as(C.c_void_p), C.c_uint64(n_refs), lf.ctypes.data_as(C.c_void_p),
                                      C.c_uint32(u), C.c_uint32(i1), tz.ctypes.data_as(C.c_void_p), C.byref(z), C.byref(gs))
    assert rc == 0
    return tz, z.value, gs.value


def check_histogram(emul, oracle, lf, t, counts, block, label, thr=None):
    """counts: u16 [N] with the block's references among them.  Returns (u, worst error)."""
    n_refs = len(counts)
    u, i1 = thr if thr is not None else threshold(emul, lf, t, n_refs, block)
    if u == 0:
        return 0, 0.0
    hist = np.bincount(counts, minlength=t + 1).astype(np.uint32)
    tz_o, z_o, rc = oracle.prob_tables_batch(np.array([t], np.uint32), counts[None, :])
    assert rc[0] == 0
    p_o = tz_o[0][: t + 1]                                  # table[m] / Z of the full computation
    p_d, z_d, gs_d = pruned_table(emul, lf, t, n_refs, hist, u, i1)
    h = hist.astype(np.float64)
    assert (p_d[1: u + 1] == 0).all() and p_d[0] == 0.0, label
    kept = np.arange(t + 1) > u
    present = hist > 0
    per_ref = float(np.max(np.abs(p_d - p_o)[kept & present], initial=0.0))
    # any set of references: the worst set takes every reference whose error has one sign
    d = h * (p_d - p_o)
    any_set = max(float(d[d > 0].sum()), float(-d[d < 0].sum()))
    dropped = float((h * p_o)[: u + 1].sum())                # what the oracle gives the references that are dropped
    assert per_ref < TOL and any_set < TOL and dropped < TOL, (label, u, i1, per_ref, any_set, dropped)
    # the global signal (lineage.rs:86-90) follows from the same table
    gs_o = float(np.sqrt((h * (p_o - 1.0 / n_refs) ** 2).sum()))
    assert abs(gs_d - gs_o) < 1e-9, (label, gs_d, gs_o)
    return u, max(per_ref, any_set, dropped)


CASES = [  # t, N, best hit as a share of t, size of H
    (640, 70_000, 0.90, 1), (640, 500_000, 0.90, 5), (640, 500_000, 0.92, 40), (640, 5_000_000, 0.90, 12),
    (640, 5_000_000, 0.75, 3), (195, 500_000, 0.90, 8), (195, 5_000_000, 0.97, 30), (300, 500_000, 0.85, 2),
    (900, 500_000, 0.93, 20), (1023, 5_000_000, 0.9, 6), (64, 100_000, 0.95, 4), (640, 500_000, 0.5, 10),
]


@pytest.mark.parametrize("t,n_refs,best,n_h", CASES)
def test_everything_else_exactly_at_the_threshold(emul, oracle, t, n_refs, best, n_h):
    lf = _lnfact(oracle, 2 * t + 8)
    rng = np.random.default_rng(t * 31 + n_h)
    M = int(best * t)
    block = np.sort(rng.integers(int(0.8 * M) + 1, M + 1, n_h).astype(np.uint32))[::-1].copy()
    block[0] = M
    u, _ = threshold(emul, lf, t, n_refs, block)
    if u == 0:
        pytest.skip(f"no threshold for t={t} M={M}: nothing is pruned")
    worst = 0.0
    # (1) H at the top, everything else EXACTLY at u; (2) ... one below; (3) half at u, half without a hit;
    # (4) a crowd just ABOVE the threshold (kept) on top of the crowd at it; (5) everything else spread over [0, u]; (6), (7) below
    for k, rest in enumerate((lambda n: np.full(n, u), lambda n: np.full(n, max(u - 1, 0)),
                              lambda n: np.where(np.arange(n) % 2 == 0, u, 0),
                              lambda n: np.where(np.arange(n) % 50 == 0, np.minimum(u + 1 + (np.arange(n) // 50) % 7, int(0.8 * M)), u),
                              lambda n: rng.integers(0, u + 1, n),
                              # (6) half of the database just above the threshold (kept: they carry the density F' of Z'), half at it
                              lambda n: np.where(np.arange(n) % 2 == 0, np.minimum(u + 1 + np.arange(n) % 3, int(0.8 * M)), u),
                              # (7) a second family as good as H elsewhere in the database (kept), the rest at the threshold
                              lambda n: np.where(np.arange(n) < 200, M - (np.arange(n) % 9), u))):
        counts = np.concatenate([block, rest(n_refs - n_h)]).astype(np.uint16)
        u2, err = check_histogram(emul, oracle, lf, t, counts, block, f"t={t} N={n_refs} M={M} |H|={n_h} pattern {k}")
        assert u2 == u
        worst = max(worst, err)
    print(f"t={t} N={n_refs} M={M} |H|={n_h}: u={u}, worst error {worst:.2e}")


@pytest.mark.parametrize("t,n_refs", [(640, 500_000), (195, 5_000_000), (900, 70_000)])
def test_full_overlap_reference(emul, oracle, t, n_refs):
    """A reference shares every k-mer with the query (an exact copy): prob.rs:24-41 applies, table[m] = pmf_m(n)."""
    lf = _lnfact(oracle, 2 * t + 8)
    block = np.array([t, t - 3, t - 40], np.uint32)
    u, i1 = threshold(emul, lf, t, n_refs, block)
    assert u > 0 and i1 == 0
    for rest in (np.full(n_refs - 3, u), np.full(n_refs - 3, u // 2)):
        counts = np.concatenate([block, rest]).astype(np.uint16)
        check_histogram(emul, oracle, lf, t, counts, block, f"full overlap t={t} N={n_refs}")


def test_the_threshold_is_not_vacuous(emul, oracle):
    """One count ABOVE the threshold, at the same adversarial histogram, may break the budget: the criterion is not slack by
    orders of magnitude everywhere (printed, not asserted: how much room the proof leaves)."""
    t, n_refs = 640, 500_000
    lf = _lnfact(oracle, 2 * t + 8)
    block = np.array([580, 575, 560, 541, 530], np.uint32)
    u, i1 = threshold(emul, lf, t, n_refs, block)
    assert 200 < u < 464
    for du in (0, 10, 25, 50):
        counts = np.concatenate([block, np.full(n_refs - 5, u + du)]).astype(np.uint16)
        tz_o, z_o, rc = oracle.prob_tables_batch(np.array([t], np.uint32), counts[None, :])
        mass = float(tz_o[0][u + du] * (n_refs - 5))
        print(f"everything else at u + {du} = {u + du}: the oracle gives those references {mass:.3e} together")


@pytest.mark.parametrize("seed", range(6))
def test_random_cases(emul, oracle, seed):
    """Random t, database size, best hit and best block (ten per seed) against the two hardest patterns."""
    rng = np.random.default_rng(1000 + seed)
    n_thr = 0
    for _ in range(10):
        t = int(rng.integers(40, 1024))
        n_refs = int(rng.choice([20_000, 200_000, 2_000_000]))
        M = int(rng.uniform(0.35, 0.99) * t)
        n_h = int(rng.choice([1, 2, 7, 30, 64]))
        lf = _lnfact(oracle, 2 * t + 8)
        block = np.sort(rng.integers(int(0.8 * M) + 1, M + 1, n_h).astype(np.uint32))[::-1].copy()
        block[0] = M
        u, _ = threshold(emul, lf, t, n_refs, block)
        if u == 0:
            continue
        n_thr += 1
        for rest in (lambda n: np.full(n, u), lambda n: np.where(np.arange(n) % 2 == 0, np.minimum(u + 1 + np.arange(n) % 3, max(int(0.8 * M), u)), u)):
            counts = np.concatenate([block, rest(n_refs - n_h)]).astype(np.uint16)
            check_histogram(emul, oracle, lf, t, counts, block, f"t={t} N={n_refs} M={M} |H|={n_h}")
    assert n_thr >= 5


TILE_CASES = [  # t, tiles, best hit as a share of t, size of H, tiles whose bound lies near the threshold
    (640, 62, 0.90, 5, 3), (640, 62, 0.75, 12, 6), (640, 62, 0.55, 8, 20), (640, 62, 0.43, 8, 40), (640, 611, 0.9, 12, 9),
    (195, 14, 0.93, 20, 6), (195, 62, 0.7, 6, 30), (900, 30, 0.8, 3, 4), (64, 12, 0.95, 4, 2), (640, 8, 0.6, 2, 8),
]


@pytest.mark.parametrize("t,ntiles,best,n_h,n_near", TILE_CASES)
def test_tile_aware_threshold_against_tiles_filled_to_their_bounds(emul, oracle, t, ntiles, best, n_h, n_near):
    """Criterion "(4)": the threshold from the best block AND the largest bound of every tile.  The adversary respects the bounds and
    nothing else: every reference of a dead tile sits exactly at the tile's bound, every reference of a live tile at the threshold
    (or a third of them just above it: kept, they carry the density of the pruned Z); far tiles, tiles just below the threshold, tiles
    just above it.  The threshold must lie at or above the one of criterion (3) and keep every error within the budget."""
    rng = np.random.default_rng(t * 7 + ntiles + n_h)
    n_refs = ntiles * 8192 - 1234
    lf = _lnfact(oracle, 2 * t + 8)
    M = int(best * t)
    block = np.sort(rng.integers(int(0.8 * M) + 1, M + 1, n_h).astype(np.uint32))[::-1].copy()
    block[0] = M
    u3, _ = threshold(emul, lf, t, n_refs, block)
    if u3 == 0:
        pytest.skip(f"no threshold for t={t} M={M}")
    worst, gains = 0.0, []
    for spread in (4, 25, 80):       # how far the near tiles' bounds lie from the threshold of (3)
        ub = rng.integers(min(20, u3), max(int(0.55 * u3), 21), ntiles)                    # far tiles: unrelated clades
        near = rng.choice(np.arange(1, ntiles), min(n_near, ntiles - 1), replace=False)
        ub[near] = np.clip(u3 + rng.integers(-spread, spread + 1, len(near)), 1, int(0.8 * M))
        ub[0] = min(M + 40, t)                                                             # the tile of the best block
        u, i1 = threshold_tiles(emul, lf, t, n_refs, block, ub)
        assert u >= u3, (u, u3)
        gains.append(u - u3)
        size = np.minimum(8192, n_refs - np.arange(ntiles) * 8192)
        for pattern in range(3):
            counts = np.empty(n_refs, np.uint16)
            for T in range(ntiles):
                lo, hi = T * 8192, T * 8192 + int(size[T])
                if ub[T] <= u:
                    counts[lo:hi] = ub[T]                                                  # dead: everything at the bound
                elif pattern == 0:
                    counts[lo:hi] = u                                                      # live: everything at the threshold
                elif pattern == 1:
                    counts[lo:hi] = np.where(np.arange(hi - lo) % 3 == 0, min(u + 1 + T % 5, int(ub[T]), int(0.8 * M)), u)
                else:
                    counts[lo:hi] = rng.integers(0, min(u, int(ub[T])) + 1, hi - lo)
            counts[: len(block)] = block                                                   # H lives in tile 0
            u2, err = check_histogram(emul, oracle, lf, t, counts, block, f"t={t} tiles={ntiles} M={M} |H|={n_h} spread {spread} pattern {pattern}",
                                      thr=(u, i1))
            worst = max(worst, err)
    print(f"t={t} tiles={ntiles} M={M} |H|={n_h}: criterion (3) u={u3}, tile-aware +{gains} counts, worst error {worst:.2e}")
