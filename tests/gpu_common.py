"""Shared pieces of the GPU parity tests: the documented excuses (exact ties between sibling taxa, confidences on a
rounding boundary) are COUNTED, printed and bounded by a committed expectation per test, and
the full-size checks (size-independent properties + a seeded oracle sample) are the same for every configuration."""
from __future__ import annotations

import json
import os
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
EXPECTED = ROOT / "tests" / "golden" / "expected_excuses.json"


class Excuses:
    """Ledger of the queries a test excused.  check() prints the counts and fails if any of them exceeds the committed
    expectation of this test id (tests/golden/expected_excuses.json; an id that is not listed expects 0).  The device
    path is deterministic, so the expectations are the counts of a recorded run, not loose bounds: a regression that
    turns real mismatches into "ties" shows up as a count above its expectation.  RTX_RECORD_EXCUSES=<file> appends
    the observed counts there (how the expectations were produced)."""

    def __init__(self, test_id: str):
        self.id = test_id
        self.n = {"ties": 0, "boundary": 0}
        self.checked = 0

    def tie(self, k=1):
        self.n["ties"] += int(bool(k))

    def boundary(self, k=1):
        self.n["boundary"] += int(k)

    def check(self):
        exp = json.loads(EXPECTED.read_text()).get(self.id, {}) if EXPECTED.exists() else {}
        print(f"EXCUSED {self.id}: {self.n['ties']} queries with an exact tie, {self.n['boundary']} on a rounding boundary, "
              f"of {self.checked} checked (expected at most {exp.get('ties', 0)} / {exp.get('boundary', 0)})")
        rec = os.environ.get("RTX_RECORD_EXCUSES")
        if rec:
            with open(rec, "a") as f:
                f.write(json.dumps({self.id: dict(self.n, checked=self.checked)}) + "\n")
            return
        for k, v in self.n.items():
            assert v <= exp.get(k, 0), f"{self.id}: {v} queries excused as {k}, expectation {exp.get(k, 0)}"


def check_properties(res, db, n_q):
    """What holds for every query whatever the size of the database."""
    assert res.n_queries == n_q and (res.status == 0).all()
    assert (np.diff(res.row_off.astype(np.int64)) >= 1).all()
    L = db.length
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
    # the confidences of the rows of a query at one level cannot sum to more than 1 (+ rounding of each)
    nrows = np.diff(res.row_off.astype(np.int64))
    top = np.add.reduceat(conf[:, 5] if conf.shape[1] > 5 else conf[:, 0], first)
    assert (top <= 1.0 + 0.005 * nrows + 1e-9).all()


def rows_of(res, q):
    a, b = int(res.row_off[q]), int(res.row_off[q + 1])
    return res.row_lineage[a:b], res.row_conf[a:b], res.row_local_signal[a:b]


def emul_threshold(emul, lf, t, n_refs, block_counts, tab_tmax=1023):
    """(u, i* + 1) of rtx_emul.cpp's restatement of prune_kernel's step 3."""
    import ctypes as C

    hm = np.zeros(64, np.uint32)
    hm[: len(block_counts)] = block_counts
    u, i1 = C.c_uint32(), C.c_uint32()
    emul.emul_prune_threshold(C.c_uint32(t), C.c_uint64(n_refs), hm.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p),
                              C.c_uint32(tab_tmax), C.byref(u), C.byref(i1))
    return int(u.value), int(i1.value)


def check_run_as_left(index, j, t, counts_o, p_o, n_refs, emul=None, lf=None, label="", tol=1e-9):
    """What the PRUNED run itself computed for query j of the last sub-batch (no recount: rtx_debug_run_counts,
    rtx_debug_pruned_prob_table, rtx_debug_prune_detail) against the oracle's full computation (counts_o: raxtax.rs:58-68, p_o =
    table / Z: prob.rs:8-103):
      * the counts hit_count wrote for the tiles it visited are the oracle's, bit for bit;
      * every tile it did not visit holds no count above the query's threshold (and the query has one);
      * the histogram it left = the oracle's counts of the visited tiles above the threshold + everything else (unvisited references,
        counts up to the threshold) in bin 0;
      * the probabilities of the pruned run equal the oracle's above the threshold (`tol`: 1e-9 by default, north_star allows 1e-6)
        and what the oracle gives the references at or below it -- which the pruned run sets to 0 -- is below 1e-9 in total;
      * (debug_taps) prune_kernel's best block holds the oracle's counts, its bound holds, and its threshold is the one the CPU
        restatement derives from those counts (whose safety tests/test_prune_threshold_cpu.py attacks).
    Returns a dict of what was seen."""
    rc = index.debug_run_counts(j, t)
    live, thr = rc["tile_live"], rc["threshold"]
    ntiles = len(live)
    pad = ntiles * 8192 - n_refs
    co = np.concatenate([counts_o, np.zeros(pad, np.uint16)]).reshape(ntiles, 8192)
    cr = np.concatenate([rc["counts"], np.zeros(pad, np.uint16)]).reshape(ntiles, 8192)
    assert np.array_equal(cr[live], co[live]), f"{label}: counts of the visited tiles differ from the oracle"
    tile_max_o = co.max(axis=1)
    if not live.all():
        assert thr > 0, f"{label}: tiles left out for a query without a threshold"
        assert int(tile_max_o[~live].max()) <= thr, f"{label}: an unvisited tile holds a count above the threshold {thr}"
    in_tile = np.minimum(8192, n_refs - np.arange(ntiles) * 8192)
    want_hist = np.bincount(co[live].reshape(-1), minlength=t + 1)[: t + 1].astype(np.int64)
    want_hist[0] += int(in_tile[~live].sum()) - int(pad if live[-1] else 0)      # the padding of the last tile is no reference
    if thr:      # the epilogue of a pruned query puts the counts up to its threshold into bin 0 as one number (they are references without a hit to prob.rs)
        want_hist[0] += int(want_hist[1: thr + 1].sum())
        want_hist[1: thr + 1] = 0
    assert np.array_equal(rc["hist"].astype(np.int64), want_hist), f"{label}: histogram as the run left it"
    tz_p, z_p, thr2 = index.debug_pruned_prob_table(j, t)
    assert thr2 == thr
    hist_o = np.bincount(counts_o, minlength=t + 1)[: t + 1]
    above = (np.arange(t + 1) > thr) & (hist_o > 0) if thr else hist_o > 0
    d = float(np.max(np.abs(tz_p[above] - p_o[: t + 1][above]), initial=0.0))
    assert d < tol, f"{label}: probabilities of the pruned run differ by {d}"
    dropped = float((hist_o * p_o[: t + 1])[: thr + 1].sum()) if thr else 0.0
    assert dropped < 1e-9, f"{label}: the references up to the threshold {thr} hold {dropped} in the oracle"
    if thr:
        assert (tz_p[: thr + 1] == 0).all()
    out = dict(threshold=thr, live=int(live.sum()), needed=int((tile_max_o > thr).sum()) if thr else ntiles, dp=d, dropped=dropped)
    if emul is not None:
        det = index.debug_prune_detail(j)
        b = det["block"]
        blk = np.zeros(64, np.uint32)
        seg = counts_o[b * 64:(b + 1) * 64]
        blk[: len(seg)] = seg
        assert np.array_equal(det["block_counts"], blk), f"{label}: exact counts of the best block {b}"
        assert det["M"] == int(blk.max()) and det["t"] == t and det["threshold"] == thr
        assert det["largest_bound"] >= int(counts_o.max()), f"{label}: the largest bound lies below a count"
        u_e, i1_e = emul_threshold(emul, lf, t, n_refs, blk)
        assert (u_e, i1_e) == (thr, rc["i1"]), f"{label}: kernel threshold {(thr, rc['i1'])}, CPU restatement {(u_e, i1_e)}"
    return out


def oracle_sample_parity(index, oracle, otree, db, qs, sample, skip, excuses, full_res=None, chunk=250, tol=1e-6, emul=None):
    """The seeded sample classified as a batch of its own and compared with the oracle, stage by stage:
    t and hit counts bit-exact (raxtax.rs:58-68), probabilities table[m]/Z within `tol` (north_star: 1e-6; asserted
    tighter below), result rows identical (ties counted in `excuses`).  If `full_res` is given the rows must also
    equal those the same queries got inside the full batch (composition and order of a batch never matter)."""
    from test_gpu_parity import assert_rows_equivalent

    L = db.length
    threads = os.cpu_count() or 1
    B = qs.bases.reshape(-1, L)
    sub = np.ascontiguousarray(B[sample]).reshape(-1)
    off = (np.arange(len(sample) + 1) * L).astype(np.uint64)
    ex = index.exact_matches(sub, off)
    res = index.classify(sub, off, *ex, skip_exact_matches=skip)
    assert (res.status == 0).all()
    lineages = None
    worst = 0.0
    if index.debug_prune_stats()["pairs"] > 0:
        # first the run exactly as it was (the taps below recount the sub-batch in full and would wipe it)
        lf = np.array([oracle.lib.orc_ln_factorial(i) for i in range(2 * int(res.t.max()) + 8)], dtype=np.float64) if emul is not None else None
        seen = dict(n=0, thr=0, live=0, needed=0, dp=0.0, dropped=0.0, with_thr=0)
        for a in range(0, len(sample), chunk):
            b = min(len(sample), a + chunk)
            t_o, counts_o = otree.hit_counts_batch(sub[a * L:b * L], off[a:b + 1] - off[a], skip_exact=skip, threads=threads)
            tables_o, z_o, rc = oracle.prob_tables_batch(t_o, counts_o, threads=threads)
            assert (rc == 0).all()
            for j in range(a, b):
                o = check_run_as_left(index, j, int(t_o[j - a]), counts_o[j - a], tables_o[j - a], index.n_refs, emul, lf,
                                      f"query {int(sample[j])} skip {skip}")
                seen["n"] += 1
                seen["with_thr"] += o["threshold"] > 0
                seen["thr"] += o["threshold"]
                seen["live"] += o["live"]
                seen["needed"] += o["needed"]
                seen["dp"] = max(seen["dp"], o["dp"])
                seen["dropped"] = max(seen["dropped"], o["dropped"])
        print(f"pruned run as it was: {seen['n']} queries, {seen['with_thr']} with a threshold (mean {seen['thr'] / max(seen['n'], 1):.1f}), "
              f"{seen['live'] / max(seen['n'], 1):.2f} tiles visited per query ({seen['needed'] / max(seen['n'], 1):.2f} hold a count above the threshold), "
              f"visited counts bit-exact, max |p - p_oracle| = {seen['dp']:.2e}, most the oracle gives a dropped set = {seen['dropped']:.2e}")
    for a in range(0, len(sample), chunk):
        b = min(len(sample), a + chunk)
        t_o, counts_o = otree.hit_counts_batch(sub[a * L:b * L], off[a:b + 1] - off[a], skip_exact=skip, threads=threads)
        tables_o, z_o, rc = oracle.prob_tables_batch(t_o, counts_o, threads=threads)
        assert (rc == 0).all()
        bad, rows_o, nrows_o = otree.classify_batch(sub[a * L:b * L], off[a:b + 1] - off[a], skip_exact=skip, raw_confidence=True,
                                                    threads=threads, cap=64)
        assert bad == 0
        for j in range(a, b):
            t = int(t_o[j - a])
            assert res.t[j] == t, (skip, int(sample[j]))
            assert np.array_equal(index.debug_hit_counts(j), counts_o[j - a]), f"hit counts differ: query {int(sample[j])}, skip {skip}"
            tz, z = index.debug_prob_table(j, t)
            present = np.bincount(counts_o[j - a], minlength=t + 1)[: t + 1] > 0
            d = float(np.max(np.abs(tz[present] - tables_o[j - a][: t + 1][present])))
            worst = max(worst, d)
            assert d < min(tol, 1e-9), f"probabilities differ by {d}: query {int(sample[j])}, skip {skip}"
            # Z itself is not compared: the device drops factors cmf^h = 1 + O(h 1e-16) that are common to every
            # table entry (they cancel in table / Z, which is what is checked above and what reaches the output)
            want = otree.rows_of(rows_o, nrows_o, j - a, 64)
            got = res.rows(j)
            excuses.checked += 1
            if [g.lineage for g in got] != [r["idx"] for r in want] or [g.confidence_values for g in got] != [r["conf"] for r in want]:
                if lineages is None:
                    lineages = otree.lineages
                probs = tables_o[j - a][counts_o[j - a]]
                excuses.tie(assert_rows_equivalent(got, want, probs, lineages, f"query {int(sample[j])} skip {skip}"))
            else:
                for g, r in zip(got, want):
                    assert abs(g.local_signal - r["local_signal"]) < 1e-6 and abs(g.global_signal - r["global_signal"]) < 1e-9
            if full_res is not None:
                x, y = rows_of(res, j), rows_of(full_res, int(sample[j]))
                assert all(np.array_equal(p, q) for p, q in zip(x, y)), int(sample[j])
                assert res.global_signal[j] == full_res.global_signal[int(sample[j])]
    print(f"oracle sample: {len(sample)} queries, skip={skip}: counts bit-exact, max |p - p_oracle| = {worst:.3e}")
    return res
