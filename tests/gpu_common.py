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


def oracle_sample_parity(index, oracle, otree, db, qs, sample, skip, excuses, full_res=None, chunk=250, tol=1e-6):
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
