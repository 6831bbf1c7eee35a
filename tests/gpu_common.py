"""Shared pieces of the GPU parity tests: the documented excuses (exact ties between sibling taxa, confidences on a
rounding boundary) are COUNTED, printed and bounded by a committed expectation per test, and
the full-size checks (size-independent properties + a seeded oracle sample) are the same for every configuration."""
from __future__ import annotations

import json
import os
from pathlib import Path

import numpy as np

from raxtax_amd.checks import (as_run_oracle_sample, assert_rows_equivalent, check_properties, check_run_as_left,  # noqa: F401
                               emul_threshold, last_sub_batch_queries, rows_of)

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


def oracle_sample_parity(index, oracle, otree, db, qs, sample, skip, excuses, full_res=None, chunk=250, tol=1e-6, emul=None):
    """The seeded sample classified as a batch of its own and compared with the oracle, stage by stage:
    t and hit counts bit-exact (raxtax.rs:58-68), probabilities table[m]/Z within `tol` (north_star: 1e-6; asserted
    tighter below), result rows identical (ties counted in `excuses`).  If `full_res` is given the rows must also
    equal those the same queries got inside the full batch (composition and order of a batch never matter)."""
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
