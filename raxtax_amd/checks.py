"""Checks of a classified batch that the GPU tests AND bench.py's self-verification share (SURVEY.md 8d asks for
parity on every run).  Nothing here imports the oracle: the functions take what the checker computed (hit counts of raxtax.rs:58-68,
table / Z of prob.rs:8-103, result rows of lineage.rs:80-112) as arguments and hold the device's results against it, so the product
package stays free of the oracle while `tests/` and the `cpu_baseline` leg of `bench.py` -- the places that may use it -- pass it in.

  check_properties      what holds for every query whatever the size of the database (no oracle at all)
  check_run_as_left     one query of the last sub-batch exactly as the pruned, timed run left it, against the oracle's full computation
  assert_rows_equivalent  result rows identical, or an exact tie between sibling taxa verified from the oracle's probabilities
  as_run_oracle_sample  a seeded sample of the queries of the LAST sub-batch of a run through the three checks above
"""
from __future__ import annotations

import numpy as np


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



def emul_threshold(emul, lf, t, n_refs, block_counts, tab_tmax=2047, tile_ub=None):
    """(u, i* + 1) of rtx_emul.cpp's restatement of prune_kernel's step 3: criterion (3) from the best block alone (reference shards),
    or -- given the largest bound of every tile -- the tile-aware criterion (4) of a whole-database handle."""
    import ctypes as C

    hm = np.zeros(64, np.uint32)
    hm[: len(block_counts)] = block_counts
    u, i1 = C.c_uint32(), C.c_uint32()
    if tile_ub is None:
        emul.emul_prune_threshold(C.c_uint32(t), C.c_uint64(n_refs), hm.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p),
                                  C.c_uint32(tab_tmax), C.byref(u), C.byref(i1))
    else:
        ub = np.ascontiguousarray(tile_ub, dtype=np.uint16)
        emul.emul_prune_threshold_tiles(C.c_uint32(t), C.c_uint64(n_refs), hm.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p),
                                        C.c_uint32(tab_tmax), C.c_uint32(len(ub)), ub.ctypes.data_as(C.c_void_p), C.byref(u), C.byref(i1))
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
    if rc.get("record_segments", 0):
        # the records path (RTX_OPT_RECORDS): the run wrote (reference, count) records of the counts ABOVE the threshold and nothing else --
        # they must be exactly the oracle's counts above the threshold in the visited tiles (every other reference reads 0)
        assert thr > 0, f"{label}: records without a threshold"
        assert np.array_equal(cr[live], np.where(co[live] > thr, co[live], 0)), f"{label}: the records of the visited tiles are not the oracle's counts above the threshold {thr}"
    else:
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
    out = dict(threshold=thr, live=int(live.sum()), needed=int((tile_max_o > thr).sum()) if thr else ntiles, dp=d, dropped=dropped,
               records=bool(rc.get("record_segments", 0)))
    if emul is not None:
        det = index.debug_prune_detail(j)
        b = det["block"]
        blk = np.zeros(64, np.uint32)
        seg = counts_o[b * 64:(b + 1) * 64]
        blk[: len(seg)] = seg
        assert np.array_equal(det["block_counts"], blk), f"{label}: exact counts of the best block {b}"
        assert det["M"] == int(blk.max()) and det["t"] == t and det["threshold"] == thr
        tile_ub = index.debug_tile_bounds(j) if getattr(index, "tile_aware_threshold", True) else None
        # the bound of the best block covers its own references; every count lies at or below the largest bound of a tile (with the
        # two-level bounds pass the best block comes from the refined tiles only: the largest count may sit in another one)
        assert det["largest_bound"] >= int(blk.max()), f"{label}: the best block's bound lies below one of its counts"
        assert max(det["largest_bound"], int(tile_ub.max()) if tile_ub is not None else 0) >= int(counts_o.max()), f"{label}: the largest bound lies below a count"
        if tile_ub is not None:   # the bounds are bounds: every tile's largest count lies at or below its bound
            assert (tile_ub.astype(np.int64) >= tile_max_o).all(), f"{label}: a tile bound lies below a count of the tile"
        u_e, i1_e = emul_threshold(emul, lf, t, n_refs, blk, tile_ub=tile_ub)
        assert (u_e, i1_e) == (thr, rc["i1"]), f"{label}: kernel threshold {(thr, rc['i1'])}, CPU restatement {(u_e, i1_e)}"
    return out



def _path_confidences(lineages, probs, idx):
    """Unrounded confidence of every ancestor of reference `idx` (sum of probs over the references that
    share the first d+1 lineage levels), computed from the oracle's probabilities."""
    parts = lineages[idx].split(",")
    out = []
    lo = hi = idx
    for d in range(len(parts)):
        pre = ",".join(parts[: d + 1])
        is_in = lambda s: s == pre or s.startswith(pre + ",")
        a = idx
        while a > 0 and is_in(lineages[a - 1]):
            a -= 1
        b = idx + 1
        while b < len(lineages) and is_in(lineages[b]):
            b += 1
        out.append(float(probs[a:b].sum()))
    return out



def assert_rows_equivalent(got, rows, probs_ref, lineages, label=""):
    """Rows must be identical, except that where the reference breaks an exact tie between sibling
    taxa by floating-point noise in its prefix sums (lineage.rs:62-66,158-166: arg-max of equal
    confidences; the stable sort of equal confidence vectors, lineage.rs:91-93) the device may pick the
    other sibling.  A differing lineage is accepted only if its confidences equal those of the oracle's choice
    to 1e-9 (computed from the ORACLE's probabilities) on every level down to the one where the two lineages part."""
    assert len(got) == len(rows), label
    if [g.lineage for g in got] == [r["idx"] for r in rows]:
        for g, r in zip(got, rows):
            assert g.confidence_values == r["conf"], label
        return 0
    remaining = list(rows)
    n_ties = 0
    for g in got:
        match = None
        for r in remaining:
            if r["conf"] != g.confidence_values:
                continue
            if r["idx"] == g.lineage:
                match = r
                break
            a = _path_confidences(lineages, probs_ref, g.lineage)
            b = _path_confidences(lineages, probs_ref, r["idx"])
            # the two lineages part at level `fork`: a tie there (equal confidences up to and including that level)
            # explains every difference below it (the walk continues inside the sibling it chose)
            la, lb = lineages[g.lineage].split(","), lineages[r["idx"]].split(",")
            fork = next((d for d in range(min(len(la), len(lb))) if la[d] != lb[d]), min(len(la), len(lb)) - 1)
            if len(a) == len(b) and max(abs(x - y) for x, y in zip(a[: fork + 1], b[: fork + 1])) < 1e-9:
                match = r
                n_ties += 1
                break
        assert match is not None, f"{label}: device row {g} has no equivalent oracle row"
        remaining.remove(match)
        if match["idx"] == g.lineage:   # a tied sibling may have another size, hence another expected vector / local signal
            assert abs(g.local_signal - match["local_signal"]) < 1e-6, label
    return n_ties


def last_sub_batch_queries(index, n_q: int) -> np.ndarray:
    """The queries (input numbering) that the last run processed in its last sub-batch: the only ones whose scratch (counts, histogram,
    live masks, probability table) is still on the device when the run is over -- what the as-run taps can read."""
    order = index.debug_order(n_q)
    last0, n = index.last_sub_batch()     # (length classes of different sub-batch sizes follow one another: the library knows where the last one starts)
    return np.sort(order[last0:last0 + n].astype(np.int64))


def as_run_oracle_sample(index, res, oracle, otree, bases, base_off, n_sample: int, skip: bool, seed: int = 20264, chunk: int = 250,
                         threads: int = 1, emul=None, tol: float = 1e-9):
    """A seeded sample of the queries of the last sub-batch of the run that has just been downloaded (`res`: its Result), each held
    against the oracle WITHOUT running anything again: check_run_as_left (visited counts bit-exact, unvisited tiles below the threshold,
    histogram, probabilities), then the rows the run returned for the query against the oracle's rows (ties verified and counted).
    Raises AssertionError at the first violation; returns what was seen."""
    n_q = len(base_off) - 1
    cand = last_sub_batch_queries(index, n_q)
    rng = np.random.default_rng(seed)
    sample = np.sort(rng.choice(cand, min(n_sample, len(cand)), replace=False))
    lf = None
    if emul is not None:
        lf = np.array([oracle.lib.orc_ln_factorial(i) for i in range(2 * int(res.t.max()) + 8)], dtype=np.float64)
    seen = dict(n=0, with_threshold=0, thr=0, live=0, needed=0, max_dp=0.0, max_dropped=0.0, ties=0, rows_identical=0, on_records_path=0)
    lineages = None
    pruned = index.debug_prune_stats()["pairs"] > 0
    for a in range(0, len(sample), chunk):
        ids = sample[a:a + chunk]
        sub = np.concatenate([bases[int(base_off[q]):int(base_off[q + 1])] for q in ids])
        off = np.zeros(len(ids) + 1, np.uint64)
        off[1:] = np.cumsum([int(base_off[q + 1] - base_off[q]) for q in ids])
        t_o, counts_o = otree.hit_counts_batch(sub, off, skip_exact=skip, threads=threads)
        tables_o, z_o, rc = oracle.prob_tables_batch(t_o, counts_o, threads=threads)
        assert (rc == 0).all()
        bad, rows_o, nrows_o = otree.classify_batch(sub, off, skip_exact=skip, raw_confidence=True, threads=threads, cap=64)
        assert bad == 0
        for j, q in enumerate(ids):
            q = int(q)
            t = int(t_o[j])
            assert int(res.t[q]) == t, f"query {q}: t = {int(res.t[q])}, oracle {t}"
            if pruned:
                o = check_run_as_left(index, q, t, counts_o[j], tables_o[j], index.n_refs, emul, lf, f"query {q} skip {skip}", tol=tol)
                seen["with_threshold"] += o["threshold"] > 0
                seen["on_records_path"] += int(o["records"])
                seen["thr"] += o["threshold"]
                seen["live"] += o["live"]
                seen["needed"] += o["needed"]
                seen["max_dp"] = max(seen["max_dp"], o["dp"])
                seen["max_dropped"] = max(seen["max_dropped"], o["dropped"])
            want = otree.rows_of(rows_o, nrows_o, j, 64)
            got = res.rows(q)
            if [g.lineage for g in got] == [r["idx"] for r in want] and [g.confidence_values for g in got] == [r["conf"] for r in want]:
                seen["rows_identical"] += 1
                for g, r in zip(got, want):
                    assert abs(g.local_signal - r["local_signal"]) < 1e-6 and abs(g.global_signal - r["global_signal"]) < 1e-9, q
            else:
                if lineages is None:
                    lineages = otree.lineages
                probs = tables_o[j][counts_o[j]]
                k = assert_rows_equivalent(got, want, probs, lineages, f"query {q} skip {skip}")
                assert k > 0, f"query {q}: rows differ from the oracle's without a tie"
                seen["ties"] += 1
            seen["n"] += 1
    return seen
