"""BASELINE.json configs[0]: the reference's example data (a 600-record subset of
example/diptera_queries.fasta, committed as tests/golden/diptera_subset.fasta) classified against itself
(every query header carries a lineage, so the query file is a valid database).  Every
query has at least one exact match, which exercises the only_last branch (prob.rs:24-41), the exact-match
override (raxtax.rs:73-84) and --skip-exact-matches (raxtax.rs:65-68)."""
from pathlib import Path

import numpy as np
import pytest

FASTA = Path(__file__).resolve().parent / "golden" / "diptera_subset.fasta"


def test_oracle_self_classification_plumbing(oracle):
    text = FASTA.read_text()
    otree = oracle.parse_reference_fasta_str(text)
    queries = oracle.parse_query_fasta_str(text)
    assert otree.num_tips == len(queries) == 600
    n_override = 0
    for label, seq in queries[:60]:
        rows, raw = otree.classify(seq)
        assert len(rows) >= 1
        ex = otree.exact_matches(seq)
        assert len(ex) >= 1                       # self-classification: always an exact match
        if len(ex) == 1:
            n_override += 1
            assert all(c == 1.0 for c in rows[0]["conf"]) and len(rows) == 1
        line = otree.format_out(label, raw).split("\n")[0].split("\t")
        assert line[0] == label and len(line) == 5
        rows2, _ = otree.classify(seq, skip_exact=True)   # mislabelling mode: next best match
        assert len(rows2) >= 1
    assert n_override > 30


@pytest.mark.gpu
@pytest.mark.parametrize("skip,raw", [(False, False), (True, False), (False, True)])
def test_gpu_self_classification_matches_oracle(oracle, skip, raw):
    import raxtax_amd as rx
    from gpu_common import Excuses
    from test_gpu_parity import assert_rows_equivalent

    text = FASTA.read_text()
    otree = oracle.parse_reference_fasta_str(text)
    tree = rx.parse_reference_fasta_str(text)
    queries = rx.parse_query_fasta_str(text)
    assert tree.lineages == otree.lineages
    ix = rx.Index(tree)
    want = {}
    lins = otree.lineages
    for label, seq in queries:
        rows, rawrows = otree.classify(seq, skip_exact=skip, raw_confidence=raw)
        want[label] = otree.format_out(label, rawrows)
    got = {}
    rx.raxtax(queries, ix, skip, raw, 0, lambda l, o, t: got.__setitem__(l, o), False)
    assert set(got) == set(want)
    diff = [l for l in want if got[l] != want[l]]
    ex = Excuses(f"diptera600/skip={int(skip)}/raw={int(raw)}")
    ex.checked = len(want)
    # identical text, except exact floating-point ties between sibling taxa
    for l in diff:
        seq = dict(queries)[l]
        t, counts = otree.hit_counts(seq, skip_exact=skip)
        probs = oracle.highest_hit_prob_per_reference(t, t // 2, counts)
        rows, _ = otree.classify(seq, skip_exact=skip, raw_confidence=True)
        res = ix.classify(seq, np.array([0, len(seq)], np.uint64),
                          *ix.exact_matches(seq, np.array([0, len(seq)], np.uint64)), skip_exact_matches=skip)
        ties = assert_rows_equivalent(res.rows(0), rows, probs, lins, l)
        assert ties > 0, f"{l}: text differs from the oracle's without a tie"
        ex.tie()
    # real barcodes with duplicates: once the exact matches are zeroed (--skip-exact-matches) sibling
    # species with identical hit counts tie exactly in a few per cent of the queries; the count is pinned
    ex.check()
