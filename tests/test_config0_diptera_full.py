"""BASELINE.json configs[0] at its size: ALL 7 868 records of the reference's example data (`example/diptera_queries.fasta`,
committed as the fixture tests/golden/diptera_queries.fasta -- data, the only real barcodes the reference ships; its
`diptera_references.fasta` is a missing blob) classified against themselves through the `raxtax-hip` CLI, every
line of every query against the oracle, in both exact-match modes.  Differences are accepted only as verified exact ties between
sibling taxa and are ledgered (tests/golden/expected_excuses.json: `diptera7868/...`)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
FASTA = ROOT / "tests" / "golden" / "diptera_queries.fasta"
CLI = ROOT / "raxtax_amd" / "raxtax-hip"
N = 7868


def test_fixture_is_the_reference_example(oracle):
    """The numbers SURVEY.md / BASELINE.md section 2 measured on the file: records, lengths, lineages, duplicates, t."""
    text = FASTA.read_text()
    otree = oracle.parse_reference_fasta_str(text)
    queries = oracle.parse_query_fasta_str(text)
    assert otree.num_tips == len(queries) == N
    lens = np.array([len(s) for _, s in queries])
    assert lens.min() == 195 and lens.max() == 208 and (lens == 205).mean() > 0.99
    assert len(set(otree.lineages)) == 4600
    assert N - len({bytes(s) for _, s in queries}) == 40                      # 40 duplicate sequences
    ts = np.array([len(oracle.sequence_to_kmers(s)) for _, s in queries[::16]])
    assert 180 <= ts.min() and ts.max() <= 198 and abs(ts.mean() - 194.7) < 1.0
    for _, s in queries[::97]:
        assert len(otree.exact_matches(s)) >= 1                                # self-classification: always an exact match


@pytest.mark.gpu
@pytest.mark.parametrize("skip", [False, True])
def test_cli_self_classification_of_all_records(tmp_path, oracle, skip):
    import raxtax_amd as rx
    from gpu_common import Excuses
    from test_gpu_parity import assert_rows_equivalent

    out = tmp_path / ("skip" if skip else "plain")
    args = [str(CLI), "-d", str(FASTA), "-i", str(FASTA), "-o", str(out), "--tsv"] + (["--skip-exact-matches"] if skip else [])
    p = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    text = FASTA.read_text()
    otree = oracle.parse_reference_fasta_str(text)
    queries = oracle.parse_query_fasta_str(text)
    labels = (out / "raxtax.ckp").read_text().splitlines()
    assert len(labels) == N and len(set(labels)) == N
    by_label, tsv_by_label = {}, {}
    for l in (out / "raxtax.out").read_text().splitlines():
        by_label.setdefault(l.split("\t")[0], []).append(l)
    for l in (out / "raxtax.tsv").read_text().splitlines():
        tsv_by_label.setdefault(l.split("\t")[0], []).append(l)
    assert set(by_label) == set(labels) == {l for l, _ in queries}
    differ = []
    for label, seq in queries:
        rows, raw = otree.classify(seq, skip_exact=skip)
        if otree.format_out(label, raw).split("\n") != by_label[label] or otree.format_tsv(label, raw, seq).split("\n") != tsv_by_label[label]:
            differ.append((label, seq))
    ex = Excuses(f"diptera7868/skip={int(skip)}/raw=0")
    ex.checked = N
    if differ:   # every differing query must be an exact tie of sibling taxa by the ORACLE's own probabilities
        tree = rx.parse_reference_fasta_str(text)
        ix = rx.Index(tree)
        lins = otree.lineages
        for label, seq in differ:
            t, counts = otree.hit_counts(seq, skip_exact=skip)
            probs = oracle.highest_hit_prob_per_reference(t, t // 2, counts)
            rows, _ = otree.classify(seq, skip_exact=skip, raw_confidence=True)
            off = np.array([0, len(seq)], np.uint64)
            res = ix.classify(seq, off, *ix.exact_matches(seq, off), skip_exact_matches=skip)
            ties = assert_rows_equivalent(res.rows(0), rows, probs, lins, label)
            assert ties > 0, f"{label}: text differs from the oracle's without a tie"
            ex.tie()
    print(f"configs[0] in full, skip={skip}: {len(differ)} of {N} queries print another (tied) lineage than the oracle")
    ex.check()
