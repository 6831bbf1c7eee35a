"""raxtax-hip command line (raxtax_amd/csrc/cli_main.cpp): output files, `.bin` database cache, checkpoint /
resume semantics of the reference (io.rs:47-90,156-187,202-263; main.rs:72-99,126-136).  SURVEY.md 8f #4."""
import shutil
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CLI = ROOT / "raxtax_amd" / "raxtax-hip"
FASTA = ROOT / "tests" / "golden" / "diptera_subset.fasta"


def run(*args, ok=True):
    p = subprocess.run([str(CLI), *map(str, args)], capture_output=True, text=True, timeout=300)
    if ok:
        assert p.returncode == 0, p.stderr
    return p


def test_cli_outputs_resume_and_bin_cache(tmp_path, oracle):
    out = tmp_path / "run1"
    run("-d", FASTA, "-i", FASTA, "-o", out, "--tsv", "--batch", 128)
    lines = (out / "raxtax.out").read_text().splitlines()
    labels = (out / "raxtax.ckp").read_text().splitlines()
    assert len(labels) == 600 and len(set(labels)) == 600
    assert {l.split("\t")[0] for l in lines} == set(labels)
    assert (out / "raxtax.tsv").exists() and (out / "raxtax.json").exists() and (out / "diptera_subset.bin").exists()
    # every line of every query against the oracle (.out and .tsv); without --skip-exact-matches and --raw-confidence
    # 580 of the 600 queries take the single-exact-match override and the rest the full path -- a line may differ only
    # where test_config1_diptera verifies an exact tie for the same query and mode (at most its committed count)
    import json
    text = FASTA.read_text()
    otree = oracle.parse_reference_fasta_str(text)
    by_label = {}
    for l in lines:
        by_label.setdefault(l.split("\t")[0], []).append(l)
    tsv_by_label = {}
    for l in (out / "raxtax.tsv").read_text().splitlines():
        tsv_by_label.setdefault(l.split("\t")[0], []).append(l)
    n_diff = 0
    queries = oracle.parse_query_fasta_str(text)
    assert len(queries) == 600
    for label, seq in queries:
        rows, raw = otree.classify(seq)
        same = otree.format_out(label, raw).split("\n") == by_label[label] and otree.format_tsv(label, raw, seq).split("\n") == tsv_by_label[label]
        n_diff += not same
    allowed = json.loads((ROOT / "tests" / "golden" / "expected_excuses.json").read_text()).get("diptera600/skip=0/raw=0", {}).get("ties", 0)
    print(f"CLI: {n_diff} of 600 queries differ from the oracle's text (ties allowed: {allowed})")
    assert n_diff <= allowed
    # an existing output folder without --redo and without a checkpoint is refused (io.rs:241-243)
    (tmp_path / "occupied").mkdir()
    assert run("-d", FASTA, "-i", FASTA, "-o", tmp_path / "occupied", ok=False).returncode == 73

    # ---- interrupted run: 250 queries finished, a half-written line of an unfinished query in the output
    out2 = tmp_path / "run2"
    shutil.copytree(out, out2)
    done = labels[:250]
    (out2 / "raxtax.ckp").write_text("\n".join(done) + "\n")
    keep = [l for l in lines if l.split("\t")[0] in set(done)]
    (out2 / "raxtax.out").write_text("\n".join(keep) + "\n" + lines[-1][: len(lines[-1]) // 2] + "\n")
    tsv_lines = (out / "raxtax.tsv").read_text().splitlines()
    (out2 / "raxtax.tsv").write_text("\n".join(l for l in tsv_lines if l.split("\t")[0] in set(done)) + "\n")
    # the fingerprint is that of the database path used before -> same command resumes
    p = run("-d", FASTA, "-i", FASTA, "-o", out2, "--tsv", "--batch", 128)
    assert "Restarting from checkpoint" in p.stderr
    assert sorted((out2 / "raxtax.out").read_text().splitlines()) == sorted(lines)
    assert sorted((out2 / "raxtax.tsv").read_text().splitlines()) == sorted(tsv_lines)
    assert sorted((out2 / "raxtax.ckp").read_text().splitlines()) == sorted(labels)

    # ---- the cached .bin is accepted as database (Tree::load_from_file is tried first, parser.rs:38-40)
    out3 = tmp_path / "run3"
    run("-d", out / "diptera_subset.bin", "-i", FASTA, "-o", out3, "--skip-exact-matches", "-c")
    l3 = (out3 / "raxtax.out").read_text().splitlines()
    assert len({l.split("\t")[0] for l in l3}) == 600
    assert not (out3 / "raxtax.json").exists() and not (out3 / "raxtax.ckp").exists()   # -c cleans up
    # ... and classifies like the FASTA it was made from: every line against the oracle under --skip-exact-matches (f3:
    # tree.rs:147-164 -- taxonomy, lineages, exact-sequence map and k-mer map all come out of the file here)
    by3 = {}
    for l in l3:
        by3.setdefault(l.split("\t")[0], []).append(l)
    n_diff3 = 0
    for label, seq in queries:
        rows, raw = otree.classify(seq, skip_exact=True)
        n_diff3 += otree.format_out(label, raw).split("\n") != by3[label]
    allowed3 = json.loads((ROOT / "tests" / "golden" / "expected_excuses.json").read_text()).get("diptera600/skip=1/raw=0", {}).get("ties", 0)
    print(f"CLI from the cached .bin, --skip-exact-matches: {n_diff3} of 600 queries differ from the oracle's text (ties allowed: {allowed3})")
    assert n_diff3 <= allowed3

    # ---- a .bin written by an INDEPENDENT encoder of the format (tests/test_bin_format.py, fed from the oracle's tree: what
    # upstream's Tree::save_to_file would write) as the database: same text as from the FASTA
    import numpy as np
    from test_bin_format import write_bin_from_oracle
    seq_of = dict(queries)
    olabels = [l for l, _ in queries]
    orig = otree.original_index()
    seq_sorted = [seq_of[olabels[int(o)]] for o in orig]
    ind = tmp_path / "independent.bin"
    ind.write_bytes(write_bin_from_oracle(otree, otree.lineages, seq_sorted, np.random.default_rng(5)))
    out5 = tmp_path / "run5"
    run("-d", ind, "-i", FASTA, "-o", out5, "--tsv")
    assert sorted((out5 / "raxtax.out").read_text().splitlines()) == sorted(lines)
    assert sorted((out5 / "raxtax.tsv").read_text().splitlines()) == sorted((out / "raxtax.tsv").read_text().splitlines())
    # --only-db writes the database and stops
    out4 = tmp_path / "run4"
    run("-d", FASTA, "--only-db", "-o", out4)
    assert (out4 / "diptera_subset.bin").exists() and not (out4 / "raxtax.out").exists()


def test_cli_on_several_handles(tmp_path):
    """--devices 0,0: two index handles (here on the one GPU of the box; --gpus N puts them on devices 0 .. N-1), one driving thread
    each, chunks dealt in turn -- the output files are those of a single handle, line for line and in the same order."""
    a, b = tmp_path / "one", tmp_path / "two"
    run("-d", FASTA, "-i", FASTA, "-o", a, "--skip-db", "--tsv", "--batch", 64)
    run("-d", FASTA, "-i", FASTA, "-o", b, "--skip-db", "--tsv", "--batch", 64, "--devices", "0,0")
    for f in ("raxtax.out", "raxtax.tsv", "raxtax.ckp"):
        assert (a / f).read_text() == (b / f).read_text(), f
    assert run("-d", FASTA, "-i", FASTA, "-o", tmp_path / "none", "--gpus", "0", ok=False).returncode == 64


def test_cli_reads_gzip_input(tmp_path):
    """utils.rs:42-60 get_reader: inputs whose last extension is gz / gzip are decompressed on the fly -- database and queries; the
    query file in blocks that do not line up with anything in the compressed stream."""
    import gzip
    a, b = tmp_path / "plain", tmp_path / "gz"
    dbz, qz = tmp_path / "db.fasta.gz", tmp_path / "queries.fa.GZIP"
    dbz.write_bytes(gzip.compress(FASTA.read_bytes()))
    qz.write_bytes(gzip.compress(FASTA.read_bytes(), compresslevel=1))
    run("-d", FASTA, "-i", FASTA, "-o", a, "--tsv")
    run("-d", dbz, "-i", qz, "-o", b, "--tsv", "--block-bytes", 30000, "--batch", 100)
    for f in ("raxtax.out", "raxtax.tsv", "raxtax.ckp"):
        assert (a / f).read_text() == (b / f).read_text(), f
    assert (b / "db.fasta.bin").exists()                     # with_extension("bin"): the last extension only (io.rs:269-276)
    bad = tmp_path / "broken.fasta.gz"
    bad.write_bytes(gzip.compress(FASTA.read_bytes())[:2000])
    assert run("-d", FASTA, "-i", bad, "-o", tmp_path / "c", "--skip-db", ok=False).returncode != 0


def test_cli_streamed_query_ingest(tmp_path):
    """--block-bytes: the query file is read and parsed in blocks (cut in front of header lines) on a thread of its own;
    the output must be the one of a single-block run."""
    a, b = tmp_path / "whole", tmp_path / "blocks"
    run("-d", FASTA, "-i", FASTA, "-o", a, "--skip-db")
    run("-d", FASTA, "-i", FASTA, "-o", b, "--skip-db", "--block-bytes", 20000, "--batch", 100)
    la = (a / "raxtax.out").read_text().splitlines()
    lb = (b / "raxtax.out").read_text().splitlines()
    assert len(la) >= 600 and la == lb
    assert (a / "raxtax.ckp").read_text() == (b / "raxtax.ckp").read_text()


def test_cli_on_files_of_the_cpp_generator(tmp_path, oracle):
    """raxtax-synth (csrc/synth_main.cpp, the C++ twin of the generator of SURVEY.md 8d) writes a database and queries;
    raxtax-hip classifies them; every line against the oracle's text for the same files."""
    import json

    from raxtax_amd import _build
    exe = _build.build_synth()
    db, qf, out = tmp_path / "db.fasta", tmp_path / "q.fasta", tmp_path / "out"
    subprocess.run([str(exe), "db", "3000", str(db)], check=True, capture_output=True)
    subprocess.run([str(exe), "queries", str(db), "300", str(qf)], check=True, capture_output=True)
    run("-d", db, "-i", qf, "-o", out)
    by_label = {}
    for l in (out / "raxtax.out").read_text().splitlines():
        by_label.setdefault(l.split("\t")[0], []).append(l)
    otree = oracle.parse_reference_fasta_str(db.read_text())
    queries = oracle.parse_query_fasta_str(qf.read_text())
    assert len(queries) == 300 and set(by_label) == {l for l, _ in queries}
    n_diff = 0
    for label, seq in queries:
        rows, raw = otree.classify(seq)
        n_diff += otree.format_out(label, raw).split("\n") != by_label[label]
    allowed = json.loads((ROOT / "tests" / "golden" / "expected_excuses.json").read_text()).get("cli/synth3000", {}).get("ties", 0)
    print(f"CLI on raxtax-synth files: {n_diff} of 300 queries differ from the oracle's text (ties allowed: {allowed})")
    assert n_diff <= allowed
