"""raxtax-synth (raxtax_amd/csrc/synth_main.cpp): the C++ twin of the synthetic generator.  Its PRNG is its own
(xoshiro256**), so it is held against the MODEL, not against the numpy generator's bytes: composition, divergence per
taxonomic level, shape of the taxonomy, shares of exact copies and of queries with N, determinism per seed.  The files are what
the reference's parser expects (src/parser.rs:11-40): the oracle's parser reads them back."""
import subprocess

import numpy as np
import pytest

from raxtax_amd import _build, synth


@pytest.fixture(scope="module")
def exe():
    return _build.build_synth()


def read_fasta(path):
    recs, head, cur = [], None, []
    for line in path.read_text().splitlines():
        if line.startswith(">"):
            if head is not None:
                recs.append((head, "".join(cur)))
            head, cur = line[1:], []
        else:
            cur.append(line)
    recs.append((head, "".join(cur)))
    return recs


def test_database_follows_the_model(exe, tmp_path):
    db = tmp_path / "db.fasta"
    subprocess.run([str(exe), "db", "3000", str(db)], check=True, capture_output=True)
    recs = read_fasta(db)
    assert len(recs) == 3000 and all(len(s) == synth.COI_LEN and set(s) <= set("ACGT") for _, s in recs)
    lin = [h.split("tax=")[1].rstrip(";") for h, _ in recs]
    assert [h.split(";")[0] for h, _ in recs] == [f"r{i}" for i in range(3000)]
    assert all([x[:2] for x in l.split(",")] == ["p:", "c:", "o:", "f:", "g:", "s:"] for l in lin)
    fan = synth.default_fanouts(3000)
    for d in range(6):   # fan-outs of synth.default_fanouts: distinct taxa per level
        assert len({",".join(l.split(",")[: d + 1]) for l in lin}) == int(np.prod(fan[: d + 1]))
    counts = np.unique(lin, return_counts=True)[1]
    assert counts.min() >= 2 and counts.max() <= 3                      # 3000 over 1296 species
    S = np.frombuffer("".join(s for _, s in recs).encode(), np.uint8).reshape(3000, -1)
    comp = np.array([(S == ord(c)).mean() for c in "ACGT"])
    assert np.abs(comp - synth.BASE_P).max() < 0.05     # one root of 658 sites under everything: sd of its composition ~ 0.015-0.02
    # divergence: two individuals of a species differ by about 2 * mu_7 * (1 - sum p^2) per site, two phyla by far more
    same = np.array([(S[i] != S[i + 1]).mean() for i in range(2999) if lin[i] == lin[i + 1]])
    het = 1.0 - float((synth.BASE_P ** 2).sum())
    assert abs(same.mean() - 2 * synth.MU[6] * het) < 0.003
    far = np.array([(S[i] != S[j]).mean() for i, j in zip(range(0, 900, 30), range(2100, 3000, 30))])
    assert lin[0].split(",")[0] != lin[2100].split(",")[0] and far.mean() > 0.15
    # the same seeds give the same file, another seed another one
    db2, db3 = tmp_path / "db2.fasta", tmp_path / "db3.fasta"
    subprocess.run([str(exe), "db", "3000", str(db2)], check=True, capture_output=True)
    subprocess.run([str(exe), "db", "3000", str(db3), "--seed-db", "7"], check=True, capture_output=True)
    assert db.read_bytes() == db2.read_bytes() and db.read_bytes() != db3.read_bytes()


def test_queries_follow_the_model_and_parse(exe, tmp_path, oracle):
    db, qf = tmp_path / "db.fasta", tmp_path / "q.fasta"
    subprocess.run([str(exe), "db", "1500", str(db)], check=True, capture_output=True)
    subprocess.run([str(exe), "queries", str(db), "4000", str(qf)], check=True, capture_output=True)
    refs = {s for _, s in read_fasta(db)}
    qs = read_fasta(qf)
    assert [h for h, _ in qs] == [f"q{i}" for i in range(4000)] and all(len(s) == synth.COI_LEN for _, s in qs)
    with_n = sum("N" in s for _, s in qs)
    exact = sum(s in refs for _, s in qs)
    assert 15 <= with_n <= 80 and all(1 <= s.count("N") <= 3 for _, s in qs if "N" in s)      # 1 %
    assert 320 <= exact <= 480                                                                  # 10 % (a mutated copy is never one)
    # what the reference's parsers make of the two files (parser.rs:11-40 as restated by the oracle)
    otree = oracle.parse_reference_fasta_str(db.read_text())
    assert len(otree.lineages) == 1500
    parsed = oracle.parse_query_fasta_str(qf.read_text())
    assert len(parsed) == 4000 and parsed[0][0] == "q0"
    assert subprocess.run([str(exe), "nothing"], capture_output=True).returncode == 2
