"""`bench.py --emit-fasta DIR`: the database and the queries of a bench line as files the reference itself reads
(`raxtax -d DIR/db.fasta -i DIR/queries.fasta -t 0`, README.md:29-62 of the reference).  The emitted files must parse -- through the
library's own restatement of parser.rs:46-154 -- to the very arrays the bench classifies (the same generator calls, seeds included)."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

import raxtax_amd as rx
from raxtax_amd import synth

ROOT = Path(__file__).resolve().parent.parent


def test_emitted_fasta_parses_to_the_arrays_the_bench_classifies(tmp_path):
    n_refs, n_q = 3000, 700
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--emit-fasta", str(tmp_path), "--refs", str(n_refs), "--queries", str(n_q)],
                         capture_output=True, text=True, check=True)
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info["refs"] == n_refs and info["queries"] == n_q and "raxtax -d" in info["reference_command"]
    db = synth.make_db(n_refs)                                         # what main() builds for this size ...
    qs = synth.make_queries(db, n_q, seed=3, first_label=0)            # ... and rank 0's queries
    # queries: labels and sequences (N and all) as parser.rs:107-154 reads them
    parsed = rx.parse_query_fasta_str((tmp_path / "queries.fasta").read_text())
    assert len(parsed) == n_q
    for i, (label, seq) in enumerate(parsed):
        assert label == qs.labels[i] and np.array_equal(seq, qs.seq(i)), i
    # database: the tree built from the file equals the tree built from the arrays (lineage order, sequences, posting lists)
    t_file = rx.parse_reference_fasta_str((tmp_path / "db.fasta").read_text())
    t_arr = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    assert t_file.num_tips == t_arr.num_tips == n_refs and t_file.lineages == t_arr.lineages
    off_f, post_f = t_file.csr()
    off_a, post_a = t_arr.csr()
    assert np.array_equal(off_f, off_a) and np.array_equal(post_f, post_a)
    for q in (0, 17, n_q - 1):                                        # exact matches resolve to the same references
        assert np.array_equal(t_file.exact_matches(qs.seq(q)), t_arr.exact_matches(qs.seq(q)))
