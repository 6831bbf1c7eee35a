"""rtx_records_format (host_format.cpp): the `.out` texts of every query of a packed record buffer, formatted natively on several threads --
what a writer on rank 0 does with the gathered records of the ranks (BASELINE configs[3]; main.rs:126-136, lineage.rs:17-29,
utils.rs:62-68).  Held against a plain Python restatement of the reference's format on synthetic records: rows out of query order,
queries with several rows, queries without rows (status != 0), the single-exact-match override of raxtax.rs:73-84, every thread count."""
import numpy as np
import pytest

import raxtax_amd as rx
from raxtax_amd import dist_util, synth


def _expected(lineages, labels, begin, count, status, lin, depth, conf, local, gs, exact_one, flags):
    out = []
    for q in range(len(labels)):
        if status[q] != 0:
            out.append(b"")
            continue
        one = exact_one is not None and exact_one[q] != 0xFFFFFFFF and not flags
        lines = []
        for i in range(1 if one else int(count[q])):
            r = int(begin[q]) + i
            li = int(exact_one[q]) if one else int(lin[r])
            d = lineages[li].count(",") + 1 if one else int(depth[r])
            cf = [1.0] * d if one else [conf[r][k] for k in range(d)]
            lines.append("\t".join([labels[q], lineages[li], ",".join(f"{c:.2f}" for c in cf), f"{local[int(begin[q]) + (0 if one else i)]:.5f}", f"{gs[q]:.5f}"]))
        out.append("\n".join(lines).encode())
    return out


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("flags", [0, rx.RTX_SKIP_EXACT_MATCHES])
def test_records_format_equals_the_reference_format(threads, flags):
    db = synth.make_db(3000)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    lineages = tree.lineages
    rng = np.random.default_rng(7 + threads)
    n_q = 5000
    count = rng.choice([1, 1, 1, 2, 3], n_q).astype(np.uint32)
    status = (rng.random(n_q) < 0.02).astype(np.uint8)
    count[status != 0] = 0
    n_rows = int(count.sum())
    # the rows of a query are contiguous, the queries' blocks in a shuffled (processing) order
    order = rng.permutation(n_q)
    begin = np.zeros(n_q, np.int64)
    at = 0
    for q in order:
        begin[q] = at
        at += int(count[q])
    lin = rng.integers(0, len(lineages), n_rows).astype(np.uint32)
    depth = np.array([lineages[i].count(",") + 1 for i in lin], np.uint8)
    conf = rng.integers(0, 101, (n_rows, 32)) / 100.0
    local = rng.random(n_rows) * rng.choice([1.0, 1e-3, 40.0], n_rows)
    gs = rng.random(n_q) * 3.0
    gs[:50] = (np.arange(50) * 2 + 1) / 2e5          # x.xxxxx5: the ties of the rounding (decided by the binary value, as in the reference)
    labels = [f"q{q};sample={q % 7}" for q in range(n_q)]
    exact_one = np.where(rng.random(n_q) < 0.1, rng.integers(0, len(lineages), n_q), 0xFFFFFFFF).astype(np.uint32)
    buf = dist_util.pack_records(None, lin, depth, conf, local, global_signal=gs, row_begin=begin, row_count=count, t=np.full(n_q, 640), status=status)
    text, off = dist_util.format_records(tree, buf, labels, exact_one=exact_one, flags=flags, threads=threads)
    want = _expected(lineages, labels, begin, count, status, lin, depth, conf, local, gs, exact_one, flags)
    raw = text.tobytes()
    assert int(off[n_q]) == len(raw)
    for q in range(n_q):
        a, b = int(off[q]), int(off[q + 1])
        got = raw[a:b]
        if want[q] == b"":
            assert got == b"", q
        else:
            assert got == want[q] + b"\0", (q, got[:120], want[q][:120])


def test_records_format_rejects_other_buffers():
    db = synth.make_db(200)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    with pytest.raises(rx.RtxError):
        dist_util.format_records(tree, np.zeros(64, np.uint8), [])
