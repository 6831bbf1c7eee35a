"""Eight handles in one process through rtx_raxtax_multi: the in-process counterpart of the 8-GPU run the builder cannot
launch -- eight device handles, each driven by a thread of its own inside the library, chunks dealt in turn, every handle staging its next
chunk while the current one runs, the messages back in input order.  All eight live on the one GPU of the test box (threads, not processes:
the box's guard counts processes); the database is large enough for the tile pruning and the records path (4 tiles), the size reduced."""
import numpy as np
import pytest

import raxtax_amd as rx
from raxtax_amd import synth

pytestmark = pytest.mark.gpu


def test_eight_handles_one_call(oracle):
    db = synth.make_db(30_000)
    qs = synth.make_queries(db, 24_000, seed=51, exact_frac=0.2)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    handles = [rx.Index(tree, sub_batch=0 if k % 2 == 0 else 512, overlap=bool(k % 3)) for k in range(8)]
    queries = [(qs.labels[q], qs.seq(q).copy()) for q in range(qs.n)]
    one, many = [], []
    rx.raxtax(queries, handles[0], False, False, 2048, lambda l, o, t: one.append((l, o)), False)
    rx.raxtax(queries, handles, False, False, 1000, lambda l, o, t: many.append((l, o)), False)
    assert [m[0] for m in many] == qs.labels, "messages out of input order"
    assert many == one, "eight handles give other text than one"
    assert all(h.debug_prune_stats()["pairs"] > 0 for h in handles), "a handle did not prune: the fast path was not what ran"
    # and the text is the oracle's (a seeded sample; exact ties between sibling taxa excepted and counted)
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    got = dict(many)
    n_diff = 0
    for q in np.random.default_rng(52).choice(qs.n, 400, replace=False):
        rows, raw = otree.classify(qs.seq(int(q)))
        n_diff += got[qs.labels[int(q)]] != otree.format_out(qs.labels[int(q)], raw)
    assert n_diff <= 2, n_diff
