"""RTX_OPT_RUN_AHEAD (rtx_raxtax over several chunks): chunk c + 1 is enqueued before the last sub-batch of chunk c has finished -- the result
state of a batch exists twice on the device, the scratch sets are shared.  The text of a call must not depend on it: every line equal to that of
one call with everything in one chunk (which cannot run ahead), the retry path included (a chunk that outgrows a buffer with the next one on the
device already), and equal to the oracle's on a sample."""
import numpy as np
import pytest

import raxtax_amd as rx
from gpu_common import Excuses, oracle_sample_parity
from raxtax_amd import synth

pytestmark = pytest.mark.gpu


def _lines(index, queries, chunk, skip=False):
    got = []
    rx.raxtax(queries, index, skip, False, chunk, lambda l, o, t: got.append((l, o, t)), True)
    return got


@pytest.mark.parametrize("aid", [0, 2])
def test_chunks_enqueued_ahead_print_the_lines_of_one_batch(oracle, aid):
    db = synth.make_db(60_000)
    qs = synth.make_queries(db, 100_000, seed=11)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    index = rx.Index(tree, device=0)
    assert index.prune_verdict[0]                      # pruned, two streams: the shape that runs ahead
    queries = [(qs.labels[i], qs.bases[qs.base_off[i]:qs.base_off[i + 1]]) for i in range(len(qs.labels))]
    single = _lines(index, queries, 0)
    assert index.run_ahead_stats == (0, 0)             # one chunk: nothing to run ahead of
    if aid:
        rx._lib.check(index._lib.rtx_index_set_option(index._h, 23, aid))   # every second run-ahead is abandoned (RTX_RETRY_CHUNK)
    for chunk in (40_000, 33_000, 35_000):   # (a sub-batch holds 16 384 queries at least: a chunk needs two of them for the two streams)
        before = index.run_ahead_stats
        got = _lines(index, queries, chunk)
        ahead, abandoned = (a - b for a, b in zip(index.run_ahead_stats, before))
        n_chunks = -(-len(queries) // chunk)
        assert got == single, chunk
        assert ahead >= (n_chunks - 1) // 2, (chunk, ahead, abandoned)   # (an abandoned one costs the run-ahead of the chunk behind it)
        assert abandoned > 0 or not aid, (chunk, ahead, abandoned)       # (without the aid: only while the handle's buffers find their size)
    if aid:
        rx._lib.check(index._lib.rtx_index_set_option(index._h, 23, 0))
    # ... and with exact matches skipped (the exact-match groups of a chunk are fetched before the next one is enqueued)
    assert _lines(index, queries, 40_000, skip=True) == _lines(index, queries, 0, skip=True)
    # the handle afterwards: a plain batch, held against the oracle
    sample = np.arange(0, len(queries), 97)[:200]
    otree = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    res = index.classify(qs.bases, qs.base_off)
    ex = Excuses(f"run_ahead/aid={aid}")
    oracle_sample_parity(index, oracle, otree, db, qs, sample, False, ex, full_res=res, chunk=100)
    ex.check()


def test_chunks_with_a_few_long_reads_among_the_barcodes(oracle):
    """Chunks that hold a side class (a handful of long reads) do not run ahead -- the chunk behind such a chunk is enqueued when it has left the
    device, the chunk behind a plain one ahead of its end: the text is that of one batch either way."""
    db = synth.make_db(60_000)
    qs = synth.make_queries(db, 70_000, seed=13)
    L = db.length
    rng = np.random.default_rng(14)
    refs = db.seq_bytes.reshape(db.n, L)
    queries = [(qs.labels[i], qs.bases[qs.base_off[i]:qs.base_off[i + 1]]) for i in range(len(qs.labels))]
    for k, (at, nb) in enumerate(((100, 1500), (20_000, 3000), (36_000, 1200), (69_000, 5000))):   # chunks of 35 000: the first and the second hold long reads
        s = np.concatenate([refs[int(i)] for i in rng.integers(0, db.n, nb // L + 1)])[:nb].copy()
        queries.insert(at + k, (f"long{k}", s))
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    index = rx.Index(tree, device=0)
    single = _lines(index, queries, 0)
    for chunk in (35_000, 17_600):
        assert _lines(index, queries, chunk) == single, chunk
    assert len(single) == len(queries) and all(o for _, o, _ in single)
