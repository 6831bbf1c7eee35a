"""Query sharding + result gather (the N > 1 path of bench.py) with world_size 2 on the gloo backend.
Each rank classifies its shard with the CPU oracle (no GPU here); rank 0 checks that the gathered records
equal a single-process classification of all queries."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from oracle.oracle_py import Oracle
    from raxtax_amd import dist_util, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    db = synth.make_db(600, fanouts=(2, 2, 2, 2, 2, 2))
    qs = synth.make_queries(db, 41, exact_frac=0.2)          # not divisible by the world size
    otree = Oracle().tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)

    def classify(lo, hi):
        row_off, lin, depth, conf, local = [0], [], [], [], []
        for q in range(lo, hi):
            rows, _ = otree.classify(qs.seq(q))
            for r in rows:
                lin.append(r["idx"]); depth.append(len(r["conf"])); local.append(r["local_signal"])
                conf.append(r["conf"] + [0.0] * (32 - len(r["conf"])))
            row_off.append(len(lin))
        return dist_util.pack_records(np.array(row_off), np.array(lin), np.array(depth),
                                      np.array(conf).reshape(-1, 32), np.array(local), first_query=lo)

    lo, hi = dist_util.shard_range(qs.n, rank, world)
    parts = dist_util.gather_records(dist, classify(lo, hi), rank, world)
    if rank == 0:
        got = np.concatenate(parts)
        want = classify(0, qs.n)
        np.save(out_path, np.array([got.shape == want.shape and np.array_equal(got, want), got.shape[0]]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from raxtax_amd.dist_util import shard_range

    for n in (0, 1, 7, 100, 100_000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.timeout(180)
def test_two_rank_gloo_gather(tmp_path):
    import torch.multiprocessing as mp

    out = tmp_path / "ok.npy"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    ok, n_rows = np.load(out)
    assert ok == 1 and n_rows >= 41
