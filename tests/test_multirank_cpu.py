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
                                      np.array(conf).reshape(-1, 32), np.array(local))

    lo, hi = dist_util.shard_range(qs.n, rank, world)
    parts = dist_util.gather_records(dist, classify(lo, hi), rank, world)
    if rank == 0:
        got = [dist_util.unpack_records(p) for p in parts]
        want = dist_util.unpack_records(classify(0, qs.n))
        lin = np.concatenate([g["row_lineage"] for g in got])
        conf = np.concatenate([g["row_conf"] for g in got])
        local = np.concatenate([g["row_local_signal"] for g in got])
        nrows_per_q = np.concatenate([np.diff(g["row_off"]) for g in got])
        ok = (np.array_equal(lin, want["row_lineage"]) and np.array_equal(conf, want["row_conf"]) and
              np.array_equal(local, want["row_local_signal"]) and np.array_equal(nrows_per_q, np.diff(want["row_off"])) and
              sum(g["n_queries"] for g in got) == qs.n)
        np.save(out_path, np.array([ok, len(lin)]))
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


def _shard_worker(rank, world, port, out_path):
    """Exchange logic of the reference-sharded database (raxtax_amd/sharded.py) on CPU tensors over gloo:
    histogram all-reduce + prefix all-gather/assembly reproduce the unsharded histogram and prefix sums."""
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from raxtax_amd import sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)                      # same data on every rank
    n_q, n_refs, t = 5, 1000, 40
    counts = rng.integers(0, t + 1, size=(n_q, n_refs))
    probs = rng.random((n_q, n_refs))
    bnd = np.unique(np.concatenate([[0, n_refs], rng.integers(1, n_refs, 60)]))
    cuts = sharded.shard_cuts(n_refs, world)
    bnd = np.unique(np.concatenate([bnd, cuts]))        # cut points are boundaries on every rank
    lo, hi = cuts[rank], cuts[rank + 1]
    # what this rank's kernels would produce
    hist = np.stack([np.bincount(counts[q, lo:hi], minlength=t + 1) for q in range(n_q)]).astype(np.int32)
    local_b = bnd[(bnd >= lo) & (bnd <= hi)]
    pref_local = np.stack([np.concatenate([[0.0], np.cumsum(probs[q, lo:hi])])[local_b - lo] for q in range(n_q)])
    widths = [int(((bnd >= cuts[r]) & (bnd <= cuts[r + 1])).sum()) for r in range(world)]
    comm = sharded.TorchComm(dist, world, widths)
    h = torch.from_numpy(hist.copy())
    comm.allreduce_hist([h])
    parts = comm.allgather_prefix([torch.from_numpy(pref_local)])
    pref = sharded.assemble_prefix(parts).numpy()
    want_hist = np.stack([np.bincount(counts[q], minlength=t + 1) for q in range(n_q)])
    want_pref = np.stack([np.concatenate([[0.0], np.cumsum(probs[q])])[bnd] for q in range(n_q)])
    ok = np.array_equal(h.numpy(), want_hist) and pref.shape == want_pref.shape and np.max(np.abs(pref - want_pref)) < 1e-9
    # the ranks agree on the tile pruning before a run (ShardedClassifier._agree_on_pruning): one rank that cannot prune switches all off
    ok = ok and comm.agree_min(1 if rank == 0 else 0) == 0 and comm.agree_min(1) == 1
    if rank == 0:
        np.save(out_path, np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_sharded_db_exchange(tmp_path):
    import torch.multiprocessing as mp

    out = tmp_path / "ok2.npy"
    mp.spawn(_shard_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert np.load(out)[0]
