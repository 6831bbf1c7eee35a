#!/usr/bin/env python3
"""Host-side rehearsal of configs[3] at full size WITHOUT eight GPUs: eight gloo ranks (CPU tensors) each own the packed result records
of 1 M classified queries (synthetic, the shape rtx_result_pack writes: 25 B per query + 13 + L bytes per row) and ship them to rank 0 with
the code bench.py uses at N = 8 (raxtax_amd/dist_util.py: size exchange, one padded gather, rank 0's copy of the eight buffers) -- the part
of the 8-GPU run that does not scale with the GPUs.  Prints the time per step on rank 0 beside the device step it has to stay under.
    python tools/gather_rehearsal.py [ranks] [queries per rank] [steps]"""
import os
import socket
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def worker(rank, world, port, n_q, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    import torch
    import torch.distributed as dist

    from raxtax_amd import dist_util

    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(rank)
    count = np.where(rng.random(n_q) < 0.15, 2, 1).astype(np.uint32)          # 1.15 rows per query: the bench workload
    n_rows = int(count.sum())
    row_off = np.concatenate([[0], np.cumsum(count)])
    rec = dist_util.pack_records(row_off, rng.integers(0, 500_000, n_rows), np.full(n_rows, 6), rng.integers(0, 101, (n_rows, 32)) / 100.0,
                                 rng.random(n_rows), global_signal=rng.random(n_q), t=np.full(n_q, 640), status=np.zeros(n_q))
    cache = [{}, {}]
    times = []
    dist.barrier()
    for i in range(steps + 1):
        t0 = time.perf_counter()
        parts = dist_util.gather_records(dist, rec, rank, world, device="cpu", cache=cache[i & 1])
        dt = time.perf_counter() - t0
        if i:
            times.append(dt)
    if rank == 0:
        t0 = time.perf_counter()
        u = dist_util.unpack_records(parts[-1])
        t_unpack = time.perf_counter() - t0
        assert u["n_queries"] == n_q and len(parts) == world and all(len(p) > 25 * n_q for p in parts)
        # the writer: rank 0 turns the records of EVERY rank into `.out` lines natively (rtx_records_format), all of its CPUs at work
        import ctypes

        import raxtax_amd as rx
        from raxtax_amd import synth
        db = synth.make_db(500_000)
        tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
        labels = (ctypes.c_char_p * n_q)(*[f"q{i}".encode() for i in range(n_q)])
        nt = os.cpu_count()
        # the writer keeps one output buffer per rank and formats into them step after step (a fresh 190-MB array per call costs 26 000 first-touch
        # page faults inside the call: 14 against 13 ms per rank on the bench host, 52 against 34 in an 8-CPU container)
        keep = [np.empty(4 * len(p_) + (1 << 20), np.uint8) for p_ in parts]
        for p_, k_ in zip(parts, keep):
            dist_util.format_records(tree, p_, labels, threads=nt, out=k_)    # (first touch of the tree's strings and of the buffers)
        t0 = time.perf_counter()
        n_bytes = 0
        for p_, k_ in zip(parts, keep):
            text, off = dist_util.format_records(tree, p_, labels, threads=nt, out=k_)
            n_bytes += len(text)
        t_fmt = time.perf_counter() - t0
        print(f"rank 0 formats the records of {world} ranks ({world * n_q} queries) into {n_bytes / 1e6:.0f} MB of .out lines in {1e3 * t_fmt:.0f} ms on {nt} threads "
              f"= {world * n_q / t_fmt / 1e6:.1f} M lines/s, {1e9 * t_fmt * nt / (world * n_q):.0f} ns of one thread per query (a 16-CPU grant: "
              f"{1e3 * t_fmt * nt / 16:.0f} ms for these {world} M queries beside a device step of ~80 ms)")
        print(f"{world} gloo ranks on {os.cpu_count()} CPUs, {n_q} queries per rank: {len(rec) / 1e6:.1f} MB of records per rank, "
              f"gather to rank 0: {1e3 * np.mean(times):.1f} ms per step (min {1e3 * min(times):.1f}, max {1e3 * max(times):.1f}) = "
              f"{world * len(rec) / np.mean(times) / 1e9:.2f} GB/s into rank 0; numpy unpack of ONE rank's buffer on one thread: {1e3 * t_unpack:.0f} ms "
              f"(the bench does not unpack inside its timed region: the records stay packed until a writer formats them)")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(world, port, n_q, steps), nprocs=world, join=True)
