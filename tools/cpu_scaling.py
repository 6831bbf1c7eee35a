#!/usr/bin/env python3
"""How the CPU port (oracle/) scales with threads on this host: queries/s at 1, 2, 4, ... threads with the reference's
chunk rule, plus what the kernel says about the CPUs this process may use (affinity, cgroup quota).
Usage: tools/cpu_scaling.py [refs] [seconds per point]"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.oracle_py import Oracle  # noqa: E402
from raxtax_amd import synth  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, Path(f).read_text().strip())
    except OSError:
        pass
orc = Oracle(native=True)
print("physical cores", len(orc.physical_core_ids()))
db = synth.make_db(n_refs)
qs = synth.make_queries(db, 60000)
t0 = time.time()
ot = orc.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
print(f"oracle tree {time.time() - t0:.1f}s")
L = db.length
rate1 = None
T = 1
while T <= (os.cpu_count() or 1):
    n = 100 * T if T > 1 else 40
    t0 = time.time()
    ot.classify_batch(qs.bases[: n * L], qs.base_off[: n + 1], threads=T)
    dt = time.time() - t0
    k = max(1, int(secs / dt))
    if k > 1 and n * k <= qs.n:
        n *= k
        t0 = time.time()
        ot.classify_batch(qs.bases[: n * L], qs.base_off[: n + 1], threads=T)
        dt = time.time() - t0
    r = n / dt
    rate1 = rate1 or r
    print(f"threads {T:4d}: {n:6d} queries in {dt:6.1f} s = {r:8.1f} q/s, {r / T:6.2f} per thread, efficiency {r / T / rate1:5.2f}", flush=True)
    T *= 2
