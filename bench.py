#!/usr/bin/env python3
"""bench.py -- classified queries/s of the raxtax hot path on N MI355X of one node.

A "step" is one pass of the whole device path (exact-match lookup Tree.sequences.get, raxtax.rs:42 -> processing order ->
kmer_extract -> hit_count -> prob_lookup -> taxon_prefix + lineage walk -> result rows finalised on the host) over one batch
of synthetic queries per GPU, with the queries and the index already resident in HBM when the timed region starts
(`--host-exact-match`: the lookup on the host instead, once, untimed -- rounds 1 and 2).

The line verifies itself (SURVEY.md 8d, rank 0 at N = 1): after the timed region `parity_sample` holds the LAST TIMED STEP, as the device
left it, against the CPU oracle -- the size-independent properties of all its results, and a seeded 2 000-query sample of its last
sub-batch: hit counts of every visited tile bit-exact, no unvisited tile above the query's threshold, histogram, probabilities, result
rows (raxtax_amd/checks.py).  A violation is printed in the line and the process exits with status 4.

Beside `value` the line carries its own caveats (rank 0 at N = 1, `--no-extras` leaves them out): `value_incl_h2d` (the
queries cross PCIe every step: staged beside the running step, and not overlapped), `value_end_to_end` (host buffers -> rtx_raxtax ->
formatted result strings, no disk; busy time of every pipeline stage), `value_unpruned` (RTX_OPT_TILE_PRUNE = 0: every tile counted), a
sweep over the divergence of the queries from their source reference (`divergence_sweep`: the tile pruning depends on how far a query's
best hit stands out) and `value_real_composition` (the reference's hold-out methodology on its own Diptera records: real barcodes whose
every tile holds a relative).

Default workload = BASELINE.json configs[2], the largest single-GPU configuration: 1 M synthetic COI-length
(658 bp) queries per GPU vs a 500k-sequence database replicated per GPU (`--config 1` = configs[1]: 100k queries vs
50k references).  Queries are sharded, so scaling is weak: every rank classifies its own --queries; with N > 1 the
per-rank result records are gathered on rank 0 over RCCL inside the timed region (the only collective of the path,
BASELINE.json configs[3]).  `--shard-db` = configs[4]: the REFERENCES are sharded, every rank classifies the same
queries, histograms are all-reduced and boundary prefix sums all-gathered per sub-batch.

`python bench.py --gpus N` launches the N ranks itself (torch.distributed.run, one process per GPU) when it is not
already running under a launcher; the parent never touches the GPU.

One JSON line on rank 0: metric/value (whole-job queries/s), ms_per_step, `roofline` of the dominant kernel
(hit_count) and `cpu_baseline` (the CPU oracle, a C port of the reference algorithm, timed on this host's cores on a
bounded sample).  How every number of `roofline` is derived: DESIGN.md section 5.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before torch can initialise HIP (raxtax_amd/csrc/host_threads.cpp: transfers get hardware queues of their own)
import socket  # noqa: E402
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

# MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 measured for a copy); L2 34.5 TB/s aggregate; 16.8-18.8 TB/s
# measured for L2-resident row gathers (1 KiB rows by index, the access shape of hit_count); FP64 vector 78.6 TFLOP/s
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
L2_GATHER_GBS = 18800.0
FP64_VALU_GFLOPS = 78600.0

CONFIGS = {1: (50_000, 100_000, "BASELINE.json configs[1]"), 2: (500_000, 1_000_000, "BASELINE.json configs[2]")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configs[i]: 2 = 1 M queries vs 500k references (default, the headline), "
                         "1 = 100k queries vs 50k references")
    ap.add_argument("--refs", type=int, default=0, help="reference sequences (overrides --config)")
    ap.add_argument("--queries", type=int, default=0, help="queries per GPU per step (overrides --config)")
    ap.add_argument("--sub-batch", type=int, default=0, help="queries per kernel launch (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of each leg of the cpu_baseline sample")
    ap.add_argument("--no-cluster", action="store_true", help="process the queries in input order (RTX_OPT_CLUSTER = 0)")
    ap.add_argument("--u16-counts", action="store_true", help="counts travel as u16 instead of packed 10 bits (RTX_OPT_PACKED_COUNTS = 0)")
    ap.add_argument("--no-pair", action="store_true", help="hit_count with one query per wave (RTX_OPT_HIT_PAIR = 0; A/B measurements)")
    ap.add_argument("--no-locator", action="store_true", help="processing order by min-hash alone (RTX_OPT_LOCATOR = 0; A/B measurements)")
    ap.add_argument("--no-tile-prune", action="store_true", help="hit_count counts every tile of 8192 references (RTX_OPT_TILE_PRUNE = 0; default: only the tiles that can hold a reference with any probability)")
    ap.add_argument("--tile-prune", action="store_true", help="(the default; kept for older command lines)")
    ap.add_argument("--no-fine-bounds", action="store_true", help="tile pruning with its first stage of bounds only (RTX_OPT_FINE_BOUNDS = 0; A/B measurements)")
    ap.add_argument("--emit-fasta", metavar="DIR", default=None,
                    help="write the database and the queries of this line (rank 0's, as --refs / --queries / --mu-q / --exact-frac select them) to "
                         "DIR/db.fasta and DIR/queries.fasta and exit (no GPU needed): the inputs the reference itself can be timed on, "
                         "`raxtax -d DIR/db.fasta -i DIR/queries.fasta -t 0`")
    ap.add_argument("--records", type=int, default=None, help="RTX_OPT_RECORDS: pruned queries with at most this many live tiles write records of the counts above their threshold instead of counts (0: off; default: the library's)")
    ap.add_argument("--overlap", type=int, default=None, help="RTX_OPT_OVERLAP: 1 = back half of a sub-batch on a second stream beside the front half of the next (default: the library's)")
    ap.add_argument("--no-two-level", action="store_true", help="bounds pass of the tile pruning over blocks of 64 throughout (RTX_OPT_TWO_LEVEL_BOUNDS = 0; A/B measurements)")
    ap.add_argument("--two-level-rule", type=int, default=None, help="RTX_OPT_TWO_LEVEL_BOUNDS > 1: the rule that picks the refined groups of tiles, packed (experiments)")
    ap.add_argument("--no-tile-skip", action="store_true", help="taxon_prefix sums every reference (RTX_OPT_TILE_SKIP = 0; A/B measurements)")
    ap.add_argument("--mu-q", type=float, default=0.02, help="per-site substitution rate of a query against its source reference (the headline: 0.02)")
    ap.add_argument("--exact-frac", type=float, default=0.10, help="share of the queries that are exact copies of a reference (the headline: 0.10)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-share", type=int, default=0,
                    help="rtx_set_host_share(K): size every host pool of the library as if K ranks shared this host's CPUs (K = 8 on a 16-CPU grant: "
                         "two threads -- what a rank of the 8-GPU line gets); 0 = LOCAL_WORLD_SIZE or 1")
    ap.add_argument("--host-cpus", type=int, default=0,
                    help="restrict this process (and every thread it starts: HIP runtime, library pools) to its first C CPUs with sched_setaffinity before "
                         "anything is loaded -- the stricter rehearsal of a rank's CPU share (0 = leave the affinity mask alone)")
    ap.add_argument("--no-parity", action="store_true", help="N > 1: skip the self-check of the line (rank 0 holds a seeded sample of every rank's gathered records against the oracle)")
    ap.add_argument("--no-extras", action="store_true", help="only the headline: no H2D / end-to-end / unpruned legs, no divergence sweep")
    ap.add_argument("--host-exact-match", action="store_true",
                    help="Tree.sequences.get on the host, once, untimed (rounds 1-2); default: on the device inside the timed step")
    ap.add_argument("--e2e-chunk", type=int, default=131072, help="queries per device batch of the end-to-end leg (rtx_raxtax chunk_size)")
    ap.add_argument("--skip-exact-matches", action="store_true")
    ap.add_argument("--hit-events-only", action="store_true",
                    help="HIP events around hit_count only (default: around every kernel; costs < 0.1 % of a step)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL over xGMI; gloo only to smoke-test the "
                         "multi-rank flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--shard-db", action="store_true",
                    help="BASELINE configs[4]: shard the REFERENCES over the GPUs (every rank classifies the same "
                         "queries; RCCL all-reduce of histograms + all-gather of prefix sums per sub-batch)")
    ap.add_argument("--shard-mode", default="refs", choices=["refs", "kmers"],
                    help="with --shard-db: refs = contiguous reference ranges per GPU, all-reduce of histograms + all-gather of "
                         "prefix sums (SURVEY.md 8e mode B, default); kmers = k-mer ranges per GPU, all-reduce of the u16 "
                         "per-reference hit counts (mode A, the literal wording of BASELINE.json configs[4])")
    args = ap.parse_args()
    refs, queries, name = CONFIGS[args.config]
    args.config_name = name if not (args.refs or args.queries) else "custom size"
    args.refs = args.refs or refs
    args.queries = args.queries or queries
    return args


def emit_fasta(args) -> int:
    """--emit-fasta DIR: the exact inputs of the line as files the reference reads (README.md:29-62 of the reference: `-d` a FASTA whose
    headers carry `tax=...;`, `-i` a FASTA of queries).  The arrays are the ones main() classifies: the same generator calls, seeds included
    (queries of rank 0: seed 3)."""
    from raxtax_amd import synth

    out = Path(args.emit_fasta)
    out.mkdir(parents=True, exist_ok=True)
    db = synth.make_db(args.refs)
    qs = synth.make_queries(db, args.queries, seed=3, first_label=0, mu_q=args.mu_q, exact_frac=args.exact_frac)
    nb_db = synth.write_fasta(out / "db.fasta", [f"r{i};tax={lin};" for i, lin in enumerate(db.lineages)], db.seq_bytes, db.seq_off)
    nb_q = synth.write_fasta(out / "queries.fasta", qs.labels, qs.bases, qs.base_off)
    sha = hashlib.sha256()
    sha.update(np.ascontiguousarray(db.seq_bytes).tobytes())
    sha.update(np.ascontiguousarray(qs.bases).tobytes())
    print(json.dumps({"emitted": str(out), "db_fasta_bytes": nb_db, "queries_fasta_bytes": nb_q, "refs": db.n, "queries": qs.n,
                      "sha256_of_the_encoded_arrays": sha.hexdigest(),
                      "reference_command": f"raxtax -d {out / 'db.fasta'} -i {out / 'queries.fasta'} -t 0 --redo"}))
    return 0


# ------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks (before anything in this process touches the GPU)
# ------------------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    """`python bench.py --gpus N` on its own: one child process per GPU through torch.distributed.run
    (rendezvous on 127.0.0.1), this process only relays their output and checks that the printed line is the
    line of an N-rank run.  The reference's counterpart is one call as well (par_chunks, raxtax.rs:35-36)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")                # the launcher's own default, set here to keep it quiet
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    n_gpus_seen = None
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.startswith("{"):
            try:
                n_gpus_seen = json.loads(line).get("n_gpus")
            except ValueError:
                pass
    rc = proc.wait()
    if rc != 0:
        return rc
    if n_gpus_seen != args.gpus:
        print(f"bench.py: asked for --gpus {args.gpus} but the line reports n_gpus = {n_gpus_seen}", file=sys.stderr)
        return 3
    return 0


# ------------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------------
def available_parallelism() -> int:
    """Threads the reference would start by default (rayon -> std::thread::available_parallelism): the CPUs of the
    affinity mask, capped by the cgroup CPU quota (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            quota = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_model() -> str:
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_context(db):
    """The checker of the `cpu_baseline` leg (and of `parity_sample`, which is that leg's oracle held against the timed run): the C port
    of the reference algorithm, COMPILED HERE -- `-march=native` must mean the host whose cores are timed, not the container the
    repository was built in -- and its Tree::new of the bench database.  Never on the measured path."""
    from oracle import oracle_py

    t0 = time.perf_counter()
    lib_path = oracle_py.build(native=True, force=True)
    t_build = time.perf_counter() - t0
    orc = oracle_py.Oracle(native=True)
    t0 = time.perf_counter()
    otree = orc.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    info = oracle_py.build_info(native=True)
    info.update(library=str(lib_path.name), build_seconds=round(t_build, 2), built_on=socket.gethostname(), cpu_model=cpu_model())
    return dict(orc=orc, otree=otree, t_tree=time.perf_counter() - t0, build=info)


def parity_block(args, rx, index, view, ctx, db, qs, flags, n_sample=2000):
    """parity on each run.  After the timed region, untimed, on the LAST timed step as the device left it:
      * the size-independent properties of every result of the step (raxtax_amd/checks.py:check_properties: status, t, signals,
        confidences monotone along a lineage, rows sorted, level sums);
      * a seeded sample of the queries of the step's last sub-batch -- the ones whose counts, histograms, live masks and tables are
        still in HBM -- against the oracle WITHOUT running anything again (checks.as_run_oracle_sample): t and the hit counts of every
        visited tile bit-exact, no unvisited tile above the query's threshold, the histogram, table / Z within 1e-9 (north_star:
        1e-6), the rows the step returned identical to the oracle's (exact ties between sibling taxa verified and counted).
    A violation does not stop the line: it is reported as ok = false with its message, and bench.py exits non-zero."""
    from raxtax_amd import checks

    t0 = time.perf_counter()
    out = {"n": 0, "ok": False, "where": "last sub-batch of the last timed step, as the run left it (no recount); seed 20264"}
    try:
        res = rx.Result(view)
        checks.check_properties(res, db, qs.n)
        out["properties_checked_on"] = int(qs.n)
        seen = checks.as_run_oracle_sample(index, res, ctx["orc"], ctx["otree"], qs.bases, qs.base_off, n_sample, bool(flags),
                                           threads=available_parallelism())
        out.update(n=seen["n"], ok=True, counts_bit_exact=True, max_dp=seen["max_dp"], ties=seen["ties"], rows_identical=seen["rows_identical"],
                   queries_with_threshold=seen["with_threshold"], mean_threshold=seen["thr"] / max(seen["n"], 1),
                   tiles_visited_per_query=seen["live"] / max(seen["n"], 1), tiles_above_threshold_per_query=seen["needed"] / max(seen["n"], 1),
                   max_mass_of_a_dropped_set=seen["max_dropped"], tolerance_asserted=1e-9, on_records_path=seen.get("on_records_path", 0))
    except AssertionError as e:
        out["error"] = str(e)[:400] or "assertion failed"
    except Exception as e:      # (an RtxError of a tap, an OSError of the oracle ...): ok stays false, the measured line is still printed
        out["error"] = f"{type(e).__name__}: {str(e)[:400]}"
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def leg_parity(rx, index, view, ctx, lineages, seq_bytes, seq_off, q_bases, q_off, flags, n_sample):
    """parity_sample of a caveat leg: the leg's LAST step as the device left it -- a seeded sample of its last sub-batch
    against the oracle (its own Tree::new of the leg's database): t, the counts of the visited tiles bit-exact and no unvisited tile above
    the threshold where the leg pruned, probabilities within 1e-9, the rows the step returned (checks.as_run_oracle_sample)."""
    from raxtax_amd import checks

    if ctx is None:
        return None
    t0 = time.perf_counter()
    out = {"n": 0, "ok": False, "where": "last sub-batch of the leg's last step, as the run left it (no recount)"}
    try:
        otree = ctx["orc"].tree_new_flat(lineages, seq_bytes, seq_off)
        res = rx.Result(view)
        seen = checks.as_run_oracle_sample(index, res, ctx["orc"], otree, q_bases, q_off, n_sample, bool(flags), threads=available_parallelism())
        out.update(n=seen["n"], ok=True, pruned=bool(seen["with_threshold"]), counts_bit_exact=bool(seen["with_threshold"]) or None, max_dp=seen["max_dp"], ties=seen["ties"],
                   rows_identical=seen["rows_identical"], queries_with_threshold=seen["with_threshold"],
                   tiles_visited_per_query=seen["live"] / max(seen["n"], 1), on_records_path=seen.get("on_records_path", 0), tolerance_asserted=1e-9)
    except AssertionError as e:
        out["error"] = str(e)[:400] or "assertion failed"
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {str(e)[:400]}"
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def multirank_parity_block(args, rx, index, view, ctx, db, qs, flags, world, last_parts, n_own=600, n_other=200):
    """The self-check of a line with N > 1 ranks, rank 0, after the timed region, untimed:
      * rank 0's own last timed step as its device left it: checks.as_run_oracle_sample on its last sub-batch (as parity_block at N = 1);
      * the GATHERED records of every other rank -- what arrived on rank 0 over the collective in the last timed step -- unpacked and
        held against the oracle on a seeded sample of that rank's queries (every rank's queries follow from its rank number: seed 3 + r):
        t, the lineages and the confidences of every row, the signals.
    A violation is reported as ok = false with its message (the process then exits with status 4)."""
    from raxtax_amd import checks, dist_util, synth

    t0 = time.perf_counter()
    out = {"n": 0, "ok": False, "ranks_checked": [], "where": "rank 0: its last sub-batch as the run left it + a seeded sample of every other rank's gathered records"}
    try:
        res = rx.Result(view)
        checks.check_properties(res, db, qs.n)
        seen = checks.as_run_oracle_sample(index, res, ctx["orc"], ctx["otree"], qs.bases, qs.base_off, n_own, bool(flags), threads=available_parallelism())
        out.update(n=seen["n"], counts_bit_exact=True, max_dp=seen["max_dp"], ties=seen["ties"], rows_identical=seen["rows_identical"], on_records_path=seen.get("on_records_path", 0))
        out["ranks_checked"].append(0)
        if last_parts is None or len(last_parts) != world:
            raise AssertionError("rank 0 holds no gathered records of the last step")
        otree, orc = ctx["otree"], ctx["orc"]
        lineages = None
        for r in range(1, world):
            rec = dist_util.unpack_records(last_parts[r])
            assert rec["n_queries"] == args.queries, f"rank {r}: {rec['n_queries']} queries gathered, {args.queries} expected"
            assert (rec["status"] == 0).all(), f"rank {r}: queries with a status"
            qr = synth.make_queries(db, args.queries, seed=3 + r, first_label=r * args.queries, mu_q=args.mu_q, exact_frac=args.exact_frac)
            ids = np.sort(np.random.default_rng(20264 + r).choice(args.queries, min(n_other, args.queries), replace=False))
            L = db.length
            sub = np.concatenate([qr.bases[int(q) * L:(int(q) + 1) * L] for q in ids])
            off = (np.arange(len(ids) + 1) * L).astype(np.uint64)
            t_o, counts_o = otree.hit_counts_batch(sub, off, skip_exact=bool(flags), threads=available_parallelism())
            tables_o, z_o, rc = orc.prob_tables_batch(t_o, counts_o, threads=available_parallelism())
            bad, rows_o, nrows_o = otree.classify_batch(sub, off, skip_exact=bool(flags), raw_confidence=True, threads=available_parallelism(), cap=64)
            assert bad == 0
            for j, q in enumerate(ids):
                q = int(q)
                assert int(rec["t"][q]) == int(t_o[j]), f"rank {r} query {q}: t"
                a, b = int(rec["row_off"][q]), int(rec["row_off"][q + 1])
                want = otree.rows_of(rows_o, nrows_o, j, 64)
                got_lin = [int(x) for x in rec["row_lineage"][a:b]]
                got_conf = [[float(c) for c in rec["row_conf"][i][: int(rec["row_depth"][i])]] for i in range(a, b)]
                if got_lin == [w["idx"] for w in want] and got_conf == [[float(c) for c in w["conf"]] for w in want]:
                    assert abs(float(rec["global_signal"][q]) - want[0]["global_signal"]) < 1e-9, f"rank {r} query {q}: global signal"
                    for i, w in zip(range(a, b), want):
                        assert abs(float(rec["row_local_signal"][i]) - w["local_signal"]) < 1e-6, f"rank {r} query {q}: local signal"
                    out["rows_identical"] = out.get("rows_identical", 0) + 1
                else:   # an exact tie between sibling taxa: verified from the oracle's probabilities (checks.assert_rows_equivalent)
                    class _Row:
                        pass
                    rows_g = []
                    for i in range(a, b):
                        g = _Row()
                        g.lineage, g.confidence_values, g.local_signal = int(rec["row_lineage"][i]), got_conf[i - a], float(rec["row_local_signal"][i])
                        rows_g.append(g)
                    if lineages is None:
                        lineages = otree.lineages
                    k = checks.assert_rows_equivalent(rows_g, want, tables_o[j][counts_o[j]], lineages, f"rank {r} query {q}")
                    assert k > 0, f"rank {r} query {q}: rows differ from the oracle's without a tie"
                    out["ties"] = out.get("ties", 0) + 1
                out["n"] += 1
            out["ranks_checked"].append(r)
        out["ok"] = True
    except AssertionError as e:
        out["error"] = str(e)[:400] or "assertion failed"
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {str(e)[:400]}"
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def sharded_parity_block(args, rx, index, view, ctx, db, qs, flags, cuts, n_sample=300):
    """The self-check of a --shard-db line (both modes), rank 0, untimed: every rank classified the SAME queries against its shard and
    the exchanges (histogram all-reduce + prefix all-gather, or the all-reduce of the counts) put them together -- so rank 0's final rows
    of a seeded sample must be the oracle's rows of the whole database, and (reference shards) the hit counts of rank 0's shard, read
    through the recounting tap for the sampled queries of its last sub-batch, the slice [cuts[0], cuts[1]) of the oracle's counts."""
    from raxtax_amd import checks

    t0 = time.perf_counter()
    out = {"n": 0, "ok": False, "shard_counts_checked": 0, "where": "rank 0: final rows of a seeded sample against the oracle on the whole database"}
    try:
        res = rx.Result(view)
        otree, orc = ctx["otree"], ctx["orc"]
        L = db.length
        ids = np.sort(np.random.default_rng(20264).choice(qs.n, min(n_sample, qs.n), replace=False))
        sub = np.concatenate([qs.bases[int(q) * L:(int(q) + 1) * L] for q in ids])
        off = (np.arange(len(ids) + 1) * L).astype(np.uint64)
        t_o, counts_o = otree.hit_counts_batch(sub, off, skip_exact=bool(flags), threads=available_parallelism())
        tables_o, z_o, rc = orc.prob_tables_batch(t_o, counts_o, threads=available_parallelism())
        bad, rows_o, nrows_o = otree.classify_batch(sub, off, skip_exact=bool(flags), raw_confidence=True, threads=available_parallelism(), cap=64)
        assert bad == 0
        lineages = None
        ties = 0
        for j, q in enumerate(ids):
            q = int(q)
            assert int(res.t[q]) == int(t_o[j]), f"query {q}: t"
            want, got = otree.rows_of(rows_o, nrows_o, j, 64), res.rows(q)
            if [g.lineage for g in got] == [w["idx"] for w in want] and [g.confidence_values for g in got] == [w["conf"] for w in want]:
                for g, w in zip(got, want):
                    assert abs(g.local_signal - w["local_signal"]) < 1e-6 and abs(g.global_signal - w["global_signal"]) < 1e-9, q
            else:
                if lineages is None:
                    lineages = otree.lineages
                assert checks.assert_rows_equivalent(got, want, tables_o[j][counts_o[j]], lineages, f"query {q}") > 0, f"query {q}: rows differ without a tie"
                ties += 1
            out["n"] += 1
        out["ties"] = ties
        if cuts is not None:   # the counts of this rank's references
            try:
                last = set(int(x) for x in checks.last_sub_batch_queries(index, qs.n))
                for j, q in enumerate(ids):
                    if int(q) in last and out["shard_counts_checked"] < 50:
                        assert np.array_equal(index.debug_hit_counts(int(q)), counts_o[j][cuts[0]:cuts[1]]), f"query {int(q)}: counts of rank 0's shard"
                        out["shard_counts_checked"] += 1
            except rx.RtxError as e:
                out["shard_counts_note"] = f"taps not available on this handle: {str(e)[:120]}"
        out["ok"] = True
    except AssertionError as e:
        out["error"] = str(e)[:400] or "assertion failed"
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {str(e)[:400]}"
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def cpu_baseline(db, qs, target_s: float, skip_exact: bool, ctx):
    """Times the oracle (oracle/, kind "port": a C restatement of raxtax.rs:35-88, compiled -O3 -march=native ON THIS HOST at the
    start of the leg: oracle_context) on this host's cores.  Three legs on prefixes of the same queries: one thread; all usable
    threads; one thread per physical core, pinned (the reference's --pin, utils.rs:139-197).  The parallel legs use the reference's chunk
    rule (main.rs:119-124: max(100, n / (10 T) + 1) queries per work item); their samples are whole multiples of
    100 T queries, so that every thread gets the same number of full chunks -- a shorter sample would leave most
    threads idle in the last wave and understate the rate the reference reaches on a full batch."""
    T = available_parallelism()
    orc, otree, t_tree = ctx["orc"], ctx["otree"], ctx["t_tree"]
    L = db.length

    def run(n, threads, pins=None):
        n = int(min(n, qs.n))
        t0 = time.perf_counter()
        otree.classify_batch(qs.bases[: n * L], qs.base_off[: n + 1], skip_exact=skip_exact, threads=threads, pin_cpus=pins)
        return n, time.perf_counter() - t0

    # one thread: a probe sets the sample size
    n, dt = run(4, 1)
    n1, dt1 = run(max(8, int(4 / dt * min(target_s, 6.0))), 1)
    rate1 = n1 / dt1

    def leg(threads, pins=None):
        if threads == 1:
            return dict(value=rate1, cores=1, queries=n1, seconds=dt1)
        wave = 100 * threads                       # one full chunk per thread
        n, dt = run(wave, threads, pins)           # a first wave measures the rate (SMT siblings, memory system)
        k = max(1, int(round(target_s * (n / dt) / wave)))
        if k > 1 or n < wave:
            n, dt = run(wave * k, threads, pins)
        return dict(value=n / dt, cores=threads, queries=n, seconds=dt)

    every = leg(T)
    phys = orc.physical_core_ids()[:T]          # --pin: thread i on the i-th physical core (utils.rs:139-158)
    pinned = leg(len(phys), phys) if len(phys) > 1 else None
    out = {"value": every["value"], "unit": "queries/s", "cores": T, "kind": "port", "cpu_model": cpu_model(),
           "host_cpus": {"logical": os.cpu_count(), "usable": T,
                         "note": "usable = what std::thread::available_parallelism gives the reference here: affinity mask capped by the cgroup CPU quota"},
           "sample": f"first {every['queries']} queries of the same workload on {T} threads in {every['seconds']:.1f} s "
                     f"({every['queries'] // max(100 * T, 1)} full chunk(s) of 100 per thread, chunking as src/main.rs:119-124), "
                     f"no string formatting; oracle Tree::new {t_tree:.1f} s (untimed)",
           "per_thread": every["value"] / T,
           "one_thread": {"value": rate1, "queries": n1, "seconds": round(dt1, 2)},
           "build": ctx["build"],
           "note": "the Rust reference itself cannot be built here (no cargo/rustc): this is the C port (oracle/oracle.c)"}
    if pinned is not None:
        out["pinned_physical_cores"] = {"value": pinned["value"], "cores": len(phys), "queries": pinned["queries"],
                                        "seconds": round(pinned["seconds"], 2), "per_thread": pinned["value"] / len(phys), "cpus": phys,
                                        "note": "thread i on the i-th physical core of the affinity mask in ascending CPU order -- the choice of the reference's --pin "
                                                "(utils.rs:160-197 takes the first sibling of every core in /sys order, setup_threadpool_pinned the first T of them).  On an "
                                                "EPYC host with a cgroup quota but no cpuset these are neighbouring cores of two or three CCDs: they share those CCDs' L3 "
                                                "slices and fabric links, while the unpinned threads are spread by the scheduler over the whole socket (every thread an L3 "
                                                "of its own) -- the counting loop streams 170 MB of postings per query, so the unpinned leg is the faster one here"}
    return out


# ------------------------------------------------------------------------------------------------------------
# roofline pieces
# ------------------------------------------------------------------------------------------------------------
def device_source_sha() -> str:
    """Fingerprint of the device code: profiles/traffic.json is only used for the build it was measured on."""
    h = hashlib.sha256()
    for f in sorted((ROOT / "raxtax_amd" / "csrc").glob("rtx_*")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def measured_traffic(refs: int, query_len: int, queries: int, pruned: bool):
    """Fabric-side bytes of the hit_count launches from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/make_traffic.py from `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of THIS bench command at THIS size: one whole step, every
    sub-batch): per kind of launch -- "live" (the tiles of the database that are counted) and "bounds" (the same kernel on the union
    bitmap, tile pruning) -- the sums over the launches of a step.  FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for
    16-byte-per-lane reads on gfx950, + WRITE_SIZE.  Returned only if the profile was taken on this configuration AND on this build of
    the kernels; else None (PMC counters cannot be read from inside the process being measured)."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None, "no profiles/traffic.json"
    try:
        t = json.loads(f.read_text())
        e = t.get("configs", {}).get(f"refs={refs},query_len={query_len},queries={queries}")
        if e is None:
            return None, "no PMC profile of this configuration (refs, query length, queries per step)"
        if bool(e.get("pruned")) != bool(pruned):
            return None, "PMC profile was taken with another setting of the tile pruning"
        if e.get("device_source_sha") != device_source_sha():
            return None, "PMC profile is of another build of the kernels (re-run tools/profile_bench.sh + tools/make_traffic.py)"
        out = {}
        for kind, k in e["kinds"].items():
            n = max(int(k["launches"]), 1)
            out[kind] = dict(launches=n, fetch=2.0 * k["fetch_kb"] * 1024.0 / n, write=k["write_kb"] * 1024.0 / n,
                             l2_hit_rate=(k["tcc_hit"] / k["tcc_req"]) if k.get("tcc_req") else None)
            out[kind]["bytes"] = out[kind]["fetch"] + out[kind]["write"]
        out["source"] = e.get("source", "")
        return out, None
    except Exception as ex:  # noqa: BLE001 - a broken profile file must not break the bench
        return None, f"profiles/traffic.json unreadable: {ex}"


# which kernels run inside which HIP-event stage of a step (the index build and the table build are not part of a step)
STAGE_KERNELS = {
    "order": ("rtx::sketch_kernel", "rtx::locator_kernel", "rtx::class_keys_kernel", "rtx::invert_perm_kernel", "rtx::identity_perm_kernel"),
    "exact_match": ("rtx::exact_match_kernel",),
    "kmer_extract": ("rtx::kmer_extract_kernel<false>",),
    "pair_union": ("rtx::pair_union_kernel",),
    "tile_bounds": ("rtx::bounds2_kernel<10>", "rtx::bounds2_kernel<8>", "rtx::hit_count_pair_kernel<10, true, 1, true>", "rtx::hit_count_pair_kernel<8, true, 1, true>", "rtx::heavy_items_kernel",
                    "rtx::hit_count_pair_kernel<10, true, 1, false>", "rtx::hit_count_pair_kernel<8, true, 1, false>"),
    "tile_prune": ("rtx::prune_kernel", "rtx::kmer_extract_kernel<true>", "rtx::hit_count_pair_kernel<10, true, 2, true>", "rtx::hit_count_pair_kernel<8, true, 2, true>",
                   "rtx::live_offsets_kernel", "rtx::live_items_kernel", "rtx::fine_count_kernel", "rtx::fine_scan_kernel", "rtx::fine_scatter_kernel",
                   "rtx::pair_live_recount_kernel"),
    "hit_count": ("rtx::hit_count_pair_kernel<10, true, 0, true>", "rtx::hit_count_pair_kernel<8, true, 0, true>", "rtx::hit_count_pair_kernel<10, true, 0, false>",
                  "rtx::hit_count_kernel<10, true, false>", "rtx::hit_count_kernel<12, false, false>", "rtx::hit_count_kernel<16, false, false>"),
    "prob_table": ("rtx::prob_lookup_kernel", "rtx::prob_order_kernel", "rtx::prob_table_kernel<false>", "rtx::prob_table_kernel<true>"),
    "taxon_prefix": ("rtx::taxon_prefix_kernel<2, true, true>", "rtx::taxon_prefix_kernel<4, true, true>", "rtx::records_tail_kernel", "rtx::lineage_walk_kernel"),
}


def step_bounds(args, stage_ms_per_step, ms_per_step, n_queries_step, query_len, pruned):
    """the roofline the design actually has.  SURVEY 8d's HBM-read roofline (4 H_q + L_q algorithmic bytes against 8 TB/s)
    bounds no kernel of this step any more -- a bitmap bit stands for a 4-byte posting, 60 of 62 tiles are never read.  What does:
      * step_fabric_frac: the fabric-side bytes of EVERY kernel of a step (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC passes of this
        command on this build: profiles/traffic.json `step`) / ms_per_step / 8 TB/s;
      * bounds: per stage its time (HIP events of this run), its fabric rate, and the resource that limits it with the counter that shows
        it (SQ counters per kernel of the same command: profiles/sq_counters.json) -- the L1 request rate of 64 B/clk/CU for the two
        counting launches, instruction issue or the wait for dependent round trips for the kernels around them.
    Both profiles are keyed to the hash of the device sources: a line of another build carries null instead of stale numbers."""
    out = {"step_fabric_frac": None, "bounds": None, "note": "SURVEY 8d's HBM-read roofline no longer bounds any kernel of the step (DESIGN.md section 5)"}
    sha = device_source_sha()
    per_kernel, sq = None, None
    try:
        t = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        e = t.get("configs", {}).get(f"refs={args.refs},query_len={query_len},queries={n_queries_step}")
        if e and e.get("device_source_sha") == sha and bool(e.get("pruned")) == bool(pruned) and "step" in e:
            per_kernel = e["step"]["per_kernel"]
    except (OSError, ValueError):
        pass
    try:
        q = json.loads((ROOT / "profiles" / "sq_counters.json").read_text())
        if q.get("device_source_sha") == sha:
            sq = q["kernels"]
    except (OSError, ValueError):
        pass
    if per_kernel is not None:
        in_step = {k for ks in STAGE_KERNELS.values() for k in ks}
        tot = sum(2.0 * v["fetch_kb"] * 1024.0 + v["write_kb"] * 1024.0 for k, v in per_kernel.items() if k in in_step)
        out["step_fabric_bytes"] = tot
        out["step_fabric_frac"] = tot / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9)
    rows = {}
    for stage, ms in stage_ms_per_step.items():
        if not ms:
            continue
        row = {"ms": round(ms, 2)}
        ks = STAGE_KERNELS.get(stage, ())
        if per_kernel is not None:
            b = sum(2.0 * per_kernel[k]["fetch_kb"] * 1024.0 + per_kernel[k]["write_kb"] * 1024.0 for k in ks if k in per_kernel)
            row["fabric_tb_per_s"] = round(b / (ms * 1e-3) / 1e12, 2)
        if sq is not None:
            present = [(k, sq[k]) for k in ks if k in sq]
            if present:
                k, c = max(present, key=lambda kc: kc[1]["gui_mcycles"])   # the kernel of the stage that runs longest
                if "hit_count" in k:
                    lim = "L1 request rate (64 B/clk/CU: a 1-KiB row per 16 cycles) with VALU beside it"
                elif c["wait_pct"] >= 50.0:
                    lim = "latency: chains of dependent round trips (wave cycles waiting)"
                elif c["issue_pct"] + c["stall_pct"] >= 45.0:
                    lim = "instruction issue"
                else:
                    lim = "mixed"
                row.update(limit=lim, kernel=k, counters={"wait_pct": c["wait_pct"], "stall_pct": c["stall_pct"], "issue_pct": c["issue_pct"],
                                                          "valu_per_wave": c["valu_per_wave"], "vmem_rd_per_wave": c["vmem_rd_per_wave"],
                                                          "waves_in_flight_per_cu": c["waves_in_flight_per_cu"]})
        rows[stage] = row
    out["bounds"] = rows
    if per_kernel is None or sq is None:
        out["missing"] = "profiles/traffic.json (step) and / or profiles/sq_counters.json are of another build or size: tools/refresh_profiles.sh + tools/sq_profile.sh"
    return out


def roofline_block(args, work, prob_work, stage_ms, stage_n, n_queries_step, query_len, prune=None, ntiles=None):
    """Roofline of the dominant kernel, hit_count, per launch (one launch = one sub-batch).  With tile pruning the kernel runs twice
    per sub-batch: on the union bitmap (the bounds: stage tile_bounds) and on the live (pair, tile) blocks of the database (stage
    hit_count).  The two kinds differ by a factor of five in bytes and are reported apart: the top level is the counting of the live
    tiles, `bounds_pass` the other; `launch_ms_both_kinds` is their mean, which is how rocprofv3 --stats reports the kernel."""
    n_sub = max(stage_n["hit_count"], 1)
    pruned = bool(stage_n.get("tile_bounds"))
    launches_per_step = n_sub / args.steps
    q_per_launch = n_queries_step / launches_per_step                                   # queries of a sub-batch (mean: the last one is short)
    live_ms = stage_ms["hit_count"] / n_sub
    live_per_q = work.get("live_bytes", work["bitmap_bytes_read"]) / n_queries_step
    bounds_per_q = work.get("bounds_bytes", 0) / n_queries_step
    alg_per_q = (4 * work["sum_hits"] + work["sum_query_bytes"]) / n_queries_step      # 4 H_q + L_q
    achieved = live_per_q * q_per_launch / (live_ms * 1e-3) / 1e9
    step_ms_hit = (stage_ms["hit_count"] + stage_ms.get("tile_bounds", 0.0)) / args.steps
    alg_gbs = alg_per_q * n_queries_step / (step_ms_hit * 1e-3) / 1e9                  # against everything the kernel does in a step
    tr, why = measured_traffic(args.refs, query_len, n_queries_step, pruned)
    out = {
        # the unit that limits the kernel: the path from the XCD's L2 through the vector L1 (rows are gathered by index, 1 KiB per
        # wave-instruction).  achieved = bitmap-row bytes requested per launch / launch time.
        "bound": "l2", "kernel": "hit_count_kernel" if args.no_pair else "hit_count_pair_kernel",
        "launch": "the tiles of the database that are counted" + (" (live tiles; the bounds pass of the tile pruning: bounds_pass)" if pruned else ""),
        "achieved": achieved, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": achieved / L2_PEAK_GBS,
        "frac_of_measured_l2_gather_rate": achieved / L2_GATHER_GBS,
        "launch_ms": live_ms, "launches_per_step": launches_per_step, "queries_per_launch": q_per_launch,
        "requested_bytes_per_query": live_per_q,
        # SURVEY.md 8d's per-unit figure (u32 postings the reference would stream) -- a ratio, not a fraction: one bitmap bit stands
        # for a 4-byte posting, and with tile pruning most postings are never touched
        "algorithmic_bytes_per_query": alg_per_q, "algorithmic_GBps": alg_gbs,
        "algorithmic_ratio_to_hbm_peak": alg_gbs / HBM_PEAK_GBS,
        "traffic": None, "hbm_achieved": None, "hbm_peak": HBM_PEAK_GBS, "hbm_frac": None,
    }
    if pruned:
        b_n = max(stage_n["tile_bounds"], 1)
        b_ms = stage_ms["tile_bounds"] / b_n
        b_ach = bounds_per_q * q_per_launch / (b_ms * 1e-3) / 1e9
        out["launch_ms_both_kinds"] = (stage_ms["hit_count"] + stage_ms["tile_bounds"]) / (n_sub + b_n)
        out["bounds_pass"] = {"launch_ms": b_ms, "requested_bytes_per_query": bounds_per_q, "achieved": b_ach, "frac": b_ach / L2_PEAK_GBS,
                              "frac_of_measured_l2_gather_rate": b_ach / L2_GATHER_GBS, "traffic": None,
                              "note": "the same kernel on the union bitmap (one column per block of 64 references, 66 MB at 500k references: it "
                                      "sits in the 256 MB Infinity Cache -- what FETCH_SIZE counts here are refills of the eight L2s from it, not HBM reads)"}
    if prune is not None and prune.get("pairs"):
        out["tile_pruning"] = {
            "live_tiles_per_pair": prune["live_tiles_per_pair"], "live_tiles_per_query": prune.get("live_tiles_per_query"), "tiles": ntiles,
            "live_tiles_per_pair_first_stage": prune.get("live_tiles_per_pair_first_stage"), "fine_blocks_per_pair": prune.get("fine_blocks_per_pair"),
            "mean_threshold": prune["mean_threshold"], "mean_best_hit_lower_bound": prune["mean_best_hit_lower_bound"],
            "tiles_above_threshold_per_query": prune.get("tiles_above_threshold_per_query"),     # what exact knowledge would count
            "queries_with_threshold": prune.get("queries_with_threshold"),
            "note": "a pruned query's references with a count up to its threshold carry < 1e-10 of probability together (rtx_prune.hip); "
                    "value_unpruned counts every tile, divergence_sweep shows how the live tiles grow with the distance of a query from its best hit",
        }
    if tr is not None:
        live = tr.get("live") or tr.get("all")
        out["traffic"] = live["bytes"]                                   # fabric bytes per launch (PMC), means over the launches of a step
        out["hbm_achieved"] = live["bytes"] / (live_ms * 1e-3) / 1e9
        out["hbm_frac"] = out["hbm_achieved"] / HBM_PEAK_GBS
        out["traffic_fetch_bytes"] = live["fetch"]
        out["traffic_write_bytes"] = live["write"]
        out["l2_hit_rate"] = live["l2_hit_rate"]
        out["traffic_launches_profiled"] = live["launches"]
        out["traffic_source"] = tr["source"]
        if pruned and "bounds" in tr:
            b = tr["bounds"]
            bp = out["bounds_pass"]
            bp["traffic"] = b["bytes"]
            bp["fabric_achieved"] = b["bytes"] / (bp["launch_ms"] * 1e-3) / 1e9
            bp["l2_hit_rate"] = b["l2_hit_rate"]
    else:
        out["traffic_note"] = why
    # secondary: the probability stage (prob.rs:43-90).  ops_prob = D_q (n_q + 1) (2 exp + 1 log), SURVEY.md 8d
    if stage_n.get("prob_table"):
        p_ms = stage_ms["prob_table"] / stage_n["prob_table"]
        ops_q = 3.0 * prob_work["grid_points"] / n_queries_step
        gops = ops_q * q_per_launch / (p_ms * 1e-3) / 1e9
        out["prob_stage"] = {"kernel": "prob_lookup_kernel", "bound": "fp64-valu (secondary, SURVEY.md 8d)",
                             "ops_prob_per_query": ops_q, "distinct_counts_per_query": prob_work["distinct_counts"] / n_queries_step,
                             "launch_ms": p_ms, "achieved": gops, "peak": FP64_VALU_GFLOPS, "unit": "Gop/s (f64 exp/log of the reference's grid)",
                             "frac": gops / FP64_VALU_GFLOPS}
    return out


# ------------------------------------------------------------------------------------------------------------
# the caveats of the headline, measured in the same run (rank 0, N = 1)
# ------------------------------------------------------------------------------------------------------------
def extras_block(args, rx, lib, index, tree, db, qs, flags):
    """What `value` leaves out or depends on, each timed here with its own barrier-free loop on the one GPU:
      value_incl_h2d     upload (H2D of one byte per base) + run + download per step: the queries are NOT resident
      value_end_to_end   rtx_raxtax (src/raxtax.rs:14-97 mirrored): host buffers in, formatted `.out` strings out (a sender that
                         discards them: no disk), exact-match lookup, override and formatting included
      value_unpruned     RTX_OPT_TILE_PRUNE = 0: hit_count counts every tile (the floor under the headline)
      divergence_sweep   131 072 queries whose distance from their source reference is 2 / 5 / 10 / 15 % per site (no exact copies):
                         the further a query is from its best hit, the lower its threshold and the more tiles stay live."""
    out = {}
    n_q = qs.n
    steps = 3

    def timed(fn, warm=1):
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        return (time.perf_counter() - t0) / steps

    # under --host-exact-match every leg uploads the host's ids, as the headline did; else the device looks them up
    host_ids = index.exact_matches(qs.bases, qs.base_off) if (args.host_exact_match or not index.has_exact_lookup) else ()
    lookup = "host map, ids uploaded with the queries" if host_ids else "device, inside the step"

    # ---- queries cross PCIe every step: the batch of step i + 1 is staged (packed two bases per byte into pinned memory, asynchronous
    # H2D on a stream of its own) while step i runs -- what rtx_raxtax does with its chunks
    index.prefetch(qs.bases, qs.base_off, *host_ids)

    def with_upload():
        index.activate()
        index.run(flags)
        index.prefetch(qs.bases, qs.base_off, *host_ids)      # the next step's queries: host packing + transfer beside the kernels
        index.download(copy=False)
    dt = timed(with_upload)

    def serial_upload():
        index.upload(qs.bases, qs.base_off, *host_ids)
        index.run(flags)
        index.download(copy=False)
    index.activate()
    dt_serial = timed(serial_upload)
    out["value_incl_h2d"] = {"value": n_q / dt, "ms_per_step": dt * 1e3, "steps": steps, "exact_match_lookup": lookup,
                             "value_not_overlapped": n_q / dt_serial, "ms_per_step_not_overlapped": dt_serial * 1e3,
                             "what": "every step's queries come from host memory (%.0f MB one byte per base, %.0f MB over PCIe: two bases per byte): "
                                     "rtx_batch_activate + rtx_batch_run + rtx_batch_prefetch of the NEXT step's queries + rtx_batch_download; "
                                     "not_overlapped = rtx_batch_upload + run + download one after the other" % (len(qs.bases) / 1e6, len(qs.bases) / 2e6)}
    # ---- through the host mirror of raxtax() to strings
    labels = (ctypes.c_char_p * n_q)(*[l.encode() for l in qs.labels])
    SENDER = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p)
    lib.rtx_raxtax.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_char_p), rx._lib.u8p, rx._lib.u64p,
                               ctypes.c_int, ctypes.c_int, ctypes.c_uint64, SENDER, ctypes.c_void_p, ctypes.c_int]
    discard = ctypes.cast(lib.rtx_sender_discard, SENDER)
    counted = (ctypes.c_uint64 * 2)()
    bases = np.ascontiguousarray(qs.bases)
    off = np.ascontiguousarray(qs.base_off)

    def e2e():
        rx._lib.check(lib.rtx_raxtax(index._h, tree._h, n_q, labels, rx._lib.ptr(bases, rx._lib.u8p), rx._lib.ptr(off, rx._lib.u64p),
                                     int(bool(flags)), 0, args.e2e_chunk, discard, ctypes.cast(counted, ctypes.c_void_p), 0))
    # the host side of this leg (sixteen formatting threads, the sender) shares the granted CPUs with whatever else the host runs: the
    # median of five calls, with the spread beside it
    e2e()
    e2e_ms = []
    for _ in range(5):
        t0 = time.perf_counter()
        e2e()
        e2e_ms.append((time.perf_counter() - t0) * 1e3)
    dt = float(np.median(e2e_ms)) / 1e3
    busy = (ctypes.c_double * 4)()
    n_chunks = ctypes.c_uint64()
    rx._lib.check(lib.rtx_raxtax_last_timing(busy, ctypes.byref(n_chunks)))
    out["value_end_to_end"] = {"value": n_q / dt, "ms_per_step": dt * 1e3, "steps": 5, "statistic": "median of the calls", "ms_per_call": [round(x, 1) for x in e2e_ms],
                               "chunk_size": args.e2e_chunk,
                               "chunks": int(n_chunks.value),
                               "busy_ms_last_call": {"host_lookup": busy[0] * 1e3, "device_stage": busy[1] * 1e3, "format": busy[2] * 1e3, "sender": busy[3] * 1e3},
                               "text_bytes_per_query": counted[1] / max(counted[0], 1),
                               "what": "rtx_raxtax: host buffers -> H2D -> exact-match lookup + classification on the device -> D2H -> override + "
                                       "formatting of the .out lines (raxtax.rs:73-87) -> sender (discards: no disk); pipelined over chunks"}
    # ---- every tile counted
    rx._lib.check(lib.rtx_index_set_option(index._h, 13, 0))

    def plain():
        index.run(flags)
        index.download(copy=False)
    index.upload(qs.bases, qs.base_off, *host_ids)
    dt = timed(plain)
    out["value_unpruned"] = {"value": n_q / dt, "ms_per_step": dt * 1e3, "steps": steps, "exact_match_lookup": lookup,
                             "what": "RTX_OPT_TILE_PRUNE = 0: hit_count counts every tile"}
    rx._lib.check(lib.rtx_index_set_option(index._h, 13, 0 if args.no_tile_prune else 1))
    # ---- divergence sweep
    from raxtax_amd import synth
    sweep = []
    for k, mu in enumerate((0.02, 0.05, 0.10, 0.15)):
        q2 = synth.make_queries(db, 131072, seed=40 + k, mu_q=mu, exact_frac=0.0, n_frac=0.0)
        index.upload(q2.bases, q2.base_off)
        dt = timed(plain)
        st = index.debug_prune_stats()
        stg = {s_: round(ms, 2) for s_, (ms, n) in index.stage_times().items() if n}
        # the same queries with every tile counted: what the pruning is worth at this divergence
        rx._lib.check(lib.rtx_index_set_option(index._h, 13, 0))
        index.upload(q2.bases, q2.base_off)
        dt_full = timed(plain)
        rx._lib.check(lib.rtx_index_set_option(index._h, 13, 0 if args.no_tile_prune else 1))
        sweep.append({"mu_q": mu, "value": 131072 / dt, "ms_per_step": dt * 1e3, "value_unpruned": 131072 / dt_full, "live_tiles_per_pair": st["live_tiles_per_pair"],
                      "live_tiles_per_pair_first_stage": st["live_tiles_per_pair_first_stage"], "fine_blocks_per_pair": st["fine_blocks_per_pair"],
                      "stage_ms_per_step": stg,
                      "live_tiles_per_query": st.get("live_tiles_per_query"),
                      "share_with_threshold": st["queries_with_threshold"] / 131072, "mean_threshold": st["mean_threshold"],
                      "mean_best_hit_lower_bound": st["mean_best_hit_lower_bound"],
                      "tiles_above_threshold_per_query": st["tiles_above_threshold_per_query"]})
    out["divergence_sweep"] = {"queries": 131072, "steps": steps, "exact_copies": 0.0,
                               "note": "per-site substitution rate of a query against its source reference (the headline workload: 0.02 and 10 % exact copies)",
                               "rows": sweep}
    index.upload(qs.bases, qs.base_off, *host_ids)     # leave the handle as the headline had it
    return out


def real_composition_block(args, rx, lib, flags, ctx=None):
    """value_real_composition: the reference's methodology (scripts/common.py:11-25: real sequences, 90 % -> database, 10 % held out as
    queries) on the only real data it ships -- the 7 868 Diptera COI records of example/diptera_queries.fasta (committed as
    tests/golden/diptera_queries.fasta; ~205 bp, t ~ 195) -- scaled to a 14-tile database (raxtax_amd/synth.py:
    real_composition_holdout: every database record 16 times with individual-level substitutions).  A query's best hit is a
    relative at its natural distance, k-mers common to most references give unrelated references 40 % of the best count: the regime
    where the tile pruning buys least -- the handle's self-sample (include/raxtax_hip.h: rtx_index_self_sample) leaves it off.  As created, pruning
    forced on and pruning off on the same handle; inputs resident as for `value`."""
    from raxtax_amd import synth

    fasta = ROOT / "tests" / "golden" / "diptera_queries.fasta"
    if not fasta.exists():
        return {"error": "tests/golden/diptera_queries.fasta not found"}
    h = synth.real_composition_holdout(fasta)
    tree = rx.Tree.new_flat(h.lineages, h.seq_bytes, h.seq_off, kmer_map=False)
    index = rx.Index(tree, device=0, stage_timing=True)
    n_q = len(h.q_off) - 1
    steps = 3

    def plain():
        index.run(flags)
        index.download(copy=False)

    def timed(fn):
        fn()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        return (time.perf_counter() - t0) / steps
    # the handle as it comes: its self-sample (rtx_index_self_sample: a property of the database) decides whether tile pruning runs
    pruning_on, self_live = index.prune_verdict
    index.upload(h.q_bases, h.q_off)
    dt = timed(plain)
    stages = {s: round(ms, 3) for s, (ms, n) in index.stage_times().items() if n}
    view = index.download(copy=False)
    ok = int((np.ctypeslib.as_array(view.status, shape=(n_q,)) == 0).sum())
    parity = leg_parity(rx, index, view, ctx, h.lineages, h.seq_bytes, h.seq_off, h.q_bases, h.q_off, flags, 300)
    # tile pruning forced on (RTX_OPT_PRUNE_SELF_SAMPLE = 0) and off (RTX_OPT_TILE_PRUNE = 0), same handle
    rx._lib.check(lib.rtx_index_set_option(index._h, 22, 0))
    index.upload(h.q_bases, h.q_off)
    dt_pruned = timed(plain)
    st = index.debug_prune_stats()
    stages_pruned = {s: round(ms, 3) for s, (ms, n) in index.stage_times().items() if n}
    rx._lib.check(lib.rtx_index_set_option(index._h, 13, 0))
    index.upload(h.q_bases, h.q_off)
    dt_full = timed(plain)
    stages_full = {s: round(ms, 3) for s, (ms, n) in index.stage_times().items() if n}
    ntiles = (len(h.lineages) + 8191) // 8192
    return {"value": n_q / dt, "ms_per_step": dt * 1e3, "tile_pruning": "on" if pruning_on else "off (self-sample of the database)",
            "self_sample_live_share": self_live, "value_pruned": n_q / dt_pruned, "ms_per_step_pruned": dt_pruned * 1e3,
            "value_unpruned": n_q / dt_full, "ms_per_step_unpruned": dt_full * 1e3, "steps": steps,
            "queries": n_q, "refs": len(h.lineages), "tiles": ntiles, "classified_ok": ok, "parity_sample": parity, "stage_ms_per_step": stages, "stage_ms_per_step_pruned": stages_pruned,
            "stage_ms_per_step_unpruned": stages_full,
            "live_tiles_per_pair": st["live_tiles_per_pair"], "live_tiles_per_query": st.get("live_tiles_per_query"),
            "tiles_above_threshold_per_query": st["tiles_above_threshold_per_query"], "mean_threshold": st["mean_threshold"],
            "mean_best_hit_lower_bound": st["mean_best_hit_lower_bound"], "share_with_threshold": st["queries_with_threshold"] / n_q,
            "what": "value: the handle as created (tile pruning as its self-sample decided); value_pruned: pruning forced on; value_unpruned: pruning off; "
                    "live tiles, thresholds: of the forced pruned run",
            "workload": f"{h.n_records_held_out} held-out Diptera COI records (~205 bp) as {n_q} queries vs the other {h.n_records_db} records x 16 "
                        f"individual-level copies = {len(h.lineages)} references (scripts/common.py:11-25 hold-out methodology)"}


def mixed_lengths_block(args, rx, lib, index, db, flags):
    """value_mixed_lengths: 131 072 COI reads alone, then the same reads with ten reads of 1 100 .. 8 000 bases scattered among them -- the
    library cuts a batch into length classes (include/raxtax_hip.h: rtx_batch_classes), so the outliers must not move the barcodes off
    the pair kernel, the memoised tables and the tile pruning (until round 4 the longest query decided for the whole batch)."""
    from raxtax_amd import synth

    n = 131072
    L = db.length
    q = synth.make_queries(db, n, seed=77)
    rng = np.random.default_rng(78)
    refs = db.seq_bytes.reshape(db.n, L)
    longs = []
    for nb in (1100, 1500, 2200, 3000, 4100, 4500, 6000, 6500, 7000, 8000):
        s = np.concatenate([refs[int(i)] for i in rng.integers(0, db.n, nb // L + 1)])[:nb].copy()
        hit = rng.random(nb) < 0.02
        s[hit] = (1 << rng.integers(0, 4, int(hit.sum()))).astype(np.uint8)
        longs.append(s)
    at = np.sort(rng.integers(0, n, len(longs)))
    parts, prev = [], 0
    for a, s in zip(at, longs):
        parts.append(q.bases[prev * L:int(a) * L])
        parts.append(s)
        prev = int(a)
    parts.append(q.bases[prev * L:])
    mixed = np.concatenate(parts)
    lens = np.full(n + len(longs), L, np.uint64)
    for k, (a, s) in enumerate(zip(at, longs)):
        lens[int(a) + k] = len(s)
    moff = np.zeros(n + len(longs) + 1, np.uint64)
    moff[1:] = np.cumsum(lens)
    steps = 3

    def timed():
        index.run(flags)
        index.download(copy=False)
        t0 = time.perf_counter()
        for _ in range(steps):
            index.run(flags)
            index.download(copy=False)
        return (time.perf_counter() - t0) / steps
    index.upload(q.bases, q.base_off)
    dt_pure = timed()
    index.upload(mixed, moff)
    dt_mixed = timed()
    classes = index.batch_classes()
    view = index.download(copy=False)
    ok = int((np.ctypeslib.as_array(view.status, shape=(len(lens),)) == 0).sum())
    return {"value": (n + len(longs)) / dt_mixed, "ms_per_step": dt_mixed * 1e3, "value_coi_alone": n / dt_pure, "ms_per_step_coi_alone": dt_pure * 1e3,
            "slowdown_of_the_batch": dt_mixed / dt_pure, "queries": n + len(longs), "long_reads": [len(s) for s in longs], "classified_ok": ok,
            "classes": classes, "steps": steps,
            "what": "131 072 COI reads alone / with ten reads of 1.1 .. 8 kb among them (one batch): length classes keep the barcodes on their path"}


def long_reads_block(args, rx, lib, flags, ctx=None):
    """value_long_reads: full-length 16S-like reads (1 500 bases, t ~ 1 490: the SINTAX use case the reference's README cites) against a
    database of as many references as the headline's, same phylo model.  Since round 6 these queries have a length class of their own
    (t <= 2047): eleven bit planes on the pair kernel, tile pruning, the memoised tables (until then: one query per wave, every tile,
    the recurrence kernel -- 0.26 M reads/s).  `parity_sample`: the leg's last step against the oracle, as run."""
    from raxtax_amd import synth

    n_refs, n_q, L = args.refs, 65536, 1500
    db = synth.make_db(n_refs, length=L)
    qs = synth.make_queries(db, n_q, seed=5)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)
    index = rx.Index(tree, device=0, stage_timing=True)
    index.upload(qs.bases, qs.base_off)
    steps = 3
    index.run(flags)
    index.download(copy=False)
    t0 = time.perf_counter()
    for _ in range(steps):
        index.run(flags)
        view = index.download(copy=False)
    dt = (time.perf_counter() - t0) / steps
    ok = int((np.ctypeslib.as_array(view.status, shape=(n_q,)) == 0).sum())
    stages = {s: round(ms, 2) for s, (ms, n) in index.stage_times().items() if n}
    parity = leg_parity(rx, index, view, ctx, db.lineages, db.seq_bytes, db.seq_off, qs.bases, qs.base_off, flags, 200)
    return {"value": n_q / dt, "ms_per_step": dt * 1e3, "queries": n_q, "refs": n_refs, "query_len": L, "classified_ok": ok, "steps": steps,
            "parity_sample": parity, "classes": index.batch_classes(), "stage_ms_per_step": stages,
            "what": f"{n_q} synthetic reads of {L} bases (phylo model, 2 % from their source) vs {n_refs} references of {L} bases"}


# ------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.emit_fasta:
        raise SystemExit(emit_fasta(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if args.host_cpus:
        os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:args.host_cpus])
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the device path)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()     # smoke test: several ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dist = None
    coll_device = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == world

    import raxtax_amd as rx
    from raxtax_amd import dist_util, synth

    # ---- inputs (untimed): identical database on every rank
    db = synth.make_db(args.refs)
    flags = rx.RTX_SKIP_EXACT_MATCHES if args.skip_exact_matches else 0
    lib = rx._lib.load()
    if args.host_share:
        rx._lib.check(lib.rtx_set_host_share(args.host_share))
    L = db.length

    if args.shard_db:
        from raxtax_amd import sharded

        tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)          # the shards are cut out of Tree.k_mer_map
        qs = synth.make_queries(db, args.queries, seed=3)                        # the same queries on every rank
        if args.shard_mode == "kmers":
            kcuts = sharded.kmer_cuts(tree.csr()[0], world)
            index = sharded.KmerShardIndex(tree, rank, kcuts, device=local_rank, sub_batch=args.sub_batch or 256)
            comm = sharded.TorchComm(dist, world, []) if dist is not None else sharded.LocalComm()
            clf = sharded.KmerShardedClassifier([index], comm)
        else:
            cuts = sharded.shard_cuts(tree.num_tips, world)
            index = sharded.ShardIndex(tree, rank, cuts, device=local_rank, sub_batch=args.sub_batch or 4096)
            if dist is not None:
                w = torch.tensor([index.n_bnd_local], device=coll_device, dtype=torch.int64)
                ws = [torch.zeros_like(w) for _ in range(world)]
                dist.all_gather(ws, w)
                comm = sharded.TorchComm(dist, world, [int(x.item()) for x in ws])
            else:
                comm = sharded.LocalComm()
            clf = sharded.ShardedClassifier([index], comm)
        t0 = time.perf_counter()
        ex_ids, ex_off = index.exact_matches(qs.bases, qs.base_off)
        t_exact = time.perf_counter() - t0
        clf.upload(qs.bases, qs.base_off, ex_ids, ex_off)                         # inputs resident from here on
        if not args.hit_events_only:
            rx._lib.check(lib.rtx_index_set_option(index._h, 6, 1))

        def step():
            return clf.run(skip_exact_matches=bool(flags), copy=False)

        def finish():
            pass
        total_q_step = args.queries                                              # strong scaling: the work is fixed
        scaling = "strong"
        parallelism = (f"references sharded x{world} (contiguous id ranges), queries replicated; all-reduce of histograms + all-gather of prefix sums"
                       if args.shard_mode == "refs" else
                       f"k-mers sharded x{world}, queries replicated; all-reduce of the u16 per-reference hit counts")
        workload = (f"{args.queries} synthetic COI-length (658 bp) queries vs {args.refs}-seq reference DB sharded by "
                    f"{'reference id' if args.shard_mode == 'refs' else 'k-mer (every rank holds the posting lists of a k-mer range over all references)'} "
                    f"over {world} GPU(s) (BASELINE.json configs[4] shape)")
    else:
        qs = synth.make_queries(db, args.queries, seed=3 + rank, first_label=rank * args.queries, mu_q=args.mu_q, exact_frac=args.exact_frac)   # rank-specific queries
        tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off, kmer_map=False)   # bitmaps built on the GPU
        index = rx.Index(tree, device=local_rank, sub_batch=args.sub_batch,
                         stage_timing=not args.hit_events_only, cluster=False if args.no_cluster else None,
                         packed_counts=False if args.u16_counts else None,
                         tile_skip=False if args.no_tile_skip else None, hit_pair=False if args.no_pair else None,
                         locator=False if args.no_locator else None, tile_prune=False if args.no_tile_prune else None,
                         fine_bounds=False if args.no_fine_bounds else None, records=args.records, overlap=args.overlap,
                         two_level=0 if args.no_two_level else args.two_level_rule)
        t_exact = None
        if args.host_exact_match or not index.has_exact_lookup:
            t0 = time.perf_counter()
            ex_ids, ex_off = index.exact_matches(qs.bases, qs.base_off)   # Tree.sequences.get, raxtax.rs:42 (host, untimed)
            t_exact = time.perf_counter() - t0
            index.upload(qs.bases, qs.base_off, ex_ids, ex_off)            # inputs resident in HBM from here on
        else:
            index.upload(qs.bases, qs.base_off)                            # no ids: looked up on the device, every step

        # N > 1: packing and gathering the records of step i happen while the device classifies step i+1 (two sets of
        # buffers); those of the last step are completed inside the timed region
        rec_buf = [None, None]
        gather_cache = [{}, {}]
        pending = [None]
        step_no = [0]
        prev_view = [None]

        def ship(view):
            """Packs the result records of a finished step and starts their gather on rank 0 (the only collective: RCCL
            over xGMI); the gather started before is completed first (two sets of buffers alternate)."""
            k = step_no[0] & 1
            step_no[0] += 1
            need = lib.rtx_result_pack(ctypes.byref(view), None, 0)       # native pack: 25 B/query + (13 + depth) B/row
            if rec_buf[k] is None or rec_buf[k].shape[0] < need:
                rec_buf[k] = dist_util.pinned_bytes(int(need * 1.25) + 64)
            n = lib.rtx_result_pack(ctypes.byref(view), rec_buf[k].ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), rec_buf[k].shape[0])
            assert n == need, "rtx_result_pack failed"
            if pending[0] is not None:
                note_gathered(dist_util.gather_finish(pending[0]))
            pending[0] = dist_util.gather_start(dist, rec_buf[k][:n], rank, world, device=coll_device, cache=gather_cache[k])

        gathered_q = [0]
        last_parts = [None]

        def note_gathered(parts):
            """Rank 0: the record buffers of all ranks are in its (pinned) host memory; their headers say how many queries arrived."""
            if parts is not None:
                gathered_q[0] = sum(int(np.frombuffer(p_[:8].tobytes(), np.int64)[0]) for p_ in parts if len(p_) >= 32)
                last_parts[0] = parts     # (views of the staging buffers: the last gather of a run is not overwritten by a later one)

        def step():
            index.run(flags)                        # enqueues every kernel of this step
            if dist is not None and prev_view[0] is not None:
                ship(prev_view[0])                  # host work of the step before (its view stays valid until the second-next
                prev_view[0] = None                 # download) while the device classifies this one
            view = index.download(copy=False)       # streams the result records back + host finalisation
            prev_view[0] = view
            return view

        def finish():
            if dist is not None and prev_view[0] is not None:   # records and gather of the last step belong to the timed region
                ship(prev_view[0])
                prev_view[0] = None
            if pending[0] is not None:
                note_gathered(dist_util.gather_finish(pending[0]))
                pending[0] = None
        total_q_step = args.queries * world
        scaling = "weak"
        parallelism = f"queries sharded x{world}, index replicated"
        workload = (f"{args.queries} synthetic COI-length (658 bp) queries per GPU vs {args.refs}-seq reference DB replicated "
                    f"in HBM ({args.config_name})")
        if args.mu_q != 0.02 or args.exact_frac != 0.10:
            workload += f" -- NOT the headline queries: mu_q = {args.mu_q}, exact copies {args.exact_frac}"

    def barrier():
        finish()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    stage_ms = {s: 0.0 for s in rx._lib.STAGES}
    stage_n = {s: 0 for s in rx._lib.STAGES}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        view = step()
        for s, (ms, n) in index.stage_times().items():   # reads already-recorded HIP events
            stage_ms[s] += ms
            stage_n[s] += n
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    work = index.work()
    prob_work = index.prob_work()
    prune_stats = index.debug_prune_stats() if hasattr(index, "debug_prune_stats") else None   # of the last step (all zero: not pruned)
    ok = int((np.ctypeslib.as_array(view.status, shape=(args.queries,)) == 0).sum())
    parity_failed = False
    if rank == 0:
        line = {
            "metric": "classified queries/sec (whole node)",
            "value": total_q_step * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32 bit-planes + f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "refs": args.refs, "queries_per_gpu": args.queries, "query_len": L,
                "synthetic_data": "phylo model of SURVEY.md 8d, numpy PCG64 streams (raxtax_amd/synth.py)",
                "parallelism": parallelism,
                "process_group": ({"backend": dist.get_backend(), "world_size": dist.get_world_size()} if dist is not None
                                  else {"backend": None, "world_size": 1}),
                "classified_ok": ok, "skip_exact_matches": bool(args.skip_exact_matches),
                "host_threads_per_rank": int(lib.rtx_host_threads()),      # affinity mask capped by the cgroup quota, divided by the ranks on this host
                "host_share": args.host_share or int(os.environ.get("LOCAL_WORLD_SIZE", "1")),
                "hbm_bytes": {"index": int(index.device_bytes), "workspace": int(index.workspace_bytes) if hasattr(index, "workspace_bytes") else None,
                              "workspace_parts": index.workspace_parts() if hasattr(index, "workspace_parts") else None,
                              "note": "index: bitmaps, segment classes, union bitmaps, taxonomy; workspace: probability tables, two scratch sets of a sub-batch, inputs, result arrays"},
                "gathered_queries_last_step": (gathered_q[0] if (dist is not None and not args.shard_db) else None),
                "sub_batch": int(round(args.queries / max(stage_n["hit_count"] / args.steps, 1))) if stage_n["hit_count"] else None,
                "exact_match_lookup": ("host hash map, once, untimed: %.3f s" % t_exact) if t_exact is not None else
                                      "device (rtx_exact.hip), inside every timed step",
            },
            "roofline": roofline_block(args, work, prob_work, stage_ms, stage_n, args.queries, L, prune=prune_stats,
                                       ntiles=(args.refs + 8191) // 8192),
            "stage_ms_per_step": {s: stage_ms[s] / args.steps for s in stage_ms},
            # every kernel of a step sits in one of the stages (HIP events on the library's stream); what is left of ms_per_step is launch
            # gaps, the host's finalisation of the last sub-batch and the D2H of the records
            "stage_ms_sum_per_step": sum(stage_ms.values()) / args.steps,
            "unstaged_ms_per_step": elapsed / args.steps * 1e3 - sum(stage_ms.values()) / args.steps,
        }
        if not args.shard_db:
            line["step_bounds"] = step_bounds(args, line["stage_ms_per_step"], line["ms_per_step"], args.queries, L,
                                              pruned=bool(prune_stats and prune_stats.get("pairs")))
            line["stage_note"] = ("with RTX_OPT_OVERLAP (default) the back half of a sub-batch runs beside the front half of the next: the stage times "
                                  "overlap and their sum exceeds ms_per_step (unstaged_ms_per_step is then negative)")
        ctx = None
        if world > 1 and not args.no_parity:             # N > 1: the line verifies itself too (rank 0, untimed; the other ranks wait at the barrier)
            try:
                ctx_m = oracle_context(db)
                if args.shard_db:
                    line["parity_sample"] = sharded_parity_block(args, rx, index, view, ctx_m, db, qs, flags,
                                                                 (cuts[rank], cuts[rank + 1]) if args.shard_mode == "refs" else None)
                else:
                    line["parity_sample"] = multirank_parity_block(args, rx, index, view, ctx_m, db, qs, flags, world, last_parts[0])
            except Exception as e:
                line["parity_sample"] = {"ok": False, "error": f"{type(e).__name__}: {str(e)[:300]}"}
            parity_failed = not line["parity_sample"]["ok"]
        if args.shard_db and world == 1 and not args.no_cpu_baseline:
            try:
                line["parity_sample"] = sharded_parity_block(args, rx, index, view, oracle_context(db), db, qs, flags,
                                                             (cuts[rank], cuts[rank + 1]) if args.shard_mode == "refs" else None)
                parity_failed = not line["parity_sample"]["ok"]
            except Exception as e:
                line["parity_sample"] = {"ok": False, "error": f"{type(e).__name__}: {str(e)[:300]}"}
                parity_failed = True
        if not args.no_cpu_baseline and world == 1:      # the oracle: rank 0 at N = 1 only, never inside the timed region
            try:
                ctx = oracle_context(db)
            except Exception as e:                       # (no C compiler on the bench host, ...): the measured line is still printed
                line["parity_sample"] = {"ok": False, "error": f"oracle_context: {type(e).__name__}: {e}"}
                parity_failed = True
            if ctx is not None and not args.shard_db:    # first of all: the last timed step is still on the device
                line["parity_sample"] = parity_block(args, rx, index, view, ctx, db, qs, flags)
                parity_failed = not line["parity_sample"]["ok"]
        if not args.no_extras and world == 1 and not args.shard_db:
            line.update(extras_block(args, rx, lib, index, tree, db, qs, flags))
            line["value_mixed_lengths"] = mixed_lengths_block(args, rx, lib, index, db, flags)
            index.upload(qs.bases, qs.base_off)     # (the headline's batch again)
            if args.config == 2 and args.config_name != "custom size":
                line["value_real_composition"] = real_composition_block(args, rx, lib, flags, ctx)
                line["value_long_reads"] = long_reads_block(args, rx, lib, flags, ctx)
                for leg in ("value_real_composition", "value_long_reads"):     # a leg that does not hold against the oracle fails the line like the headline's sample
                    ps = line[leg].get("parity_sample") if isinstance(line[leg], dict) else None
                    if ps is not None and not ps["ok"]:
                        parity_failed = True
                        line["parity_sample"] = dict(line.get("parity_sample") or {}, ok=False, error=f"{leg}: {ps.get('error', '')}")
        line["cpu_baseline"] = cpu_baseline(db, qs, args.cpu_seconds, bool(flags), ctx) if ctx is not None else None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        print("bench.py: parity_sample failed: " + line["parity_sample"].get("error", ""), file=sys.stderr)
        raise SystemExit(4)


if __name__ == "__main__":
    main()
