"""bench.py as the driver runs it, at toy sizes: the JSON contract (roofline + cpu_baseline present and physical),
`--gpus N` launching its own ranks (two gloo ranks sharing the one GPU of the test box), and the reference-sharded
mode (`--shard-db`, BASELINE.json configs[4])."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def bench(*args, timeout=600):
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), *map(str, args)], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_is_physical():
    r = bench("--refs", 20000, "--queries", 12000, "--steps", 2, "--warmup", 1, "--cpu-seconds", 1)
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["unit"] == "queries/s" and r["value"] > 0
    assert r["config"]["classified_ok"] == 12000
    roof = r["roofline"]
    assert roof["bound"] == "l2" and 0 < roof["frac"] <= 1.0 and roof["peak"] == 34500.0
    assert abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-12
    assert roof["algorithmic_ratio_to_hbm_peak"] > 0 and roof["requested_bytes_per_query"] > 0
    assert roof["traffic"] is None or roof["hbm_frac"] <= 1.0          # no PMC profile of this toy configuration
    assert 0 < roof["prob_stage"]["frac"] < 1.0 and roof["prob_stage"]["ops_prob_per_query"] > 1000
    assert all(v > 0 for k, v in r["stage_ms_per_step"].items() if k not in ("lineage_walk", "tile_bounds", "tile_prune"))   # the walk rides inside taxon_prefix; 3 tiles: no pruning
    # the caveats the line carries itself: queries crossing PCIe every step, end to end through rtx_raxtax to strings, every tile
    # counted, the sweep over the divergence of the queries
    assert 0 < r["value_incl_h2d"]["value"] and 0 < r["value_end_to_end"]["value"] and 0 < r["value_unpruned"]["value"]
    assert r["value_end_to_end"]["text_bytes_per_query"] > 50
    assert [x["mu_q"] for x in r["divergence_sweep"]["rows"]] == [0.02, 0.05, 0.1, 0.15] and all(x["value"] > 0 for x in r["divergence_sweep"]["rows"])
    assert "device" in r["config"]["exact_match_lookup"] and r["stage_ms_per_step"]["exact_match"] > 0
    cpu = r["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["one_thread"]["value"] > 0
    assert "cpu_model" in cpu and cpu["unit"] == "queries/s"


def test_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: two ranks (gloo, sharing the GPU), the gather of the result
    records inside the timed region, n_gpus = 2 on the line (the parent exits non-zero otherwise)."""
    r = bench("--gpus", 2, "--backend", "gloo", "--refs", 5000, "--queries", 3000, "--steps", 2, "--warmup", 1)
    assert r["n_gpus"] == 2 and r["config"]["process_group"] == {"backend": "gloo", "world_size": 2}
    assert r["scaling"] == "weak" and r["cpu_baseline"] is None
    assert r["config"]["classified_ok"] == 3000
    # the line verifies itself at N > 1 too: rank 0's own last sub-batch as the run left it, and a seeded sample of the records it
    # gathered from the other rank, against the oracle
    ps = r["parity_sample"]
    assert ps["ok"] and ps["ranks_checked"] == [0, 1] and ps["n"] >= 400, ps


def test_shard_db_mode_runs():
    r = bench("--shard-db", "--refs", 2000, "--queries", 256, "--steps", 1, "--warmup", 1, "--no-cpu-baseline")
    assert r["scaling"] == "strong" and r["config"]["classified_ok"] == 256
    assert r["roofline"] is not None and r["roofline"]["launch_ms"] > 0
    r2 = bench("--shard-db", "--gpus", 2, "--backend", "gloo", "--refs", 3000, "--queries", 300, "--steps", 1, "--warmup", 1,
               "--sub-batch", 64)   # several sub-batches: the pipelined exchange (histogram all-reduce beside the next count)
    assert r2["n_gpus"] == 2 and r2["config"]["classified_ok"] == 300
    assert r2["parity_sample"]["ok"] and r2["parity_sample"]["n"] == 300, r2["parity_sample"]      # final rows of every query = the oracle's on the whole database
    r3 = bench("--shard-db", "--shard-mode", "kmers", "--gpus", 2, "--backend", "gloo", "--refs", 3000, "--queries", 300,
               "--steps", 1, "--warmup", 1, "--sub-batch", 64)
    assert r3["n_gpus"] == 2 and r3["config"]["classified_ok"] == 300 and "k-mers sharded" in r3["config"]["parallelism"]
    assert r3["parity_sample"]["ok"], r3["parity_sample"]


def test_three_ranks_rehearse_the_eight_gpu_runs():
    """What the driver's 8-GPU runs will do that two ranks do not show: a world size that is not a power of two and divides neither the
    references nor the tiles, every rank with its own thread budget (LOCAL_WORLD_SIZE -> rtx_host_threads), rank 0 gathering from several,
    shards that disagree on whether they could prune.  Three gloo ranks share the one GPU of the test box: the box's process guard allows
    six processes on its card -- a five-rank version of this test was killed by it (the ranks, this pytest process and the launcher
    count) -- which is why the rehearsal has three ranks and not eight.  configs[3] (queries sharded) and both modes of configs[4]."""
    r = bench("--gpus", 3, "--backend", "gloo", "--refs", 6000, "--queries", 1111, "--steps", 2, "--warmup", 1)
    assert r["n_gpus"] == 3 and r["config"]["process_group"] == {"backend": "gloo", "world_size": 3}
    assert r["config"]["classified_ok"] == 1111 and r["config"]["host_threads_per_rank"] >= 1
    assert r["config"]["gathered_queries_last_step"] == 3 * 1111        # rank 0 read the headers of all three record buffers
    assert r["parity_sample"]["ok"] and r["parity_sample"]["ranks_checked"] == [0, 1, 2], r["parity_sample"]
    r2 = bench("--shard-db", "--gpus", 3, "--backend", "gloo", "--refs", 73730, "--queries", 333, "--steps", 1, "--warmup", 1,
               "--sub-batch", 128)      # 73730 = 3 * 24576 + 2 references: two shards of 4 tiles (could prune alone), one of 3 (cannot): the
                                        # ranks agree not to (ShardedClassifier._agree_on_pruning over gloo) instead of mixing two orders of the queries
    assert r2["n_gpus"] == 3 and r2["config"]["classified_ok"] == 333 and r2["parity_sample"]["ok"], r2.get("parity_sample")
    r3 = bench("--shard-db", "--shard-mode", "kmers", "--gpus", 3, "--backend", "gloo", "--refs", 3001, "--queries", 257,
               "--steps", 1, "--warmup", 1, "--sub-batch", 64)
    assert r3["n_gpus"] == 3 and r3["config"]["classified_ok"] == 257 and r3["parity_sample"]["ok"], r3.get("parity_sample")


def test_rccl_backend_with_one_rank():
    """The RCCL code path of bench.py at N > 1 (device tensors, pinned staging, asynchronous gather overlapped with the next step's
    kernels on the library's stream) with a process group of one rank -- all of it that one GPU can exercise (tools/nccl_world1_check.py)."""
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "nccl_world1_check.py")], capture_output=True, text=True, timeout=600,
                       env={**__import__("os").environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert p.returncode == 0 and "nccl world-1 gather ok" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
