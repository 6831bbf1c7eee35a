"""End-to-end timing of the raxtax-hip CLI (FASTA in, .out/.tsv/.ckp out) on synthetic inputs.
Usage: python tools/cli_e2e.py [--refs N] [--queries Q] [--tsv] [--batch B]"""
import argparse, json, subprocess, sys, tempfile, time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from raxtax_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--refs", type=int, default=50000)
ap.add_argument("--queries", type=int, default=100000)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--tsv", action="store_true")
ap.add_argument("--repeat", type=int, default=2)
a = ap.parse_args()
db = synth.make_db(a.refs)
qs = synth.make_queries(db, a.queries)
tmp = Path(tempfile.mkdtemp(prefix="rtxcli"))
(tmp / "db.fasta").write_text(db.fasta())
letters = "ACGT"
import numpy as np
lut = np.zeros(256, np.uint8)
for code, ch in ((1, "A"), (2, "C"), (4, "G"), (8, "T"), (15, "N")):
    lut[code] = ord(ch)
with open(tmp / "q.fasta", "w") as f:
    for i in range(qs.n):
        f.write(f">q{i}\n{lut[qs.seq(i)].tobytes().decode()}\n")
for r in range(a.repeat + 1):
    # the last run takes the .bin cache the first one wrote as its database (Tree::load_from_file path)
    dbp = tmp / "db.fasta" if r < a.repeat else tmp / "out0" / "db.bin"
    cmd = [str(ROOT / "raxtax_amd" / "raxtax-hip"), "-d", str(dbp), "-i", str(tmp / "q.fasta"), "-o", str(tmp / f"out{r}"),
           "--timing"] + (["--tsv"] if a.tsv else []) + (["--batch", str(a.batch)] if a.batch else [])
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.time() - t0
    for l in p.stderr.splitlines():
        if l.startswith("[TIMING]"):
            print(l)
    last = [l for l in p.stderr.splitlines() if l.startswith("{")]
    print(json.dumps({"rc": p.returncode, "wall_s": round(wall, 3), "stages": json.loads(last[-1]) if last else p.stderr[-500:]}))
