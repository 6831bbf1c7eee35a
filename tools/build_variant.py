#!/usr/bin/env python3
"""Builds the library with extra compiler flags into gpurun_scratch/lib_<name>.so (travels to the GPU box; RTX_LIB_PATH selects it):
    python tools/build_variant.py <name> [flags...]        e.g.  tools/build_variant.py t16 -DRTX_PRUNE_TURNS=16"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from raxtax_amd import _build  # noqa: E402

import os  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
# RTX_VARIANT_CSRC: another copy of the sources (e.g. `git worktree add gpurun_scratch/head HEAD` for a before / after pair)
CSRC = Path(os.environ.get("RTX_VARIANT_CSRC", _build.CSRC))
out_dir = ROOT / "gpurun_scratch"
obj_dir = out_dir / f"obj_{name}"
obj_dir.mkdir(parents=True, exist_ok=True)
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{CSRC}"] + extra


def one(src):
    obj = obj_dir / (src + ".o")
    # the compiler's diagnostics (register spills are what the experiments care about) go to a log beside the object; shown on failure
    log = obj_dir / (src + ".log")
    with open(log, "w") as fh:
        rc = subprocess.call([_build._hipcc()] + flags + ["-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)], stderr=fh)
    if rc:
        sys.stderr.write(log.read_text())
        raise SystemExit(f"{src}: hipcc failed ({rc}); diagnostics in {log}")
    return str(obj)


with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(one, _build.SOURCES))
lib = out_dir / f"lib_{name}.so"
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", str(lib)] + objs + ["-lpthread"])
print(lib)
