"""Builds raxtax_amd/libraxtax_hip.so in-tree with hipcc for gfx950 (MI355X).

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with the
repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libraxtax_hip.so"
SOURCES = ["rtx_kernels.hip", "rtx_hit_pair.hip", "rtx_prob_tables.hip", "rtx_cluster.hip", "rtx_segments.hip", "rtx_prune.hip", "rtx_bounds2.hip", "rtx_records.hip", "rtx_finalise.hip", "rtx_exact.hip", "rtx_ingest.hip", "rtx_api_index.hip", "rtx_api_batch.hip", "rtx_api_download.hip", "rtx_api_shard.hip", "rtx_api_debug.hip", "host_tree.cpp", "host_format.cpp", "host_raxtax.cpp", "host_bin.cpp", "host_threads.cpp"]
HEADERS = ["rtx_index.hpp", "rtx_kernels.hpp", "rtx_hit_common.hpp", "rtx_internal.hpp", "rtx_math.hpp", "rtx_wave.hpp", "rtx_walk.hpp", "host_raxtax.hpp"]
CLI = PKG / "raxtax-hip"
SYNTH = PKG / "raxtax-synth"   # the generator of the synthetic inputs as a host program: csrc/synth_main.cpp


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libraxtax_hip.so)")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps if Path(d).exists())


OBJ = PKG / "_obj"   # per-source objects (git-ignored): sources compile in parallel and only when they changed


def build_lib(force: bool = False, verbose: bool = False) -> Path:
    from concurrent.futures import ThreadPoolExecutor

    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    hdrs = [CSRC / h for h in HEADERS] + [ROOT / "include" / "raxtax_hip.h"]
    if not force and not _stale(LIB, srcs + hdrs):
        return LIB
    OBJ.mkdir(exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{CSRC}"] + os.environ.get("RTX_EXTRA_CXXFLAGS", "").split()

    def compile_one(src: Path) -> Path:
        obj = OBJ / (src.name + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [_hipcc()] + flags + ["-x", "hip", "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", str(LIB)] + [str(o) for o in objs] + ["-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_cli(force: bool = False) -> Path | None:
    main = CSRC / "cli_main.cpp"
    if not main.exists():
        return None
    build_lib(force=force)
    if not force and not _stale(CLI, [main, LIB]):
        return CLI
    cmd = [_hipcc(), "-O2", "-std=c++17", f"-I{ROOT / 'include'}", f"-I{CSRC}", "-o", str(CLI), str(main),
           f"-L{PKG}", "-lraxtax_hip", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lz"]
    subprocess.check_call(cmd)
    return CLI


def build_synth(force: bool = False) -> Path:
    main = CSRC / "synth_main.cpp"
    if force or _stale(SYNTH, [main]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(SYNTH), str(main)])
    return SYNTH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    print(build_cli(force=True))
    print(build_synth(force=True))
