import os
import sys
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (raxtax_amd/__init__.py: before anything initialises HIP)

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/liboracle.so) -- the checker, never the thing under test in -m gpu."""
    from oracle.oracle_py import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def kats():
    import json

    return json.loads((ROOT / "tests" / "golden" / "reference_kats.json").read_text())


@pytest.fixture(scope="session")
def emul():
    """x86 build of the device arithmetic (raxtax_amd/csrc/rtx_emul.cpp over rtx_math.hpp): test support, never on the product path."""
    import ctypes as C
    import subprocess

    out = ROOT / "tests" / "_build" / "librtx_emul.so"
    out.parent.mkdir(exist_ok=True)
    src = ROOT / "raxtax_amd" / "csrc" / "rtx_emul.cpp"
    hdr = ROOT / "raxtax_amd" / "csrc" / "rtx_math.hpp"
    if not out.exists() or out.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", f"-I{src.parent}", "-o", str(out), str(src)])
    lib = C.CDLL(str(out))
    lib.emul_prob_table.restype = C.c_int
    return lib
