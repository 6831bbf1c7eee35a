import sys
from pathlib import Path

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
