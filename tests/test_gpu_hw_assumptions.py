"""Hardware behaviour the kernels rely on, checked on the device itself."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_raw_buffer_bounds_check_includes_sgpr_offset(tmp_path):
    """prob_lookup reads table rows through ONE buffer descriptor per table with the row start in the SGPR offset and
    num_records = end of the wanted part of the row (rtx_prob_tables.hip: row_load_f64).  That clips a row at its
    saturation index only if the bounds check of a raw buffer load counts the SGPR offset -- true on gfx950."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    exe = tmp_path / "soffset_bounds"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-w", "-o", str(exe), str(ROOT / "tools" / "micro" / "soffset_bounds.hip")])
    out = subprocess.check_output([str(exe)], text=True, timeout=120)
    lines = [l for l in out.splitlines() if l.startswith("nrec")]
    assert len(lines) >= 5, out
    for l in lines:
        m = re.search(r"first zero lane (\d+) \(incl: (\d+), excl: (\d+)\)", l)
        assert m, l
        assert int(m.group(1)) == int(m.group(2)), l
