#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel of the built library, read from the code-object metadata (no GPU needed).
    python tools/kernel_resources.py [pattern]"""
import os
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = "/opt/rocm/lib/llvm/bin/"
lib = os.environ.get("RTX_LIB_PATH") or str(Path(__file__).resolve().parent.parent / "raxtax_amd" / "libraxtax_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
data = open(lib, "rb").read()
rows, seen, pos = [], set(), 0
while True:
    i = data.find(b"\x7fELF\x02\x01\x01\x40", pos)   # ELF64, little endian, OS ABI 0x40 = AMDGPU HSA: an embedded code object
    if i < 0:
        break
    pos = i + 4
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[i:])
    txt = subprocess.run([LLVM + "llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    for b in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        b = ".agpr_count:" + b
        g = lambda k: (re.search(re.escape(k) + r"\s+(\S+)", b) or [None, "?"])[1]
        name = g(".name:")
        if name in seen or (pat and pat not in name):
            continue
        seen.add(name)
        rows.append((name, g(".vgpr_count:"), g(".agpr_count:"), g(".sgpr_count:"), g(".private_segment_fixed_size:"),
                     g(".group_segment_fixed_size:"), g(".vgpr_spill_count:")))
print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>7} {'spill':>6}  kernel")
for n, v, a, s, p, l, sp in sorted(rows):
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    if not pat and "rtx::" not in d:   # rocPRIM's sort kernels: only on request
        continue
    print(f"{v:>5} {a:>5} {s:>5} {p:>8} {l:>7} {sp:>6}  {d[:120]}")
