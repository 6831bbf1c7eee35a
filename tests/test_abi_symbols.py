"""The C-ABI library loads without a GPU and exports every symbol include/raxtax_hip.h declares."""
import ctypes
import re
from pathlib import Path

from raxtax_amd import _lib

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "raxtax_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", text)) - {"rtx_sender_fn"})


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"libraxtax_hip.so does not export {n}"
    # and the binding covers the whole header
    assert set(names) == set(_lib._SIGNATURES)


def test_abi_version_and_no_device_is_loud():
    lib = _lib.load()
    assert lib.rtx_abi_version() == 6
    if lib.rtx_device_count() == 0:
        # no CPU fallback: index creation must fail with RTX_ERR_NO_DEVICE
        import numpy as np
        import raxtax_amd as rx
        tree = rx.Tree.new(["a,b"], [np.array([1, 2, 4, 8, 1, 2, 4, 8, 1], np.uint8)])
        try:
            rx.Index(tree)
            raise AssertionError("Index() must not succeed without a GPU")
        except rx.RtxError as e:
            assert e.code == _lib.RTX_ERR_NO_DEVICE
