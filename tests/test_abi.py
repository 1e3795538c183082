"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/lob.h
declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "lob.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"(?:\bint|const\s+char\s*\*)\s+(lob_\w+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from lstm_ode_bci_amd import build
    lib_path = build.build(force=False, verbose=False)
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.lob_version.restype = ctypes.c_int
    assert lib.lob_version() >= 200


def test_python_binding_covers_the_header():
    from lstm_ode_bci_amd import _lib
    assert sorted(_lib._SIGS) == _declared()


def test_argument_errors_without_a_gpu():
    """Argument validation happens before any HIP call, so it is testable on CPU."""
    from lstm_ode_bci_amd import _lib
    L = _lib.lib()
    assert L.lob_gemm_nt_f32(None, 1, None, 1, None, None, 1, 1, 1, 1, 0, None) == -1
    assert L.lob_softmax_rows_f32(None, None, 0, 0, None) == -1
    rates = (ctypes.c_double * 6)(*[0.1] * 6)
    assert L.lob_ode_rk4_f64(None, None, rates, 0.5, 10, 0.0, 10.0, 16, None, None, None, 4, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback anywhere: without liblob.so every entry into the product path raises LobError."""
    from lstm_ode_bci_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "no_such_liblob.so"))
    with pytest.raises(_lib.LobError, match="not built|not found"):
        _lib.lib()
