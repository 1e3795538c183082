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
    return sorted(set(re.findall(r"(?:\bint|\bsize_t|const\s+char\s*\*)\s+(lob_\w+)\s*\(", text)))


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


def test_python_constants_match_the_header():
    """The flag bits OR-ed into `act` and the variant indices are restated in Python (ops.py, _lib.py): they must be the
    header's values, every LOB_VAR_* must have a Python name, and the library must know as many variants as the header."""
    from lstm_ode_bci_amd import _lib, ops
    text = open(os.path.join(ROOT, "include", "lob.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(LOB_\w+)\s+(0x[0-9a-fA-F]+|\d+)\b", text)}
    assert ops.OUT_BF16 == defs["LOB_OUT_BF16"] and ops.DY_BF16 == defs["LOB_DY_BF16"] and ops.X_BF16 == defs["LOB_X_BF16"]
    assert ops.LN_IDENTITY == defs["LOB_LN_IDENTITY"]
    flags = [defs[k] for k in ("LOB_OUT_BF16", "LOB_DY_BF16", "LOB_X_BF16", "LOB_LN_IDENTITY")]
    assert len(set(flags)) == 4 and all(f & 0xff == 0 for f in flags)          # distinct, clear of the activation code
    header_vars = {k[len("LOB_VAR_"):]: v for k, v in defs.items() if k.startswith("LOB_VAR_") and k != "LOB_VAR_COUNT"}
    assert header_vars == _lib.VAR, (sorted(header_vars.items()), sorted(_lib.VAR.items()))
    assert sorted(header_vars.values()) == list(range(defs["LOB_VAR_COUNT"]))
    L = _lib.lib()
    assert L.lob_debug_get_variant(defs["LOB_VAR_COUNT"] - 1) >= 0
    assert L.lob_debug_set_variant(defs["LOB_VAR_COUNT"], 1) < 0                # out of range: argument error
    assert L.lob_version() == defs["LOB_VERSION"]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or
                    not os.path.exists(os.path.join(ROOT, "tools", "asan_abi_check.py")),
                    reason="needs hipcc and the sanitizer tool (CPU container only: the tool is in .gpurunignore)")
def test_argument_layer_under_address_sanitizer():
    """SURVEY.md section 5 (optional sanitizer pass): the HOST code of every entry point, built with the host-side
    address sanitizer, rejects bad arguments without touching memory it should not (tools/asan_abi_check.py; a GPU
    sanitizer build is not available on this pool -- the kernels are covered by the parity tests -- so the tool does
    not travel to the GPU box)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_abi_check.py")], capture_output=True, text=True,
                       timeout=900)
    if r.returncode == 77:
        pytest.skip("ASan runtime not installed")
    assert r.returncode == 0 and "asan_abi_check: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
