"""CPU: the C-ABI library builds, loads and exports every symbol include/cmbpo_hip.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cmbpo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cmbpo_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    import __graft_entry__ as g
    g.build()
    from cmbpo_amd import _lib
    lib = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 8
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cmbpo_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    for name in _lib.SIGNATURES:
        assert name in declared, f"{name} bound in _lib.py but not declared in the header"
    assert lib.cmbpo_version() >= 1
    assert lib.cmbpo_last_error() is not None


def test_bad_arguments_return_error_codes_without_a_gpu():
    """Argument validation happens before any HIP call, so it is checkable on a CPU-only box."""
    import ctypes as C
    from cmbpo_amd import _lib
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.cmbpo_mlp_create(C.byref(h), 7, 37, 300, 60, 0, 0) == -1      # hidden not 128/512
    assert b"hidden" in lib.cmbpo_last_error()
    assert lib.cmbpo_mlp_create(C.byref(h), 7, 37, 512, 61, 0, 0) == -1      # odd width for HEAD_PROB
    assert lib.cmbpo_mlp_create(C.byref(h), 2, 29, 128, 8, 1, 2) == -1       # policy head needs E == 1
    assert lib.cmbpo_ens_forward(None, None, 29, None, 8, None, None, 0, 0, None, None, None) == -1
    assert lib.cmbpo_fakeenv_post(9, 7, 29, 8, None, None, 0, None, None, None, None, None, 0,
                                  None, None, None, None, None, None, None, None) == -1
