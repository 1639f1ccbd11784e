import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library, built in-tree; parity tests call through it."""
    import cmbpo_amd  # noqa: F401
    from cmbpo_amd import _lib
    return _lib.lib()


@pytest.fixture
def ens_path(hip_lib, request):
    """Selects the matrix path of the 512-wide ensemble forward for one test and restores the default afterwards."""
    from cmbpo_amd import _lib
    before = hip_lib.cmbpo_get_ens_matrix_path()
    _lib.check(hip_lib.cmbpo_set_ens_matrix_path(request.param), "cmbpo_set_ens_matrix_path")
    hip_lib.cmbpo_set_ens_f16_min_rows(0)          # the f16 path at every size, also below its default threshold
    yield request.param
    hip_lib.cmbpo_set_ens_matrix_path(before)
    hip_lib.cmbpo_set_ens_f16_min_rows(0)
