"""Importable alias for the package directory ``constrained-model-based-policy-optimization_amd``.

``import cmbpo_amd`` returns that package object registered under the name
``cmbpo_amd``, so its submodules load exactly once as ``cmbpo_amd.<name>``.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "constrained-model-based-policy-optimization_amd")
_spec = importlib.util.spec_from_file_location(
    "cmbpo_amd", os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["cmbpo_amd"] = _pkg
_spec.loader.exec_module(_pkg)
