"""MI355X-native hot path for CMBPO (imagined ensemble rollout + CPO trust-region update).

The directory name follows the project naming rule and is not a valid Python
identifier; import it through the root-level alias module ``cmbpo_amd``
(``import cmbpo_amd``; ``from cmbpo_amd.fake_env import FakeEnv``).

Nothing is imported eagerly here: submodules load the HIP C-ABI library
(``libcmbpo_hip.so``) on first use and raise if it is missing -- there is no
CPU fallback in the product path.
"""

__all__ = ["__version__"]
__version__ = "0.1.0"
