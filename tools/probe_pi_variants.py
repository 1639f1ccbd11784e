"""Dev probe: build variants of csrc/policy_update.hip (-D flags) into /tmp next to the in-tree objects and run tools/probe_pi.py
with each (a subprocess per variant).    python tools/probe_pi_variants.py N "name1:-DFLAG" "name2:" ..."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "constrained-model-based-policy-optimization_amd")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    so = sys.argv[2]
    sys.path.insert(0, ROOT)
    import cmbpo_amd  # noqa: F401
    from cmbpo_amd import _lib
    _lib.LIB_PATH = so
    sys.argv = ["probe_pi.py"] + sys.argv[3:]
    exec(open(os.path.join(ROOT, "tools", "probe_pi.py")).read())
    sys.exit(0)

N = sys.argv[1]
objs = [o for o in sorted(glob.glob(os.path.join(PKG, "csrc", "*.o"))) if not o.endswith("policy_update.o")]
for spec in sys.argv[2:]:
    name, _, flags = spec.partition(":")
    o, so = f"/tmp/policy_update_{name}.o", f"/tmp/libcmbpo_pi_{name}.so"
    src = os.path.join(PKG, "csrc", "policy_update.hip")
    if flags.startswith("@"):          # another copy of the source tree's policy files (tools/_build/<dir>/policy_update.hip)
        src, flags = os.path.join(ROOT, flags[1:], "policy_update.hip"), ""
    subprocess.check_call([HIPCC] + FLAGS + flags.split() + ["-c", src, "-o", o])
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", so])
    print(f"==== variant {name} [{flags}]", flush=True)
    subprocess.call([sys.executable, os.path.abspath(__file__), "--child", so, N])
