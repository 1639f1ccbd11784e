"""Dev probe: build variants of csrc/ens_h3.hip (-D flags) into /tmp next to the in-tree objects and time the 512-wide
f16 ensemble forward with each, in a subprocess per variant (accuracy against the float64 evaluation of probe_h3.py first).
    python tools/probe_h3_variants.py B "name1:-DFLAG_A -DFLAG_B" "name2:" ...
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "constrained-model-based-policy-optimization_amd")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize"]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    so, B, iters = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    sys.path.insert(0, ROOT)
    import cmbpo_amd  # noqa: F401
    from cmbpo_amd import _lib
    _lib.LIB_PATH = so
    sys.argv = ["probe_h3.py", str(B), str(iters), "AntSafe-v2", "2"]
    exec(open(os.path.join(ROOT, "tools", "probe_h3.py")).read())
    sys.exit(0)

B = sys.argv[1]
objs = [o for o in sorted(glob.glob(os.path.join(PKG, "csrc", "*.o"))) if not o.endswith("ens_h3.o")]
for spec in sys.argv[2:]:
    name, _, flags = spec.partition(":")
    o, so = f"/tmp/ens_h3_{name}.o", f"/tmp/libcmbpo_{name}.so"
    subprocess.check_call([HIPCC] + FLAGS + flags.split() + ["-c", os.path.join(PKG, "csrc", "ens_h3.hip"), "-o", o])
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", so])
    print(f"==== variant {name} [{flags}]", flush=True)
    subprocess.call([sys.executable, os.path.abspath(__file__), "--child", so, B, "30"])
