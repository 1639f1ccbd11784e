"""Dev probe: build variants of ONE csrc file (-D flags) into /tmp next to the in-tree objects and run a probe script with each.
    python tools/probe_variants.py <file.hip> <tools/probe_x.py> "<probe args>" "name1:-DFLAG" "name2:" ..."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "constrained-model-based-policy-optimization_amd")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    so, probe = sys.argv[2], sys.argv[3]
    sys.path.insert(0, ROOT)
    import cmbpo_amd  # noqa: F401
    from cmbpo_amd import _lib
    _lib.LIB_PATH = so
    sys.argv = [probe] + sys.argv[4:]
    exec(open(os.path.join(ROOT, probe)).read())
    sys.exit(0)

src, probe, pargs = sys.argv[1], sys.argv[2], sys.argv[3].split()
stem = src[:-4]
objs = [o for o in sorted(glob.glob(os.path.join(PKG, "csrc", "*.o"))) if not o.endswith("/" + stem + ".o")]
extra = ["-fno-slp-vectorize"] if stem == "ens_h3" else []
for spec in sys.argv[4:]:
    name, _, flags = spec.partition(":")
    o, so = f"/tmp/{stem}_{name}.o", f"/tmp/libcmbpo_{stem}_{name}.so"
    subprocess.check_call([HIPCC] + FLAGS + extra + flags.split() + ["-c", os.path.join(PKG, "csrc", src), "-o", o])
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", so])
    print(f"==== variant {name} [{flags}]", flush=True)
    subprocess.call([sys.executable, os.path.abspath(__file__), "--child", so, probe] + pargs)
