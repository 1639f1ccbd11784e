"""Error quantiles of the three matrix paths of the 512-wide ensemble forward against float64 on the wide-range cases of
tests/test_f16_range_gpu.py (per (member, row), relative to the row's output scale).
    python tools/probe_f16_range.py > profiles/r03/f16_range.json"""
import importlib.util
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_f16_range_gpu.py"))
t = importlib.util.module_from_spec(spec)
spec.loader.exec_module(t)
import cmbpo_amd  # noqa: E402,F401
from cmbpo_amd import _lib, synthetic  # noqa: E402

lib = _lib.lib()
out = {}


def summarise(errs):
    r = {}
    for name, (em, el) in errs.items():
        r[name] = {"mean": {"p50": float(np.percentile(em, 50)), "p99": float(np.percentile(em, 99)), "max": float(em.max()),
                            "max_per_member": [float(x) for x in em.max(axis=1)]},
                   "logvar": {"p50": float(np.percentile(el, 50)), "p99": float(np.percentile(el, 99)), "max": float(el.max())}}
    ratio = errs["splitf16"][0] / np.maximum(errs["fp32mfma"][0], t.FLOOR)
    r["row_ratio_f16_over_fp32"] = {"p50": float(np.percentile(ratio, 50)), "p99": float(np.percentile(ratio, 99)), "max": float(ratio.max())}
    return r


for kind in ("loguniform6", "lognormal_tail"):
    for task in ("AntSafe-v2", "HumanoidSafe-v2"):
        rng = np.random.default_rng(zlib.crc32(f"{kind}/{task}".encode()))
        D, A = synthetic.ENV_DIMS[task]
        E, I, O = 7, D + A, D + 1
        ws, bs = t._wide_weights(rng, kind, E, I, 512, 2 * O)
        sc_in, sc_out = synthetic.scaler(rng, I), synthetic.scaler(rng, O)
        pe = t._pe(E, I, O, ws, bs, sc_in, sc_out)
        x = rng.standard_normal((777, I)).astype(np.float32)
        with np.errstate(all="ignore"):
            rm, rl = t.f64_forward(x, ws, bs, sc_in, sc_out)
        out[f"{kind}/{task}"] = summarise(t.path_errors(lib, pe, x, rm, rl))
pe, xin, _ = t._trained_model(7, "AntSafe-v2", 220)
ws, bs = pe.get_weights()
sc_in = (pe.scaler_in.cached_mu, pe.scaler_in.cached_var)
sc_out = (pe.scaler_out.cached_mu, pe.scaler_out.cached_var)
x = xin[:1500].copy()
x[::11] *= np.float32(30.0)
rm, rl = t.f64_forward(x, ws, bs, sc_in, sc_out)
out["trained_220_adam_steps"] = summarise(t.path_errors(lib, pe, x, rm, rl))
out["trained_220_adam_steps"]["weight_spread"] = [
    {"max_abs": float(np.abs(w).max()), "median_abs": float(np.median(np.abs(w))),
     "min_column_max_over_matrix_max": float((np.abs(w).max(axis=1) / np.abs(w).max(axis=(1, 2), keepdims=True)[:, 0]).min())}
    for w in ws]
print(json.dumps(out, indent=1))
