"""Summarise the rocprofv3 --pmc passes of tools/collect_pmc_r03.sh into the two JSON files bench.py / DESIGN cite.
    python tools/summarise_pmc_h3.py gpurun_out/r03/<tag> profiles/r03
Reads every *counter_collection.csv under <tag>/pmc_fetch, pmc_write, pmc_sq, pmc_sq2; keeps the dispatches of ens_h3_kernel
(skipping each pass's first two forwards: warm-up), and writes pmc_traffic_h3.json (FETCH_SIZE / WRITE_SIZE per forward, with
the gfx950 correction of MI355X_MICROARCH.md's HBM section) and pmc_sq_ens_h3.json (SQ counters of the main launch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]


def load(sub):
    out = defaultdict(list)      # counter -> [(kernel name, value, duration ns)]
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ens_h3_kernel" in r["Kernel_Name"]:
                out[r["Counter_Name"]].append((r["Kernel_Name"], float(r["Counter_Value"]),
                                               int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Dispatch_Id"])))
    return out


def split(rows):
    """(main launches, tail launches) in dispatch order without the first two forwards"""
    rows = sorted(rows, key=lambda t: t[3])
    main = [t for t in rows if ", 4>" in t[0]][2:]
    tail = [t for t in rows if ", 4>" not in t[0]][2:]
    return main, tail


mean = lambda xs: sum(xs) / max(1, len(xs))
fetch, write = load("pmc_fetch"), load("pmc_write")
if fetch and write:
    fm, ft = split(fetch["FETCH_SIZE"])
    wm, wt = split(write["WRITE_SIZE"])
    f_kb = mean([t[1] for t in fm]) + mean([t[1] for t in ft])
    w_kb = mean([t[1] for t in wm]) + mean([t[1] for t in wt])
    us = (mean([t[2] for t in fm]) + mean([t[2] for t in ft])) / 1e3
    traffic = {
        "kernel": "ens_h3_kernel<3,2,4> + ens_h3_kernel<3,2,2> (the two launches of one 100 000-row forward)",
        "workload": "AntSafe-v2 B=100000 rows per forward (python3 tools/probe_h3.py 100000 4 AntSafe-v2 2: the f16 matrix path)",
        "collected": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_pmc_r03.sh + "
                     "tools/summarise_pmc_h3.py), means over %d forwards; per launch: main %.1f + tail %.1f KB fetched, %.1f + %.1f KB "
                     "written" % (len(fm), mean([t[1] for t in fm]), mean([t[1] for t in ft]), mean([t[1] for t in wm]),
                                  mean([t[1] for t in wt])),
        "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
        "fetch_bytes_corrected": 2.0 * f_kb * 1024.0, "write_bytes": w_kb * 1024.0,
        "hbm_bytes_per_launch": 2.0 * f_kb * 1024.0 + w_kb * 1024.0,
        "algorithmic_bytes_per_launch": 191690000.0,
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a 16-B/lane coalesced stream -> doubled "
                      "(MI355X_MICROARCH.md, HBM); WRITE_SIZE: the 168.0 MB of outputs are 120-byte rows (not line aligned) "
                      "written as 8-byte lane stores",
        "rows_per_launch": 100000, "kernel_us_under_pmc": us,
    }
    json.dump(traffic, open(os.path.join(dst, "pmc_traffic_h3.json"), "w"), indent=1)
    print("traffic: %.1f MB per forward (%.1f fetched x 2 + %.1f written), %.0f us under the counters"
          % (traffic["hbm_bytes_per_launch"] / 1e6, f_kb * 1024 / 1e6, w_kb * 1024 / 1e6, us))
sq = load("pmc_sq")
sq.update(load("pmc_sq2"))
if sq:
    o = {"kernel": "ens_h3_kernel<3,2,4> (the main launch of a 100 000-row forward: 5376 of its 5474 items)",
         "command": "tools/collect_pmc_r03.sh + tools/summarise_pmc_h3.py: two passes of tools/probe_h3.py 100000 6 AntSafe-v2 2"}
    dur = []
    for name, rows in sorted(sq.items()):
        m, _ = split(rows)
        o[name] = mean([t[1] for t in m])
        dur += [t[2] for t in m]
    if "SQ_INSTS_MFMA" in o and "SQ_VALU_MFMA_BUSY_CYCLES" in o:
        o["busy_cycles_per_mfma"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["SQ_INSTS_MFMA"]
        o["valu_per_mfma"] = o["SQ_INSTS_VALU"] / o["SQ_INSTS_MFMA"]
    o["duration_us_under_pmc"] = mean(dur) / 1e3
    if "GRBM_GUI_ACTIVE" in o:
        o["clock_ghz"] = o["GRBM_GUI_ACTIVE"] / (o["duration_us_under_pmc"] * 1e3) / 8.0   # (summed over the 8 XCDs)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in o:
            o["mfma_pipe_utilisation_under_pmc"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * o["GRBM_GUI_ACTIVE"] / 8.0)
    json.dump(o, open(os.path.join(dst, "pmc_sq_ens_h3.json"), "w"), indent=1)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in o.items() if k not in ("kernel", "command")})
