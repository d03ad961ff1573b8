#!/usr/bin/env python3
"""gpurun_out/<tag>_* (tools/profile_round.sh <tag>) -> profiles/<out>_bench.json, <out>_bench_kernel_stats.csv,
<out>_pmc.csv (means per launch of the kernels of interest) and traffic_<round>.json.
usage: python tools/summarize_profiles.py r02b r02"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
KERNELS = {"swt": "k_swt_slide", "topk": "k_rank_window", "rankmap": "k_rank_window", "head": "k_head_front"}

shutil.copy(os.path.join(G, f"{tag}_bench.json"), os.path.join(P, f"{out}_bench.json"))
shutil.copy(glob.glob(os.path.join(G, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)[0],
            os.path.join(P, f"{out}_bench_kernel_stats.csv"))

rows, traffic = {}, {}
for what, kname in KERNELS.items():
    for d in sorted(glob.glob(os.path.join(G, f"{tag}_pmc_{what}_*"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            for r in csv.DictReader(open(f)):
                if kname in r["Kernel_Name"]:
                    key = (what, r["Kernel_Name"].split("(")[0].replace("void ", ""))
                    a = acc.setdefault((key, r["Counter_Name"]), [0.0, 0])
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
            for (key, cname), (tot, n) in acc.items():
                rows.setdefault(key, {})[cname] = tot / n
cols = sorted({c for v in rows.values() for c in v})
with open(os.path.join(P, f"{out}_pmc.csv"), "w") as fh:
    w = csv.writer(fh)
    w.writerow(["run", "kernel"] + cols)
    for (what, k), v in sorted(rows.items()):
        w.writerow([what, k] + [v.get(c, "") for c in cols])

bench = json.loads(open(os.path.join(P, f"{out}_bench.json")).read().strip().splitlines()[-1])
alg = {"k_swt_slide": bench["roofline"].get("algorithmic_bytes") or 5240782848}
traffic["_how"] = ("rocprofv3 --pmc <one counter set per pass> --kernel-trace (tools/profile_round.sh); FETCH_SIZE / WRITE_SIZE "
                   "are in KiB per launch (mean over the launches of the run); FETCH_SIZE is doubled as MI355X_MICROARCH.md "
                   "prescribes for gfx950 streaming reads")
for (what, k), v in sorted(rows.items()):
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        name = {"swt": "k_swt_slide", "topk": "k_rank_window", "rankmap": "k_rank_window_ap", "head": "k_head_front"}[what]
        if name in traffic:
            continue
        traffic[name] = {"fetch_kib_raw": round(v["FETCH_SIZE"], 3), "write_kib": round(v["WRITE_SIZE"], 3),
                         "hbm_bytes_per_launch": int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)}
traffic["k_swt_slide"]["algorithmic_bytes"] = 5240782848
traffic["k_rank_window"]["algorithmic_bytes"] = 51416384
json.dump(traffic, open(os.path.join(P, f"traffic_{out}.json"), "w"), indent=1)
print(open(os.path.join(P, f"{out}_pmc.csv")).read())
print(json.dumps(traffic, indent=1))
