#!/usr/bin/env python3
"""Readable summary of a bench.py JSON line: python tools/summarize_bench.py file.json"""
import json
import sys

r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(f"value {r['value']} {r['unit']}  ms/step {r['ms_per_step']}  n_gpus {r['n_gpus']}  mAP {r.get('map_at_5000')}")
if "value_first_allocation" in r:
    print(f"value_first_allocation {r['value_first_allocation']}  {r['first_allocation']}")
if "roofline" in r:
    print("roofline", {k: v for k, v in r["roofline"].items() if k != "kernel"})
print("placement", r["config"].get("swt_output_placement"), " host syncs in timed steps:", r["config"].get("host_syncs_in_timed_steps"))
for k in r.get("kernels", []):
    print(f"{k['ms'] * 1000:10.1f} us  {k['frac']:.3f} of {k['peak']:g} {k['unit']:8s} {k['kernel'][:150]}")
for s in r.get("kernels_skipped", []):
    print("SKIPPED", s)
if r["config"].get("exchange"):
    ex = dict(r["config"]["exchange"])
    ranks = ex.pop("per_rank", [])
    print("exchange", ex)
    for x in ranks:
        print("  ", x)
if "cpu_baseline" in r:
    print("cpu_baseline", r["cpu_baseline"]["value"], r["cpu_baseline"]["unit"], "cores", r["cpu_baseline"]["cores"])
