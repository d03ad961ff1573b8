#!/bin/bash
# usage (on the GPU box): tools/pmc_mfma.sh <tag> <bench_kernels what...>  -- matrix-pipe occupancy of a kernel: one PMC pass
# (kernel-trace only) with SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / SQ_INSTS_MFMA / SQ_WAVES; the trace csv gives the duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_mfma -- python3 $R/tools/bench_kernels.py "$@" --reps 10 > $R/gpurun_out/pmc_${TAG}_mfma.log 2>&1
python3 - <<PY
import csv, glob, os, collections
d = "$R/gpurun_out/pmc_${TAG}_mfma"
cc = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
c = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    c[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in c.items():
    m = {n: sum(x) / len(x) for n, x in v.items()}
    if m.get("SQ_INSTS_MFMA", 0) < 1:
        continue
    us = sum(dur[k]) / len(dur[k])
    busy = m["SQ_BUSY_CU_CYCLES"]
    print(f"{k:60s} {us:8.1f} us  MFMA {m['SQ_INSTS_MFMA']:.0f}  pipe busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * busy):.3f} of busy CU cycles  "
          f"clock {busy / 256 / us / 1000:.2f} GHz (busy cycles per CU / duration)")
PY
