cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_clk -- python3 $R/tools/bench_kernels.py swt head --reps 5 > $R/gpurun_out/pmc_clk.log 2>&1
python3 - <<PY
import csv, glob, os, collections
d = "$R/gpurun_out/pmc_clk"
cc = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
c = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    c[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[r["Kernel_Name"][:50]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in c.items():
    m = {n: sum(x) / len(x) for n, x in v.items()}
    us = sum(dur[k]) / len(dur[k])
    if us < 20: continue
    print(f"{k:50s} {us:8.1f} us  busy CU cycles / 256 / us = {m.get('SQ_BUSY_CU_CYCLES', 0) / 256 / us / 1000:.2f} GHz   GRBM_GUI_ACTIVE / us = {m.get('GRBM_GUI_ACTIVE', 0) / us / 1000:.2f} GHz (raw {m.get('GRBM_GUI_ACTIVE', 0):.0f})")
PY
