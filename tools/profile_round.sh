#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>   -- bench line, rocprof kernel stats of the same command,
# and FETCH_SIZE / WRITE_SIZE / VALU counters of the SWT kernel (each PMC set in its own pass, kernel-trace only)
set -e
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 30 --warmup 5 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
tail -c 600 $R/gpurun_out/${TAG}_bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_$n -o pmc -- python3 $R/tools/bench_kernels.py swt --reps 3 > $R/gpurun_out/${TAG}_pmc_$n.log 2>&1
done
find $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_pmc_* -name "*.csv" | head -20
