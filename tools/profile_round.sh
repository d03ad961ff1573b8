#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>
#   bench line; rocprofv3 --kernel-trace --stats of the same command; FETCH_SIZE / WRITE_SIZE (each PMC set in its own
#   pass, kernel-trace only) for the SWT kernel, the ranking kernels and the head -> gpurun_out/<tag>_*; the summaries are then
#   copied into profiles/ by hand.
set -e
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 30 --warmup 5 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
tail -c 300 $R/gpurun_out/${TAG}_bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-grid > $R/gpurun_out/${TAG}_stats.log 2>&1
for what in swt topk rankmap head; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"; do
    n=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_${what}_$n -o pmc -- python3 $R/tools/bench_kernels.py $what --reps 3 > $R/gpurun_out/${TAG}_pmc_${what}_$n.log 2>&1
  done
done
find $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_pmc_* -name "*.csv" | head -40
