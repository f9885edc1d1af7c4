#!/bin/bash
# rocprofv3 per-kernel statistics of a short bench run.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/rocprof.log
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -12 $f; done
