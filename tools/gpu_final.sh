#!/bin/bash
# Closing GPU session of a round: parity suite, bench (with cpu baseline), rocprof kernel stats.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== bench"; timeout 300 python bench.py --steps 1500 --warmup 100 2>&1 | tail -1 | tee gpurun_out/bench.json
echo "== rocprof"; cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_final.log 2>&1; cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/rocprof_final.log
for f in $(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1); do head -8 $f; done
