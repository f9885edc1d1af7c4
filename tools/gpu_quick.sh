#!/bin/bash
# Quick GPU session: microbench + bench + rocprof (no pytest).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== microbench"; timeout 400 python tools/gpu_microbench.py --batch 1 8 32 2>&1 | tail -24
echo "== bench"; timeout 600 python bench.py --steps 1500 --warmup 100 2>&1 | tail -2 | tee gpurun_out/bench.json
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/rocprof.log
find gpurun_out/prof -type f | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -14 $f; done
