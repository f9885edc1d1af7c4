#!/bin/bash
# One GPU session: smoke, parity tests, micro-benchmark, bench, rocprof summary.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== microbench"; timeout 400 python tools/gpu_microbench.py --batch 1 8 32 2>&1 | grep -E "^variant|clock wave|ablate"
echo "== bench sync0"; timeout 600 python bench.py --steps 1000 --warmup 100 --sync-mode 0 --no-cpu-baseline 2>&1 | tail -1
echo "== bench"; timeout 600 python bench.py --steps 1500 --warmup 100 2>&1 | tail -1 | tee gpurun_out/bench.json
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/rocprof.log
echo "== pmc"; bash tools/gpu_pmc.sh
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -12 $f; done
