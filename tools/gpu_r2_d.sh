#!/bin/bash
# Round-2 session D: whole GPU suite, conv32w8 vs conv32k A/B, bench lines, rocprof kernel stats.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 | tee gpurun_out/r02_pytest_gpu.txt
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 8 32 2>&1 | tee gpurun_out/r02_ab_k.txt | tail -30
echo "== bench v4"; timeout 600 python bench.py --steps 1500 --warmup 100 2>&1 | tail -1 | tee gpurun_out/r02_bench_v4.json
echo "== bench v5"; timeout 600 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --conv-variant 5 2>&1 | tail -1 | tee gpurun_out/r02_bench_v5.json
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/rocprof.log | cut -c1-300
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r02_bench_kernel_stats.csv; head -12 $f; done
rm -rf gpurun_out/prof
