#!/bin/bash
# Round-2 session J: conv32d v2 (one accumulator, DMA first, hidden weight loads).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest predict"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predict or anisotropic_fov or layerwise or range" 2>&1 | tail -5
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 8 32 --variants 4 6 2>&1 | tee gpurun_out/r02_ab_d2.txt | tail -30
echo "== bench v6"; timeout 600 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --conv-variant 6 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/r02_bench_v6.json
echo "== rocprof v6"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline --conv-variant 6 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r02_bench_v6_kernel_stats.csv; head -9 $f | cut -c1-200; done
rm -rf gpurun_out/prof
