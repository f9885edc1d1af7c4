#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -25 | tee gpurun_out/r02_pytest_gpu_n.txt
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 8 32 --variants 4 6 2>&1 | tee gpurun_out/r02_ab_d3.txt | grep -E "^batch|layer . wave 0"
echo "== bench default"; timeout 600 python bench.py --steps 1500 --warmup 100 2>&1 | tail -1 | tee gpurun_out/r02_bench_default.json | cut -c1-300
