#!/bin/bash
# Round-2 session A: conv32k (variant 5) first light -- parity vs the oracle, same-process
# A/B against conv32w8, bench with the in-run parity leg.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest predict"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predict" 2>&1 | tail -6
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 8 32 2>&1 | tee gpurun_out/r02_ab_k.txt | tail -30
echo "== bench v4"; timeout 600 python bench.py --steps 1000 --warmup 100 --cpu-seconds 8 2>&1 | tail -1 | tee gpurun_out/r02_bench_v4.json
echo "== bench v5"; timeout 600 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --conv-variant 5 2>&1 | tail -1 | tee gpurun_out/r02_bench_v5.json
