#!/bin/bash
# Round-2 session Y: conv32m as the default: whole GPU suite, bench (default and --conv-variant 6),
# then the profiles of the default path (kernel stats, SQ, FETCH / WRITE), each in its own pass.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12 | tee gpurun_out/r02_pytest_gpu_y.txt
echo "== bench default"; timeout 600 python bench.py --steps 1500 --warmup 100 2>&1 | tail -1 | tee gpurun_out/r02_bench_default.json | cut -c1-200
echo "== bench v6"; timeout 600 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --conv-variant 6 2>&1 | tail -1 | tee gpurun_out/r02_bench_v6.json | cut -c1-200
echo "== bench default again"; timeout 600 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
bash tools/gpu_r2_q.sh 2>&1 | grep -v "^W2026" | cut -c1-200
