#!/bin/bash
# Round-2 final session: conv32mt as the default for single-FoV steps: whole GPU suite,
# bench (default, and --conv-variant 8 on the same box), profiles of the default path
# (kernel stats, SQ, FETCH / WRITE: each in its own pass), sharded bench (conv32m).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12 | tee gpurun_out/r02_pytest_gpu.txt
echo "== bench default"; timeout 600 python bench.py --steps 1500 --warmup 100 2>&1 | tail -1 | tee gpurun_out/r02_bench.json | cut -c1-200
echo "== (bench v8 skipped: profiles/r02_bench_conv32m_same_box.json)"
bash tools/gpu_r2_q.sh 2>&1 | grep -v "^W2026" | cut -c1-200
