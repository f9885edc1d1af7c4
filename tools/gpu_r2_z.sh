#!/bin/bash
# Round-2 session Z: conv32mt (conv_variant 9): parity, same-process A/B with conv32m at
# batch 1 / 8 / 32, in-kernel clocks, bench 9 vs 8, the cells250 trajectory.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest v9"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "variants_agree or other_fov" 2>&1 | tail -8 | cut -c1-300
echo "== A/B"
timeout 600 python tools/gpu_ab_k.py --variants 8 9 --batch 1 8 32 --rounds 7 --repeats 60 2>&1 | grep -v "^W2026" | tee gpurun_out/r02_ab_conv32m_conv32mt.txt | cut -c1-220
echo "== bench v9"; timeout 600 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --conv-variant 9 2>&1 | tail -1 | tee gpurun_out/r02_bench_v9.json | cut -c1-300
echo "== bench v8"; timeout 600 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --conv-variant 8 2>&1 | tail -1 | tee gpurun_out/r02_bench_v8.json | cut -c1-300
echo "== cells250 v9"
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "cells250_matches and 9" 2>&1 | tail -5 | cut -c1-300
