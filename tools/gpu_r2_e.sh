#!/bin/bash
# Round-2 session E: conv32d (variant 6) first light: parity, cells250 fixture for every
# variant, same-process A/B against conv32w8 / conv32k, bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest predict"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predict or anisotropic_fov or layerwise" 2>&1 | tail -8
echo "== pytest cells250"; timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q -k "cells250" -s 2>&1 | grep -E "variant|passed|failed|Error|error" | tail -12
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 8 32 --variants 4 5 6 2>&1 | tee gpurun_out/r02_ab_d.txt | tail -30
echo "== bench v6"; timeout 600 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --conv-variant 6 2>&1 | tail -1 | tee gpurun_out/r02_bench_v6.json
