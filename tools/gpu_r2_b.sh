#!/bin/bash
# Round-2 session B: conv32k with pinned prefetch + short address prologue; u8 canvas, resume, CLI.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 8 2>&1 | tee gpurun_out/r02_ab_k2.txt | tail -30
echo "== pytest round2"; timeout 1200 python -m pytest tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -12
echo "== pytest parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predict_matches or runner_end" 2>&1 | tail -5
