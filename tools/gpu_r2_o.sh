#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest predict"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predict or batched or ragged" 2>&1 | tail -5
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 2 4 8 32 --variants 4 6 7 2>&1 | tee gpurun_out/r02_ab_e.txt | grep -E "^batch|^variant . :|layer . wave 0|max \|logit"
