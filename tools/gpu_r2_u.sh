#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest predict"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or predict_matches or ragged or batch" 2>&1 | tail -4
echo "== cells250"; timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q -k "cells250_matches and 8 or logit_tolerance" -s 2>&1 | grep -E "variant|passed|failed|max \|logit|Error" | cut -c1-300 | tail -8
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 2 8 32 --variants 4 6 7 8 2>&1 | tee gpurun_out/r02_ab_h.txt | grep -E "^batch|layer . wave 0|max \|logit"
