#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest predict"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_agree or predict_matches or c5_model or range or layerwise" 2>&1 | tail -4
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 8 32 --variants 4 6 7 2>&1 | tee gpurun_out/r02_ab_g.txt | grep -E "^batch|layer . wave 0"
