#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "predict_matches or variants_agree or anisotropic or predict_batch or cells56 or range" 2>&1 | grep -E "variant 4|passed|failed|Error|assert" | head -12
timeout 400 python tools/gpu_microbench.py --batch 1 8 32 2>&1 | grep -E "^variant 4\+w8 batch|clock wave 0|issue exp" | head -12
