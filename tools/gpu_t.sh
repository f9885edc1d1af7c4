#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "predict_matches" 2>&1 | grep -E "variant [34].*waves8 1|passed|failed|Error|assert" | head -8
timeout 400 python tools/gpu_microbench.py --batch 1 8 32 2>&1 | grep -E "^variant [34]\+w8 batch|clock wave 0|issue exp" | head -12
