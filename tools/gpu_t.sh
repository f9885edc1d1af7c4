#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "predict or threaded" 2>&1 | tail -4
timeout 400 python tools/gpu_microbench.py --batch 1 8 2>&1 | grep -E "ffn_predict"
