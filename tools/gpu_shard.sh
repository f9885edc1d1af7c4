#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest sharded"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sharded" 2>&1 | tail -30
