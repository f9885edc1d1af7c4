#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest seeds"; timeout 900 python -m pytest tests/test_gpu_seeds.py -x -q -s 2>&1 | tail -30
