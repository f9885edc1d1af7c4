#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== pytest labels+seeds"; timeout 900 python -m pytest tests/test_gpu_labels.py tests/test_gpu_seeds.py -x -q 2>&1 | tail -4
timeout 600 python tools/gpu_labels_bench.py --size 250 400 2>&1 | tail -20
