#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6
timeout 300 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
