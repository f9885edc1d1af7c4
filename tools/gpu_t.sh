#!/bin/bash
set -u
export TMPDIR=/tmp
free -g | head -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "large_canvas" 2>&1 | tail -12
