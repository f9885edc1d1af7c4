#!/bin/bash
# GPU session for the label kernels: parity tests + the repaired full-size test.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest labels"; timeout 900 python -m pytest tests/test_gpu_labels.py -x -q -s 2>&1 | tail -25
echo "== pytest parity (full size + runner)"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size or runner" 2>&1 | tail -5
