#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q -k "cells250" -s > gpurun_out/r02_cells250_full.txt 2>&1
grep -E "^(E  |tests/|FAILED|PASSED|variant)" gpurun_out/r02_cells250_full.txt | cut -c1-400 | head -60
tail -8 gpurun_out/r02_cells250_full.txt
