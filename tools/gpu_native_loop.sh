#!/bin/bash
# GPU session: native segment loop parity + bench A/B (native vs python host loop).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"; timeout 200 python -m pytest tests -m gpu -x -q -k "native or resegmentation or runner or sharded or reference_run or exhausted" 2>&1 | tail -12
echo "== bench native"; timeout 100 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-420
echo "== bench python"; timeout 100 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline --host-loop python 2>&1 | tail -1 | cut -c1-420
