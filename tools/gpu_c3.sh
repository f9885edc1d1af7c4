#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== new GPU tests"; timeout 600 python -m pytest tests -m gpu -x -q -k "runner or abi_rejects or empty" 2>&1 | tail -4
echo "== driver mode"
timeout 300 python tools/gpu_batch_bench.py --mode driver --canvases 8 --batch 8 --size 128 --steps 300 2>&1 | tail -1
timeout 300 python tools/gpu_batch_bench.py --mode driver --canvases 32 --batch 32 --size 112 --steps 200 2>&1 | tail -1
timeout 300 python tools/gpu_batch_bench.py --mode driver --canvases 32 --batch 16 --size 112 --steps 200 2>&1 | tail -1
