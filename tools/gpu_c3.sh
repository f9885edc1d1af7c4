#!/bin/bash
# Config C3 (many canvases on one GPU): sharded end-to-end test + driver rates.
set -u
export TMPDIR=/tmp
echo "== sharded / driver GPU tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sharded or batched or runner" 2>&1 | tail -6
echo "== driver mode (overlap off / on)"
timeout 300 python tools/gpu_batch_bench.py --mode driver --canvases 8 --batch 8 --size 128 --steps 300 2>&1 | grep "^driver"
timeout 300 python tools/gpu_batch_bench.py --mode driver --canvases 16 --batch 8 --size 128 --steps 300 2>&1 | grep "^driver"
timeout 300 python tools/gpu_batch_bench.py --mode driver --canvases 32 --batch 16 --size 112 --steps 200 2>&1 | grep "^driver"
timeout 300 python tools/gpu_batch_bench.py --mode driver --canvases 64 --batch 32 --size 112 --steps 150 2>&1 | grep "^driver"
