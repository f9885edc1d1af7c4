#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== batched canvases (clients = 2 x batch)"
timeout 300 python tools/gpu_batch_bench.py --canvases 8 --batch 4 --size 128 --steps 300 2>&1 | tail -1
timeout 300 python tools/gpu_batch_bench.py --canvases 16 --batch 8 --size 128 --steps 300 2>&1 | tail -1
timeout 400 python tools/gpu_batch_bench.py --canvases 32 --batch 16 --size 112 --steps 200 2>&1 | tail -1
timeout 400 python tools/gpu_batch_bench.py --canvases 64 --batch 32 --size 112 --steps 150 2>&1 | tail -1
