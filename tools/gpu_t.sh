#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/gpu_batch_bench.py --mode driver --canvases 16 --batch 8 --size 128 --steps 300 2>&1 | grep "^driver"
timeout 300 python tools/gpu_batch_bench.py --mode driver --canvases 64 --batch 32 --size 112 --steps 150 2>&1 | grep "^driver"
bash tools/gpu_pmc_sq.sh 2>&1 | tail -24
