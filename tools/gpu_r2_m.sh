#!/bin/bash
# Round-2 session M: whole GPU suite with conv32d as the default, the new fixtures,
# c5 config and the sharded mode.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q --durations=6 -x 2>&1 | tail -25 | tee gpurun_out/r02_pytest_gpu_m.txt
echo "== bench default"; timeout 600 python bench.py --steps 1500 --warmup 100 2>&1 | tail -1 | tee gpurun_out/r02_bench_default.json | cut -c1-300
echo "== bench c5"; timeout 900 python bench.py --config c5 --steps 600 --warmup 50 --cpu-steps 12 2>&1 | tail -1 | tee gpurun_out/r02_bench_c5.json | cut -c1-600
echo "== bench sharded"; timeout 1200 python bench.py --mode sharded --sharded-volume 256 --sharded-sub 144 2>&1 | tail -1 | tee gpurun_out/r02_bench_sharded.json | cut -c1-1200
