#!/bin/bash
# Quick GPU session: smoke + fast parity subset + microbench + bench.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest subset"; timeout 600 python -m pytest tests -m gpu -x -q -k "predict or anisotropic or layerwise or cells56 or full_size or canvas_step" 2>&1 | tail -4
echo "== microbench"; timeout 400 python tools/gpu_microbench.py --batch 1 8 32 2>&1 | grep -E "^variant"
echo "== bench"; timeout 600 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_quick.json
