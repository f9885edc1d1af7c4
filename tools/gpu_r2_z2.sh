#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "variants_agree or other_fov or batch_and_ragged or matches_oracle" 2>&1 | tail -4 | cut -c1-300
echo "== A/B"
timeout 600 python tools/gpu_ab_k.py --variants 8 9 --batch 1 8 32 --rounds 7 --repeats 40 2>&1 | grep -v "^W2026" | tee gpurun_out/r02_ab_xahead.txt | grep "^batch\|max\|layer 3" | cut -c1-220
echo "== bench"; timeout 600 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02_bench_xahead.json | cut -c1-200
echo "== timeline"; timeout 300 python tools/gpu_wg_timeline.py --variants 9 2>&1 | grep -v "^W2026" | tee gpurun_out/r02_wg_timeline_xahead.txt
