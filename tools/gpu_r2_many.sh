#!/bin/bash
# Round-2: ffn_canvas_segment_many: its GPU test, the batched / sharded / resegmentation tests
# (their drivers now run whole segments in the library), sharded bench native vs per-step driver.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"
timeout 1200 python -m pytest tests -m gpu -q -x -k "segment_many or sharded or reseg or batched or run_many or cli" 2>&1 | tail -6 | cut -c1-300
echo "== sharded, native driver"; timeout 900 python bench.py --mode sharded --sharded-volume 256 --sharded-sub 144 2>&1 | tail -1 | tee gpurun_out/r02_bench_sharded_native.json | cut -c1-700
echo "== sharded, per-step Python driver"; FFN_AMD_NATIVE_MANY=0 timeout 900 python bench.py --mode sharded --sharded-volume 256 --sharded-sub 144 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02_bench_sharded_python_driver.json | cut -c1-300
