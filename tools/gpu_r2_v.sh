#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "variants_agree or predict_matches or (cells250_matches and 8)" 2>&1 | tail -3
echo "== A/B"; timeout 600 python tools/gpu_ab_k.py --batch 1 8 32 --variants 6 7 8 2>&1 | tee gpurun_out/r02_ab_i.txt | grep -E "^batch|variant 8 layer . wave 0"
echo "== sharded v6/7"; timeout 900 python bench.py --mode sharded --sharded-volume 256 --sharded-sub 144 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['conv_variant'], d['merge_ms'], d['reconcile_ms'], d['assembly']['ids_after_reconcile'])"
echo "== sharded v8"; timeout 900 python bench.py --mode sharded --sharded-volume 256 --sharded-sub 144 --no-cpu-baseline --conv-variant 8 2>&1 | tail -1 | tee gpurun_out/r02_bench_sharded_v8.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['conv_variant'], d['merge_ms'], d['reconcile_ms'], d['assembly']['ids_after_reconcile'])"
echo "== sharded v8 batch 32"; timeout 900 python bench.py --mode sharded --sharded-volume 256 --sharded-sub 112 --sharded-batch 32 --no-cpu-baseline --conv-variant 8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['sub_boxes'], d['merge_ms'], d['reconcile_ms'], d['assembly']['ids_after_reconcile'])"
