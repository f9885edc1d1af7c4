#!/bin/bash
# A/B of the bench on ONE box: current default vs option overrides.
set -u
export TMPDIR=/tmp
for i in 1 2; do
echo "== bench default"; timeout 600 python bench.py --steps 1200 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['host_breakdown_us_per_step'], b['roofline']['avg_launch_us'])"
done
echo "== microbench"; timeout 400 python tools/gpu_microbench.py --batch 1 8 2>&1 | grep -E "^variant"
rocm-smi --showclocks 2>/dev/null | head -20
