#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/gpu_labels_bench.py --size 250 400 2>&1 | tail -20
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_labels -o lab -- python $GRAFT_REPO_ROOT/tools/gpu_labels_bench.py --size 250 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/prof_labels -name "*kernel_stats.csv" | head -1); do head -40 $f | cut -c1-200; done
