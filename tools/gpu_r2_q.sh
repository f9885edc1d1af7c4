#!/bin/bash
# Round-2 session Q: profiles of the default path (conv32d): rocprofv3 kernel stats,
# SQ counters and HBM traffic counters, each in its own pass.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocprof stats"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/rocprof.log | cut -c1-200
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r02_default_kernel_stats.csv; head -9 $f | cut -c1-160; done
rm -rf gpurun_out/prof
echo "== SQ"; bash tools/gpu_pmc_sq.sh 2>&1 | tee gpurun_out/r02_pmc_sq_default.txt | head -40
rm -rf gpurun_out/pmc_SQ
echo "== traffic"; bash tools/gpu_pmc.sh 2>&1 | tee gpurun_out/r02_pmc_fetch_write.txt
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
echo "== bench batch (sharded 256)"; timeout 1200 python bench.py --mode sharded --sharded-volume 256 --sharded-sub 144 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r02_bench_sharded.json | cut -c1-400
