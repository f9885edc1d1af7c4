#!/bin/bash
# Round-6 profile set of the default bench.py command (paced resident stack)
#   COMMIT=$(git rev-parse --short HEAD) gpurun -- "COMMIT=$COMMIT bash tools/gpu_profile_r6.sh"
#   1. rocprofv3 --kernel-trace --stats                    -> gpurun_out/r6prof/kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE, SQ counters (separate passes, kernel-trace only)
#                                                          -> gpurun_out/r6prof/conv32ps_pmc.json
# (what bench.py's roofline.traffic / mfma_busy read once copied to profiles/r06_conv32ps_pmc.json;
# the JSON carries the sha of ffn_amd/csrc it was taken on: roofline.traffic_stale)
set -u
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6prof
mkdir -p $O
SHA=$(cd $R && python -c "from ffn_amd import _lib; print(_lib.csrc_sha())")
CMD="python $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-batched-leg --no-c5-leg --no-full-volume"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
cd $R
grep "^{" $O/stats.log | tail -1 > $O/bench_under_kernel_trace.json
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
head -8 $O/kernel_stats.csv
t=$(find $O/stats -name "*kernel_trace.csv" | head -1)
python tools/kernel_gaps.py "$t" | tee $O/kernel_gaps.txt
SHORT="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batched-leg --no-c5-leg --no-full-volume"
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- $SHORT > $O/pmc_$c.log 2>&1; cd $R
done
cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $O/pmc_SQ -o pmc -- $SHORT > $O/pmc_SQ.log 2>&1; cd $R
python tools/pmc_traffic_json.py "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
  "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $O/conv32ps_pmc.json \
  --sq-csv "$(find $O/pmc_SQ -name '*counter_collection.csv' | head -1)" --commit "${COMMIT:-}" --csrc-sha "$SHA" \
  --command "bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batched-leg --no-c5-leg --no-full-volume"
