#!/bin/bash
# Round-4 profile set of the default bench.py command (gpurun -- bash tools/gpu_profile_r4.sh):
#   1. rocprofv3 --kernel-trace --stats      -> gpurun_out/r4prof/kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes, kernel-trace only)
#                                            -> gpurun_out/r4prof/conv32ps_pmc_traffic.json
#   3. rocprofv3 --pmc SQ counters of the stack kernel
# The summaries are copied into profiles/ by hand (r04_*).
set -u
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4prof
mkdir -p $O
CMD="python $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-batched-leg --no-full-volume"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
cd $R
tail -1 $O/stats.log | cut -c1-400 > $O/bench_under_kernel_trace.json
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
head -12 $O/kernel_stats.csv
SHORT="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batched-leg --no-full-volume"
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- $SHORT > $O/pmc_$c.log 2>&1; cd $R
done
python tools/pmc_traffic_json.py "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
  "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $O/conv32ps_pmc_traffic.json \
  --command "bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batched-leg --no-full-volume"
cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $O/pmc_SQ -o pmc -- $SHORT > $O/pmc_SQ.log 2>&1; cd $R
python - "$(find $O/pmc_SQ -name '*counter_collection.csv' | head -1)" <<'PY' > $O/pmc_sq_conv32ps.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
  agg[r['Kernel_Name'].split('(')[0][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
  if 'conv' in k or 'faces' in k:
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, 'dispatches', len(next(iter(d.values()))))
PY
cat $O/pmc_sq_conv32ps.txt
