#!/bin/bash
# LDS-array load of the conv kernels (own rocprofv3 --pmc pass): is the tap loop bound by the LDS?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
c="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_LDS -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --prewarm-seconds 0 --no-cpu-baseline --no-batched-leg > $GRAFT_REPO_ROOT/gpurun_out/pmc_LDS.log 2>&1; cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/pmc_LDS.log | cut -c1-160
f=$(find gpurun_out/pmc_LDS -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0][:58]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0])))[:4]:
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%5d mean = %.1f' % (c, len(v), sum(v) / len(v)))
PY
rm -rf gpurun_out/pmc_LDS
