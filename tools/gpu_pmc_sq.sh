#!/bin/bash
# MFMA utilisation of the conv kernel from SQ counters (own rocprofv3 --pmc pass).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
c="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS"
cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --prewarm-seconds 0 --no-cpu-baseline --no-batched-leg > $GRAFT_REPO_ROOT/gpurun_out/pmc_SQ.log 2>&1; cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/pmc_SQ.log | cut -c1-200
f=$(find gpurun_out/pmc_SQ -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0][:58]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0])))[:6]:
    print(k)
    for c, v in sorted(d.items()):
        print('   %-32s n=%5d mean = %.1f' % (c, len(v), sum(v) / len(v)))
PY
