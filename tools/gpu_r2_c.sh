#!/bin/bash
# Round-2 session C: conv32k issue experiments + SQ / LDS counters (own --pmc pass).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== ablations"; timeout 600 python tools/gpu_ab_k.py --batch 1 --rounds 2 --ablate 2>&1 | tee gpurun_out/r02_ab_k3.txt | tail -24
c="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_SQk -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --prewarm-seconds 0 --no-cpu-baseline --conv-variant 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc_SQk.log 2>&1; cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/pmc_SQk.log | cut -c1-120
f=$(find gpurun_out/pmc_SQk -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/r02_pmc_sq_conv32k.txt
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0][:58]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0])))[:4]:
    print(k)
    for c, v in sorted(d.items()):
        print('   %-32s n=%5d mean = %.1f' % (c, len(v), sum(v) / len(v)))
PY
rm -rf gpurun_out/pmc_SQk
