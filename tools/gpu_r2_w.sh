#!/bin/bash
# Round-2 session W: counters of the BATCHED configuration (conv32m, batch 8).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== stats"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profb -o b8 -- python $GRAFT_REPO_ROOT/tools/gpu_batch_profile.py --batch 8 --variant 8 > $GRAFT_REPO_ROOT/gpurun_out/batch8_stats.log 2>&1; cd $GRAFT_REPO_ROOT
grep "^variant" gpurun_out/batch8_stats.log
for f in $(find gpurun_out/profb -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/r02_batch8_conv32m_kernel_stats.csv; head -6 $f | cut -c1-150; done
rm -rf gpurun_out/profb
echo "== SQ"
c="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcb -o pmc -- python $GRAFT_REPO_ROOT/tools/gpu_batch_profile.py --batch 8 --variant 8 --repeats 10 > $GRAFT_REPO_ROOT/gpurun_out/batch8_pmc.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/pmcb -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/r02_pmc_sq_conv32m_batch8.txt
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
rm -rf gpurun_out/pmcb
