#!/bin/bash
# HBM traffic of the conv kernel: separate rocprofv3 --pmc passes (no other
# trace domains than kernel-trace), short bench run.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-batched-leg "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1; cd $GRAFT_REPO_ROOT
  tail -1 gpurun_out/pmc_$c.log | cut -c1-200
  f=$(find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get('Counter_Name') == c:
        agg[r['Kernel_Name'].split('(')[0][:60]].append(float(r['Counter_Value']))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print('%-62s n=%5d mean %s = %.1f' % (k, len(v), c, sum(v) / len(v)))
PY
done
