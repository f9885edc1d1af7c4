#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-2400
