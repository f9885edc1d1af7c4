#!/bin/bash
# torchrun path of bench.py (one rank: RCCL init, barriers, merge leg) + new tests.
set -u
export TMPDIR=/tmp
echo "== anisotropic canvas test"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "anisotropic" 2>&1 | tail -4
echo "== torchrun bench (1 rank)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 600 --warmup 50 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
