#!/bin/bash
# GPU session: resegmentation + EDT parity first, EDT timing, then the whole GPU suite.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== reseg tests"; timeout 300 python -m pytest tests/test_resegmentation.py -m gpu -x -q 2>&1 | tail -15
echo "== edt timing"; timeout 120 python - <<'PY' 2>&1 | tail -3
import numpy as np, time
from scipy import ndimage
from ffn_amd import seeding
s = seeding.default_seeder(0)
rng = np.random.default_rng(0)
m = ndimage.binary_dilation(rng.random((250, 250, 250)) < 1e-4, iterations=12)
s.edt(m)
t = time.time(); d = s.edt(m); wall = time.time() - t
ms, _ = s.last_timing()
t = time.time(); ref = ndimage.distance_transform_edt(m); cpu = time.time() - t
print('edt 250^3: kernels %.2f ms, call %.1f ms, scipy %.0f ms, equal %s' % (ms, wall * 1e3, cpu * 1e3, np.array_equal(d, ref)))
PY
echo "== full gpu suite"; timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
