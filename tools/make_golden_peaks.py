#!/opt/conda/bin/python3.9
"""Mints tests/golden/ref_policy_peaks.npz with the reference's own PolicyPeaks.

Must run under /opt/conda/bin/python3.9 -- the only interpreter in this image
with scikit-image (0.18.3), whose `peak_local_max` the reference calls
(ffn/inference/seed.py:133-139).  `edt` (also un-vendored, absent everywhere)
is shimmed with scipy's exact EDT (tools/ref_shims/edt.py); everything else is
the reference's unmodified code (seed.py:63-95 margin filter, :142-199
PolicyPeaks.init_coords).
"""
import os
import sys

import numpy as np
import skimage  # the real one: imported BEFORE the shim path is added
import skimage.feature
import skimage.morphology

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'tools', 'ref_shims'))
sys.path.insert(0, ROOT)

from ffn.inference import seed as ref_seed  # noqa: E402
from ffn_amd import synthetic  # noqa: E402


class FakeCanvas:
  restrictor = None
  voxel_size_zyx = (1, 1, 1)


def main():
  out = {}
  for name, shape, seed, dil in [('a', (64, 64, 64), 21, 1),
                                 ('b', (48, 72, 80), 22, 2)]:
    vol = synthetic.cells_volume(shape, seed=seed, membrane_dilate=dil)
    canvas = FakeCanvas()
    canvas.image = synthetic.normalize(vol)
    canvas.shape = canvas.image.shape
    canvas.margin = np.array([4, 4, 4])
    canvas.segmentation = np.zeros(shape, np.int32)
    canvas.segmentation[20:30, 20:30, 20:30] = 3  # exercised exclusion mask
    pol = ref_seed.PolicyPeaks(canvas)
    coords = np.array([p for p in pol], dtype=np.int64).reshape(-1, 3)
    raw = np.array(pol.coords)
    out[name + '_volume'] = vol
    out[name + '_seeds'] = coords
    print(name, shape, 'peaks after margin filter:', len(coords),
          'first:', coords[:3].tolist())
    assert len(raw) == len(coords)
  np.savez_compressed(os.path.join(ROOT, 'tests', 'golden',
                                   'ref_policy_peaks.npz'), **out)


if __name__ == '__main__':
  main()
