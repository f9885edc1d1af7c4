#!/opt/conda/bin/python3.9
"""Mints tests/golden/ref_seed_policies.npz with the reference's own seed
policies (ffn/inference/seed.py:202-408, 433-450, 472-549).

Runs under /opt/conda/bin/python3.9 (scikit-image 0.18.3: `peak_local_max`,
`morphology.disk`, `binary_erosion` are the real ones); `edt` is shimmed with
scipy's exact EDT (tools/ref_shims/edt.py).  Everything else is the reference's
unmodified code, margin filter of BaseSeedPolicy.__next__ included.
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


class FakeRestrictor:
  mask = None
  seed_mask = None


class FakeCanvas:
  restrictor = None
  voxel_size_zyx = (1, 1, 1)


def make_canvas(shape, seed, with_mask=False):
  vol = synthetic.cells_volume(shape, seed=seed, membrane_dilate=1)
  canvas = FakeCanvas()
  canvas.image = synthetic.normalize(vol)
  canvas.shape = canvas.image.shape
  canvas.margin = np.array([3, 4, 5])
  canvas.segmentation = np.zeros(shape, np.int32)
  canvas.segmentation[8:20, 10:30, 12:40] = 3
  canvas.segmentation[25:33, 5:15, 30:50] = 7
  if with_mask:
    canvas.restrictor = FakeRestrictor()
    canvas.restrictor.mask = np.zeros(shape, bool)
    canvas.restrictor.mask[:, 36:, :8] = True
  return vol, canvas


CASES = [
    ('peaks2d', 'PolicyPeaks2d', {}),
    ('peaks2d_desc', 'PolicyPeaks2d',
     {'min_distance': 3, 'threshold_abs': 0, 'sort_cmp': 'descending'}),
    ('fill_empty', 'PolicyFillEmptySpace', {}),
    ('max_peaks', 'PolicyMaxPeaks', {}),
    ('max_peaks_rel', 'PolicyMaxPeaks',
     {'min_distance': 2, 'threshold_abs': 0.5, 'threshold_rel': 0.3}),
    ('image_peaks_3d2d', 'PolicyImagePeaks3D2D', {}),
    ('image_peaks_2d_disk', 'PolicyImagePeaks2DDisk', {}),
    ('image_peaks_2d_disk_r2', 'PolicyImagePeaks2DDisk',
     {'min_distance_2d': 2, 'threshold_rel': 0.3, 'disk_radius': 2}),
    ('grid2d', 'PolicyGrid2d', {}),
    ('dense', 'PolicyDenseSeeds', {'threshold': 1.0}),
    ('dense_eroded_inverted', 'PolicyDenseSeeds',
     {'threshold': 0.2, 'num_erosions': 2, 'invert': True}),
    ('reverse_grid3d', 'ReverseCoords',
     {'policy_to_reverse': 'PolicyGrid3d', 'step': 8, 'offsets': (0, 4)}),
    ('sequential', 'SequentialPolicies',
     {'policies': [('PolicyGrid3d', {'step': 12, 'offsets': (0,)}),
                   ('PolicyMaxPeaks', {'min_distance': 4})]}),
]


def main():
  out = {}
  shape = (40, 48, 56)
  for with_mask in (False, True):
    vol, canvas = make_canvas(shape, 31, with_mask)
    tag = '_masked' if with_mask else ''
    out['volume'] = vol
    out['segmentation'] = canvas.segmentation
    if with_mask:
      out['mask'] = canvas.restrictor.mask
    for name, cls, kwargs in CASES:
      if with_mask and name not in ('peaks2d', 'max_peaks'):
        continue
      image = canvas.image
      if cls.startswith('PolicyImagePeaks'):
        # raw-image peaks: skimage orders equal intensities by an unstable sort
        # (undefined); these two are pinned on a tie-free float64 image
        canvas.image = (image.astype(np.float64) +
                        np.random.RandomState(5).rand(*shape) * 1e-3)
      pol = getattr(ref_seed, cls)(canvas, **kwargs)
      coords = np.array([p for p in pol], dtype=np.int64).reshape(-1, 3)
      out[name + tag] = coords
      canvas.image = image
      print(name + tag, len(coords), coords[:2].tolist())
  np.savez_compressed(os.path.join(ROOT, 'tests', 'golden',
                                   'ref_seed_policies.npz'), **out)


if __name__ == '__main__':
  main()
