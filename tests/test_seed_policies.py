"""Every seed policy a request can name (reference ffn/inference/seed.py:202-549)
against lists minted by the reference's own classes with scikit-image 0.18.3
(tools/make_golden_seed_policies.py -> tests/golden/ref_seed_policies.npz)."""
import os

import numpy as np
import pytest

from ffn_amd import synthetic
from ffn_amd.inference import seed as seed_lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                      'ref_seed_policies.npz')

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
    ('peaks2d_masked', 'PolicyPeaks2d', {}),
    ('max_peaks_masked', 'PolicyMaxPeaks', {}),
]


class _Restrictor:
  mask = None
  seed_mask = None


class _Canvas:
  restrictor = None
  voxel_size_zyx = (1, 1, 1)


def _canvas(g, masked):
  c = _Canvas()
  c.image = synthetic.normalize(g['volume'])
  c.shape = c.image.shape
  c.margin = np.array([3, 4, 5])
  c.segmentation = np.array(g['segmentation'])
  if masked:
    c.restrictor = _Restrictor()
    c.restrictor.mask = np.array(g['mask'])
  return c


@pytest.mark.parametrize('name,cls,kwargs', CASES)
def test_policy_matches_the_reference(name, cls, kwargs):
  g = np.load(GOLDEN)
  canvas = _canvas(g, name.endswith('_masked'))
  if cls.startswith('PolicyImagePeaks'):
    # pinned on a tie-free image (equal intensities: skimage's order is not
    # defined, see seed_lib.peak_local_max)
    canvas.image = (canvas.image.astype(np.float64) +
                    np.random.RandomState(5).rand(*canvas.shape) * 1e-3)
  pol = getattr(seed_lib, cls)(canvas, **kwargs)
  got = np.array(list(pol), dtype=np.int64).reshape(-1, 3)
  want = g[name]
  assert got.shape == want.shape, (got.shape, want.shape)
  assert np.array_equal(got, want)


def test_every_reference_policy_name_resolves():
  """`Runner` looks the policy class up by the name in the request
  (runner.py:423-431): every class of the reference's seed.py exists here."""
  for name in ('PolicyPeaks', 'PolicyPeaks2d', 'PolicyFillEmptySpace',
               'PolicyMax', 'PolicyMaxPeaks', 'PolicyImagePeaks3D2D',
               'PolicyImagePeaks2DDisk', 'PolicyGrid3d', 'PolicyGrid2d',
               'PolicyInvertOrigins', 'PolicyDenseSeeds', 'ReverseCoords',
               'SequentialPolicies'):
    assert issubclass(getattr(seed_lib, name), seed_lib.BaseSeedPolicy), name


def test_sequential_policies_state_roundtrip():
  g = np.load(GOLDEN)
  canvas = _canvas(g, False)
  kw = {'policies': [('PolicyGrid3d', {'step': 12, 'offsets': (0,)}),
                     ('PolicyMaxPeaks', {'min_distance': 4})]}
  a = seed_lib.SequentialPolicies(canvas, **kw)
  first = [next(a) for _ in range(5)]
  state = a.get_state()
  b = seed_lib.SequentialPolicies(canvas, **kw)
  b.set_state(state)
  assert list(b) == list(a)
  assert first == [tuple(v) for v in g['sequential'][:5]]
