"""GPU parity of PolicyPeaks (include/ffn_seeds.h) through the C-ABI: every
intermediate volume bit-exact against the scipy routines the reference calls,
the seed list against the reference-minted fixture and against the oracle."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def seeder():
  from ffn_amd import seeding
  return seeding.default_seeder(0)


def _image(shape, seed):
  from ffn_amd import synthetic
  return synthetic.normalize(synthetic.cells_volume(shape, seed=seed))


def _check_stages(seeder, image, exclusion=None, force=None, voxel=(1, 1, 1),
                  exact_dt=True):
  from oracle import seeds_oracle
  stages = {}
  want = seeds_oracle.policy_peaks(image, exclusion, force, voxel,
                                   stages=stages)
  got = seeder.peaks(image, exclusion, force, voxel)
  edges = seeder.read_stage(0, image.shape)
  thresh = seeder.read_stage(1, image.shape)
  assert np.array_equal(edges, stages['edges'])
  assert np.array_equal(thresh, stages['thresh'])
  if want is None:
    assert got is None
    return got
  dt = seeder.read_stage(2, image.shape)
  if exact_dt:
    assert np.array_equal(dt, stages['dt'])
  else:
    assert np.allclose(dt, stages['dt'], rtol=2e-7, atol=0)
  assert np.array_equal(got, want)
  return got


def test_stages_and_seeds_match_scipy_bitwise(seeder):
  rng = np.random.RandomState(0)
  for shape, seed in (((64, 64, 64), 21), ((48, 72, 80), 22),
                      ((20, 31, 45), 23), ((9, 70, 12), 24)):
    image = _image(shape, seed)
    ex = np.zeros(shape, bool)
    ex[2:shape[0] // 2, 5:25, 3:11] = True
    got = _check_stages(seeder, image, ex)
    assert got is not None
  # arbitrary float images (not just the phantom's value range)
  image = rng.normal(0, 3, (33, 40, 37)).astype(np.float32)
  image = np.round(image * 7) / 7
  _check_stages(seeder, image.astype(np.float32))


def test_force_edge_anisotropy_and_degenerate_cases(seeder):
  image = _image((40, 48, 56), 31)
  force = np.zeros(image.shape, bool)
  force[:, 20:24, :] = True
  _check_stages(seeder, image, None, force)
  # anisotropic voxels: same distances up to f64 summation order
  _check_stages(seeder, image, None, None, voxel=(2.5, 1.0, 1.0),
                exact_dt=False)
  # every voxel an edge -> no seeds at all (seed.py:178-179)
  assert seeder.peaks(image, None, np.ones(image.shape, bool)) is None
  # constant image: no edge anywhere -> dt = -1 everywhere -> empty list
  flat = np.full((24, 24, 24), 1.5, np.float32)
  got = seeder.peaks(flat)
  assert got is not None and got.shape == (0, 3)


def test_policy_peaks_matches_reference_fixture(seeder):
  """Product PolicyPeaks on host arrays == the reference's PolicyPeaks (real
  scikit-image 0.18.3, tools/make_golden_peaks.py)."""
  from ffn_amd import synthetic
  from ffn_amd.inference import seed as seed_lib
  g = np.load(os.path.join(GOLDEN, 'ref_policy_peaks.npz'))

  class C:
    restrictor = None
    voxel_size_zyx = (1, 1, 1)

  for n in 'ab':
    c = C()
    c.image = synthetic.normalize(g[n + '_volume'])
    c.shape = c.image.shape
    c.margin = np.array([4, 4, 4])
    c.segmentation = np.zeros(c.shape, np.int32)
    c.segmentation[20:30, 20:30, 20:30] = 3
    got = np.array([p for p in seed_lib.PolicyPeaks(c)]).reshape(-1, 3)
    assert np.array_equal(got, g[n + '_seeds']), n


def test_device_canvas_peaks_in_place(seeder):
  """DeviceCanvas path: image and segmentation are read in HBM."""
  from ffn_amd import synthetic
  from ffn_amd.inference import executor
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  from ffn_amd.inference import seed as seed_lib
  from ffn_amd.training.models import convstack_3d
  from oracle import seeds_oracle
  import bench
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33],
                                           deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(GOLDEN, 'fib25_weights.npz'))
  request = bench.make_request()
  counters = inference_utils.Counters()
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), model,
                                  model.info, None, counters, 1)
  g = np.load(os.path.join(GOLDEN, 'ref_policy_peaks.npz'))
  image = synthetic.normalize(g['a_volume'])
  canvas = inference.DeviceCanvas(
      model.info, exe.get_client(counters, direct=True), image,
      request.inference_options, counters=counters,
      movement_policy_fn=movement.get_policy_fn(request, model.info))
  seg = np.zeros(image.shape, np.int32)
  seg[20:30, 20:30, 20:30] = 3
  seg[5, 5, 5] = -1  # "excluded" markers are not segments
  canvas.segmentation[...] = seg
  policy = seed_lib.PolicyPeaks(canvas)
  got = np.array([p for p in policy]).reshape(-1, 3)
  want = seeds_oracle.policy_peaks(image, seg > 0)
  m = np.array(canvas.margin)
  want = want[np.all((want - m >= 0) & (want + m < image.shape), axis=1)]
  assert np.array_equal(got, want) and len(got) > 3
  canvas.close()


def test_full_size_250(seeder):
  """BASELINE size: equality with the scipy pipeline (seconds of CPU) plus
  properties of a peak list."""
  from oracle import seeds_oracle
  image = _image((250, 250, 250), 1234)
  got = seeder.peaks(image)
  ms, vox = seeder.last_timing()
  assert len(got) > 500
  assert np.array_equal(got, np.unique(got, axis=0))  # sorted, no duplicates
  assert got.min() >= 3 and (got < np.array(image.shape) - 3).all()
  # no two peaks inside each other's 7^3 neighbourhood
  occ = np.zeros(image.shape, bool)
  occ[tuple(got.T)] = True
  from scipy import ndimage
  assert ndimage.maximum_filter(occ.astype(np.uint8), size=7)[
      tuple(got.T)].all()
  counts = ndimage.uniform_filter(occ.astype(np.float64), size=7,
                                  mode='constant') * 343
  assert np.round(counts[tuple(got.T)]).max() == 1
  want = seeds_oracle.policy_peaks(image)
  assert np.array_equal(got, want)
  print('\\n250^3 PolicyPeaks on the GPU: %.2f ms of kernels (%.0f Mvox/s), '
        '%d seeds' % (ms, vox / ms / 1e3, len(got)))


def test_seeder_abi_errors(seeder):
  import ctypes
  from ffn_amd import _lib
  lib = _lib.load()
  h = ctypes.c_void_p()
  assert lib.ffn_seeder_create(0, ctypes.byref(h)) == 0
  img = np.zeros((4, 4, 4), np.float32)
  shape = (ctypes.c_int64 * 3)(4, 4, 4)
  voxel = (ctypes.c_double * 3)(1, 1, 1)
  n = ctypes.c_size_t(0)
  ae = ctypes.c_int32(0)
  coords = np.zeros((8, 3), np.int32)
  # gaussian / noise not set yet
  assert lib.ffn_seeder_peaks(h, img.ctypes.data, None, None, shape, voxel, 8,
                              coords.ctypes.data, ctypes.byref(n),
                              ctypes.byref(ae)) < 0
  assert b'set_gaussian' in lib.ffn_last_error()
  w = np.ones(3) / 3
  assert lib.ffn_seeder_set_gaussian(h, w.ctypes.data, 1) == 0
  assert lib.ffn_seeder_peaks(h, img.ctypes.data, None, None, shape, voxel, 8,
                              coords.ctypes.data, ctypes.byref(n),
                              ctypes.byref(ae)) < 0
  assert b'noise' in lib.ffn_last_error()
  assert lib.ffn_seeder_set_gaussian(h, w.ctypes.data, 100000) < 0
  bad = (ctypes.c_int64 * 3)(4, 0, 4)
  assert lib.ffn_seeder_peaks(h, img.ctypes.data, None, None, bad, voxel, 8,
                              coords.ctypes.data, ctypes.byref(n),
                              ctypes.byref(ae)) < 0
  assert lib.ffn_seeder_read_stage(h, 0, img.ctypes.data) < 0
  lib.ffn_seeder_destroy(h)
