"""Resegmentation against the reference's own
`resegmentation.process_point` output (tests/golden/ref_reseg.npz, minted by
tools/make_golden_reseg.py with the reference's code)."""
import os

import numpy as np
import pytest

from ffn_amd.inference import align
from ffn_amd.inference import inference
from ffn_amd.inference import inference_utils
from ffn_amd.inference import movement
from ffn_amd.inference import request as request_lib
from ffn_amd.inference import resegmentation
from ffn_amd.training.model import ModelInfo

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def build_request(g, out_dir):
  """The request tools/make_golden_reseg.py ran the reference with."""
  request = request_lib.ResegmentationRequest()
  o = request.inference.inference_options
  o.init_activation = 0.95
  o.pad_value = 0.05
  o.move_threshold = 0.9
  o.segment_threshold = 0.6
  o.min_segment_size = 1000
  o.min_boundary_dist.x = o.min_boundary_dist.y = o.min_boundary_dist.z = 1
  request.radius.x = request.radius.y = request.radius.z = 24
  request.output_directory = str(out_dir)
  request.max_retry_iters = 2
  request.exclusion_radius.x = request.exclusion_radius.y = 4
  request.exclusion_radius.z = 4
  request.segment_recovery_fraction = 0.5
  request.analysis_radius.x = request.analysis_radius.y = 8
  request.analysis_radius.z = 8
  gid_a, gid_b = (int(v) for v in g['ids'])
  point = [int(v) for v in g['point']]
  p = request.points.add()
  p.id_a, p.id_b = gid_a, gid_b
  p.point.z, p.point.y, p.point.x = point
  p2 = request.points.add()  # endpoint request: id_b omitted
  p2.id_a = gid_b
  p2.point.z, p2.point.y, p2.point.x = point
  return request


class StandInRunner:
  """What process_point needs of a Runner, over in-memory volumes; mirrors the
  stand-in the golden file was minted with."""

  def __init__(self, g, client_fn, request):
    self.volume = g['volume']
    self.init_seg_volume = g['init_seg'][np.newaxis]
    self.counters = inference_utils.Counters()
    self.client_fn = client_fn
    self.request = request
    self.info = ModelInfo(deltas=(8, 8, 8), pred_mask_size=(33, 33, 33),
                          input_seed_size=(33, 33, 33),
                          input_image_size=(33, 33, 33))

  def make_canvas(self, corner, subvol_size, **kwargs):
    corner = np.array(corner)
    end = corner + np.array(subvol_size)
    sel = tuple(slice(int(c), int(e)) for c, e in zip(corner, end))
    image = (self.volume[sel].astype(np.float32) - 128.0) / 33.0
    counters = self.counters.get_sub_counters()
    canvas = inference.make_canvas(
        self.info, self.client_fn(counters), image,
        self.request.inference_options, counters=counters,
        movement_policy_fn=movement.get_policy_fn(self.request, self.info),
        corner_zyx=corner, **kwargs)
    canvas.init_segmentation_from_volume(self.init_seg_volume, corner, end)
    return canvas, align.Aligner().generate_alignment(corner, subvol_size)


def check_against_golden(g, request, n, exact_probs):
  path = os.path.join(request.output_directory, str(g['p%d_name' % n]))
  assert os.path.exists(path), os.listdir(request.output_directory)
  with np.load(path, allow_pickle=True) as d:
    sp = d['start_points']
    assert np.array_equal(np.array(sp[0]).reshape(-1, 3),
                          g['p%d_start_points_a' % n])
    assert np.array_equal(np.array(sp[1]).reshape(-1, 3),
                          g['p%d_start_points_b' % n])
    ref_hist = g['p%d_histories' % n]
    assert len(d['histories']) == len(ref_hist)
    for mine, ref in zip(d['histories'], ref_hist):
      assert np.array_equal(np.asarray(mine), np.asarray(ref))
    for mine, ref in zip(d['deletes'], g['p%d_deletes' % n]):
      assert np.array_equal(np.asarray(mine), np.asarray(ref))
    assert np.array_equal(d['corner_zyx'], g['p%d_corner_zyx' % n])
    assert bool(d['is_shift']) == bool(g['p%d_is_shift' % n])
    for key in ('probs', 'raw_probs'):
      mine, ref = d[key], g['p%d_%s' % (n, key)]
      assert mine.shape == ref.shape and mine.dtype == ref.dtype
      if exact_probs:
        assert np.array_equal(mine, ref)
      else:
        # quantised probability: the HIP forward differs from the f32 CPU
        # forward in the last bits of the logit -> at most one quantum
        diff = np.abs(mine.astype(np.int16) - ref.astype(np.int16))
        assert diff.max() <= 1
        assert np.mean(diff != 0) < 1e-3
    rq = request_lib.ResegmentationRequest()
    rq.ParseFromString(d['request'].item())
    assert rq.points[n].id_a == request.points[n].id_a
    counters = inference_utils.Counters()
    counters.loads(str(d['counters']))
    assert counters['edt-calls'].value == (2 if n == 0 else 1)


@pytest.fixture
def golden():
  g = np.load(os.path.join(GOLDEN, 'ref_reseg.npz'), allow_pickle=True)
  return {k: g[k] for k in g.files}


@pytest.mark.parametrize('native', [False, True])
def test_resegmentation_reproduces_reference(golden, fib25_blob, tmp_path,
                                             tmp_path_factory, native):
  """native: every segment_at of the point runs in the library's C++ loop
  (tests/native_shim.py), as it does on the GPU."""
  from tests.emulated_device import EmulatedDeviceClient, EmulatedSeeder
  from tests import native_shim
  request = build_request(golden, tmp_path)
  if native:
    native_shim.ShimHandle.shim = native_shim.build_shim(
        tmp_path_factory.mktemp('shim'))
  cls = native_shim.ShimClient if native else EmulatedDeviceClient
  native_shim.ShimHandle.total_native_calls = 0
  made = []

  def client_fn(counters):
    made.append(cls(counters, fib25_blob, 12, (33, 33, 33), (8, 8, 8)))
    return made[-1]

  runner = StandInRunner(golden, client_fn, request.inference)
  for n in range(2):
    resegmentation.process_point(request, runner, n, (1, 1, 1),
                                 seeder=EmulatedSeeder())
    check_against_golden(golden, request, n, exact_probs=True)
  # existing output -> skipped (get_target_path returns None)
  assert resegmentation.get_target_path(request, 0) is None
  assert runner.counters['resegmentation-calls'].value == 2
  assert len(made) == 2
  if native:
    assert native_shim.ShimHandle.total_native_calls >= 3


def test_process_many_batches_points_without_changing_results(golden, fib25_blob,
                                                              tmp_path):
  """process_many: the points advance concurrently in batched engine steps and
  every point's file equals what the one-at-a-time reference run wrote."""
  from tests.emulated_device import EmulatedDeviceClient, EmulatedSeeder
  request = build_request(golden, tmp_path)
  client = EmulatedDeviceClient(inference_utils.Counters(), fib25_blob, 12,
                                (33, 33, 33), (8, 8, 8))

  class EmulatedEngine:
    max_batch = 2

    def __init__(self):
      self.calls, self.pending = [], {}

    def step_submit(self, handles, reqs, params):
      self.calls.append(len(handles))
      ticket = len(self.calls)
      self.pending[ticket] = [client.step(h, q, params)
                              for h, q in zip(handles, reqs)]
      return ticket

    def step_wait(self, ticket):
      return self.pending.pop(ticket)

  runner = StandInRunner(
      golden, lambda counters: EmulatedDeviceClient(
          counters, fib25_blob, 12, (33, 33, 33), (8, 8, 8)), request.inference)
  engine = EmulatedEngine()
  for k in range(2):  # two more points next to the golden ones
    p = request.points.add()
    p.id_a = request.points[1].id_a
    p.point.z, p.point.y = request.points[1].point.z, request.points[1].point.y
    p.point.x = request.points[1].point.x + 1 + k
  resegmentation.process_many(request, runner, (1, 1, 1), engine=engine,
                              seeder=EmulatedSeeder())
  for n in range(2):
    check_against_golden(golden, request, n, exact_probs=True)
  assert len(os.listdir(str(tmp_path))) == 4
  # two steps in flight, four live points -> groups of two
  assert max(engine.calls) == 2 and not engine.pending
  assert runner.counters['resegmentation-calls'].value == 4
  # a second pass finds both outputs and does nothing
  resegmentation.process_many(request, runner, (1, 1, 1), engine=engine,
                              seeder=EmulatedSeeder())


def test_get_target_path_subdirs(tmp_path):
  import hashlib
  request = request_lib.ResegmentationRequest()
  request.output_directory = str(tmp_path)
  request.subdir_digits = 2
  p = request.points.add()
  p.id_a, p.id_b = 12, 34
  p.point.x, p.point.y, p.point.z = 1, 2, 3
  path = resegmentation.get_target_path(request, 0)
  sub = hashlib.md5(b'1234').hexdigest()[:2]
  assert path == os.path.join(str(tmp_path), sub, '12-34_at_1_2_3.npz')
  assert os.path.isdir(os.path.dirname(path))


def test_get_starting_location_excludes_neighbourhood():
  d = np.zeros((9, 9, 9))
  d[4, 4, 4] = 5
  d[4, 4, 6] = 4
  d[0, 0, 0] = 3

  class R:
    z = y = x = 2
  assert resegmentation.get_starting_location(d, R) == (4, 4, 4)
  assert d[4, 4, 6] == 0 and d[4, 4, 4] == 0
  assert resegmentation.get_starting_location(d, R) == (0, 0, 0)


def test_not_enough_context_returns_no_canvas(golden, fib25_blob, tmp_path):
  request = build_request(golden, tmp_path)
  request.points[0].point.x = 3
  runner = StandInRunner(golden, None, request.inference)
  resegmentation.process_point(request, runner, 0, (1, 1, 1), seeder=object())
  assert os.listdir(str(tmp_path)) == []


@pytest.mark.gpu
def test_gpu_edt_matches_scipy():
  from scipy import ndimage
  from ffn_amd import seeding
  rng = np.random.default_rng(3)
  seeder = seeding.default_seeder(0)
  for shape, voxel in (((49, 49, 49), (1, 1, 1)), ((40, 70, 33), (4, 1, 2.5)),
                       ((1, 5, 300), (1, 1, 1))):
    mask = ndimage.binary_dilation(rng.random(shape) < 0.01, iterations=3)
    got = seeder.edt(mask, voxel)
    want = ndimage.distance_transform_edt(mask, sampling=voxel)
    assert got.dtype == np.float64
    assert np.array_equal(got, want)
  # all-foreground: scipy's result is unbounded garbage-free only with a
  # background voxel; a single background corner pins it
  mask = np.ones((20, 20, 20), bool)
  mask[0, 0, 0] = False
  assert np.array_equal(seeder.edt(mask),
                        ndimage.distance_transform_edt(mask))


@pytest.mark.gpu
def test_gpu_resegmentation_reproduces_reference(golden, fib25_model, tmp_path):
  from ffn_amd.inference import executor
  request = build_request(golden, tmp_path)
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), fib25_model,
                                  fib25_model.info, None,
                                  inference_utils.Counters(), 1)
  runner = StandInRunner(
      golden, lambda counters: exe.get_client(counters, direct=True),
      request.inference)
  for n in range(2):
    resegmentation.process_point(request, runner, n, (1, 1, 1))
    check_against_golden(golden, request, n, exact_probs=False)


@pytest.mark.gpu
def test_gpu_process_many_reproduces_reference(golden, fib25_model, tmp_path):
  import time
  from ffn_amd.inference import executor
  request = build_request(golden, tmp_path)
  # 16 more points: the golden pair / endpoint requests again, shifted by one
  # voxel each (their own files; only the two golden ones are compared)
  for k in range(16):
    src = request.points[k % 2]
    p = request.points.add()
    p.id_a = src.id_a
    if src.HasField('id_b'):
      p.id_b = src.id_b
    p.point.z, p.point.y = src.point.z, src.point.y
    p.point.x = src.point.x + 1 + k // 2
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), fib25_model,
                                  fib25_model.info, None,
                                  inference_utils.Counters(), 8)
  runner = StandInRunner(
      golden, lambda counters: exe.get_client(counters, direct=True),
      request.inference)
  t0 = time.time()
  resegmentation.process_many(request, runner, (1, 1, 1), engine=exe.engine)
  dt = time.time() - t0
  for n in range(2):
    check_against_golden(golden, request, n, exact_probs=False)
  assert len(os.listdir(str(tmp_path))) == 18
  steps = runner.counters['update_at-calls'].value
  print('\nprocess_many: 18 points, %d FoV steps in %.3f s (%.0f steps/s)' %
        (steps, dt, steps / dt))
