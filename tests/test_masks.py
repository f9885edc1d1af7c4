"""Exclusion masks and the MovementRestrictor, against
tests/golden/ref_masks.npz -- minted by the reference's own storage.build_mask,
MovementRestrictor and Canvas (tools/make_golden.py --only masks).

Reference: storage.py:323-411 (build_mask), movement.py:247-336
(MovementRestrictor), runner.py:218-305 (make_restrictor), inference.py:497-499
and :573-577 (where the Canvas asks the restrictor).
"""

import functools
import json
import os

import numpy as np
import pytest

from ffn_amd import synthetic
from ffn_amd.inference import inference
from ffn_amd.inference import inference_utils
from ffn_amd.inference import movement
from ffn_amd.inference import request as req_lib
from ffn_amd.inference import seed as seed_lib
from ffn_amd.inference import storage
from ffn_amd.training import model as ffn_model
from tests.conftest import GOLDEN
from tests.emulated_device import EmulatedDeviceClient

FIX = os.path.join(GOLDEN, 'ref_masks.npz')


def _mask_configs():
  c0 = req_lib.MaskConfig()
  c0.coordinate_expression.expression = '(x + 2 * y > 60) & (z % 3 == 0)'
  c1 = req_lib.MaskConfig()
  ch = c1.image.channels.add()
  ch.channel = 0
  ch.min_value = 100
  ch.max_value = 140
  ch = c1.image.channels.add()
  ch.channel = 0
  ch.values = [3, 250]
  c2 = req_lib.MaskConfig()
  c2.volume.mask.hdf5 = 'unused:unused'
  ch = c2.volume.channels.add()
  ch.channel = 1
  ch.values = [2, 5]
  ch = c2.volume.channels.add()
  ch.channel = 0
  ch.min_value = 0
  ch.max_value = 1
  ch.invert = True
  c2.invert = True
  return [c0, c1, c2]


def test_build_mask_matches_reference_kats():
  g = np.load(FIX)
  configs = _mask_configs()
  corner = tuple(int(v) for v in g['bm_corner'])
  size = tuple(int(v) for v in g['bm_size'])
  vol_map = {configs[2].volume.mask.SerializeToString(): g['bm_labels']}
  for i, sel in enumerate(([0], [1], [2], [0, 1, 2])):
    got = storage.build_mask([configs[k] for k in sel], corner, size,
                             dict(vol_map), g['bm_image'])
    assert got.dtype == bool
    assert np.array_equal(got, g['build_mask_%d' % i]), i
  assert storage.build_mask([], corner, size) is None


def test_mask_messages_parse_from_text():
  r = req_lib.request_from_text('''
    image { npy: "x.npy" }
    masks { coordinate_expression { expression: "x > 3" } }
    masks { image { channels { channel: 0 min_value: 1 max_value: 2 } } invert: true }
    seed_masks { volume { mask { npy: "m.npy" } channels { channel: 0 values: 4 values: 7 } } }
    shift_mask { npy: "s.npy" }
    shift_mask_scale: 2
    shift_mask_fov { start { x: -6 y: -6 z: -4 } size { x: 13 y: 13 z: 9 } }
  ''')
  assert len(r.masks) == 2 and len(r.seed_masks) == 1
  assert r.masks[0].WhichOneof('source') == 'coordinate_expression'
  assert r.masks[1].invert and r.masks[1].image.channels[0].max_value == 2
  assert list(r.seed_masks[0].volume.channels[0].values) == [4, 7]
  assert r.shift_mask.which_volume() == 'npy' and r.shift_mask_fov.size.z == 9
  again = req_lib.request_from_text(r.SerializeToString().decode())
  assert again.SerializeToString() == r.SerializeToString()


class _Box:

  def __init__(self, start, size):
    self.start = np.array(start)
    self.end = self.start + np.array(size)


def _restrictor(g):
  return movement.MovementRestrictor(
      mask=g['run_mask'], seed_mask=g['run_seed_mask'], shift_mask=g['run_shift'],
      shift_mask_fov=_Box((-6, -6, -4), (13, 13, 9)), shift_mask_threshold=4,
      shift_mask_scale=2)


def _options():
  r = req_lib.InferenceRequest()
  o = r.inference_options
  o.init_activation = 0.95
  o.pad_value = 0.05
  o.move_threshold = 0.9
  o.segment_threshold = 0.6
  o.min_segment_size = 1000
  o.min_boundary_dist.x = 1
  o.min_boundary_dist.y = 1
  o.min_boundary_dist.z = 1
  return r


def _check_run(canvas, g):
  steps = []
  inner = canvas.update_at

  def rec(pos):
    steps.append(tuple(int(v) for v in pos))
    return inner(pos)

  canvas.update_at = rec
  canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                   coords=g['run_seeds']))
  assert steps == [tuple(int(v) for v in p) for p in g['run_steps']]
  assert np.array_equal(np.asarray(canvas.segmentation), g['run_segmentation'])
  ref = json.loads(str(g['run_counters']))
  for key in ('update_at-calls', 'skip_restriced_pos', 'skip_invalid_pos',
              'skip_threshold', 'voxels-segmented'):
    assert canvas.counters[key].value == ref[key], key


def test_restricted_canvas_reproduces_reference_run(fib25_blob):
  """A DeviceCanvas (emulated device) under a mask + seed mask + shift mask
  restrictor visits the FoV positions of the reference's Canvas, skips what it
  skipped and commits the same segments."""
  g = np.load(FIX)
  r = _options()
  info = ffn_model.ModelInfo(np.array([8, 8, 8]), np.array([33, 33, 33]),
                             np.array([33, 33, 33]), np.array([33, 33, 33]))
  client = EmulatedDeviceClient(inference_utils.Counters(), fib25_blob, 12,
                                (33, 33, 33), (8, 8, 8))
  canvas = inference.make_canvas(
      info, client, synthetic.normalize(g['run_volume']), r.inference_options,
      restrictor=_restrictor(g),
      movement_policy_fn=movement.get_policy_fn(r, info))
  assert isinstance(canvas, inference.DeviceCanvas)
  assert not canvas._native_loop_ok()  # the restrictor is a Python hook
  _check_run(canvas, g)


def test_runner_builds_the_restrictor_from_the_request(tmp_path, fib25_model):
  """Runner.make_restrictor: masks / seed_masks / shift_mask of the request ->
  MovementRestrictor (npy volumes), ALL_MASKED when nothing is left."""
  from ffn_amd.inference import align
  from ffn_amd.inference import runner as runner_lib
  g = np.load(FIX)
  shape = g['run_mask'].shape
  np.save(tmp_path / 'mask.npy', g['run_mask'].astype(np.uint8))
  np.save(tmp_path / 'shift.npy', g['run_shift'])
  r = _options()
  m = r.masks.add()
  ch = m.volume.channels.add()
  ch.channel = 0
  ch.values = [1]
  m.volume.mask.npy = str(tmp_path / 'mask.npy')
  sm = r.seed_masks.add()
  sm.coordinate_expression.expression = '(z < 24) & (x < 30)'
  r.shift_mask.npy = str(tmp_path / 'shift.npy')
  r.shift_mask_scale = 2
  r.shift_mask_fov.start.x, r.shift_mask_fov.start.y, r.shift_mask_fov.start.z = (
      -6, -6, -4)
  r.shift_mask_fov.size.x, r.shift_mask_fov.size.y, r.shift_mask_fov.size.z = (
      13, 13, 9)
  run = runner_lib.Runner.__new__(runner_lib.Runner)
  run.request = r
  run.counters = inference_utils.Counters()
  run._mask_volumes = {}
  run._shift_mask_volume = storage.decorated_volume(r.shift_mask)
  run._model_info = fib25_model.info
  corner, size = (0, 0, 0), shape
  got = run.make_restrictor(corner, size, None, align.Alignment(corner, size))
  want = _restrictor(g)
  assert np.array_equal(got.mask, want.mask)
  assert np.array_equal(got.seed_mask, want.seed_mask)
  assert np.array_equal(got.shift_mask, want.shift_mask)
  for pos in [(30, 20, 20), (41, 29, 37), (25, 19, 49), (10, 10, 60), (40, 40, 40)]:
    assert got.is_valid_pos(pos) == want.is_valid_pos(pos), pos
    assert got.is_valid_seed(pos) == want.is_valid_seed(pos), pos
  # everything masked -> ALL_MASKED
  r2 = _options()
  r2.masks.add().coordinate_expression.expression = 'z >= 0'
  run.request = r2
  run._shift_mask_volume = None
  assert run.make_restrictor(corner, size, None,
                             align.Alignment(corner, size)) == run.ALL_MASKED


@pytest.mark.gpu
def test_restricted_canvas_on_the_gpu(fib25_model):
  from ffn_amd.inference import executor
  g = np.load(FIX)
  r = _options()
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), fib25_model,
                                  fib25_model.info, None,
                                  inference_utils.Counters(), 1, device_id=0)
  try:
    counters = inference_utils.Counters()
    canvas = inference.DeviceCanvas(
        fib25_model.info, exe.get_client(counters, direct=True),
        synthetic.normalize(g['run_volume']), r.inference_options,
        counters=counters, restrictor=_restrictor(g),
        movement_policy_fn=movement.get_policy_fn(r, fib25_model.info))
    _check_run(canvas, g)
    canvas.close()
  finally:
    exe.engine.close()
