"""Host-side mirror of the reference API (no GPU): Canvas / DeviceCanvas logic,
movement policy, executor protocol, storage formats, request parsing."""

import functools
import json
import os
import threading

import numpy as np
import pytest

from ffn_amd import _lib
from ffn_amd import synthetic
from ffn_amd.inference import align
from ffn_amd.inference import executor
from ffn_amd.inference import inference
from ffn_amd.inference import inference_utils
from ffn_amd.inference import movement
from ffn_amd.inference import request as req_lib
from ffn_amd.inference import seed as seed_lib
from ffn_amd.inference import segmentation
from ffn_amd.inference import storage
from ffn_amd.training import model as ffn_model
from ffn_amd.training.models import convstack_3d
from oracle import ffn_oracle
from tests.conftest import GOLDEN
from tests.emulated_device import EmulatedDeviceClient

SAMPLE_PBTXT = '''
image {
  hdf5: "third_party/neuroproof_examples/training_sample2/grayscale_maps.h5:raw"
}
image_mean: 128
image_stddev: 33
checkpoint_interval: 1800
seed_policy: "PolicyPeaks"
model_checkpoint_path: "models/fib25/model.ckpt-27465036"
model_name: "convstack_3d.ConvStack3DFFNModel"
model_args: "{\\"depth\\": 12, \\"fov_size\\": [33, 33, 33], \\"deltas\\": [8, 8, 8]}"
segmentation_output_dir: "results/fib25/training2"
inference_options {
  init_activation: 0.95
  pad_value: 0.05
  move_threshold: 0.9
  min_boundary_dist { x: 1 y: 1 z: 1}
  segment_threshold: 0.6
  min_segment_size: 1000
}
'''


def _info():
  return ffn_model.ModelInfo(np.array([8, 8, 8]), np.array([33, 33, 33]),
                             np.array([33, 33, 33]), np.array([33, 33, 33]))


def _request():
  return req_lib.request_from_text(SAMPLE_PBTXT)


def test_request_parsing_matches_reference_semantics():
  r = _request()
  assert r.image.which_volume() == 'hdf5'
  assert r.image_mean == 128.0 and r.image_stddev == 33.0
  assert json.loads(r.model_args)['depth'] == 12
  o = r.inference_options
  assert o.min_boundary_dist.x == 1 and o.min_segment_size == 1000
  assert o.move_threshold == float(np.float32(0.9))  # f32 proto field
  assert not o.HasField('disco_seed_threshold')
  assert o.disco_seed_threshold == 0.0  # default => disco bias ON
  assert not r.HasField('init_segmentation')
  r2 = req_lib.InferenceRequest()
  req_lib.parse_text(r.SerializeToString().decode(), r2)
  assert r2 == r
  with open(os.path.join(GOLDEN, 'ref_misc.json')) as f:
    k = json.load(f)
  lo = inference._logit_options(o)
  for name in ('init_activation', 'pad_value', 'move_threshold',
               'segment_threshold'):
    assert getattr(lo, name) == k['logit_' + name]
  fn = movement.get_policy_fn(r, _info())

  class C:
    pass

  c = C()
  assert fn(c).score_threshold == k['policy_threshold']


def test_move_scoring_and_face_prediction_match_reference_kats():
  g = np.load(os.path.join(GOLDEN, 'ref_movement.npz'))
  thr = float(g['threshold'])
  names = sorted({k[:-len('_map')] for k in g.files if k.endswith('_map')})
  for name in names:
    deltas, pm = g[name + '_deltas'], g[name + '_map']
    res = sorted(movement.get_scored_move_offsets(deltas, pm, thr),
                 reverse=True)
    assert [r[1] for r in res] == [tuple(o) for o in g[name + '_offsets']]
    # the device path: face maxima -> FacePrediction -> identical moves
    scores, idx = ffn_oracle.face_maxima(deltas, pm)
    fp = movement.FacePrediction([float(s) for s in scores],
                                 [int(i) for i in idx], [0] * 6, pm.shape)
    res2 = sorted(((s, o) for s, o, _ in fp.scored_move_offsets(deltas, thr)),
                  reverse=True)
    assert [r[1] for r in res2] == [tuple(o) for o in g[name + '_offsets']]
    assert np.array_equal(np.array([r[0] for r in res2], np.float32),
                          g[name + '_scores'])


def test_quantize_pos_and_queue_state_roundtrip():
  with open(os.path.join(GOLDEN, 'ref_misc.json')) as f:
    k = json.load(f)

  class C:

    def is_valid_pos(self, pos):
      return True

  c = C()
  pol = movement.FaceMaxMovementPolicy(c, deltas=(8, 8, 8),
                                       score_threshold=2.0)
  pol.reset_state((100, 100, 100))
  for p, qv in k['quantize_pos'].items():
    pos = tuple(int(v) for v in p.strip('()').split(','))
    assert list(pol.quantize_pos(pos)) == qv
  pol.append((4.0, (108, 100, 100)))
  pol.append((3.0, [100, 92, 100]))
  state = pol.get_state()
  queue = state[0][0]
  assert [tuple(c) for _, c in queue] == [(108, 100, 100), (100, 92, 100)]
  pol2 = movement.FaceMaxMovementPolicy(c, deltas=(8, 8, 8),
                                        score_threshold=2.0)
  pol2.restore_state(state)
  assert next(pol2) == (108, 100, 100)
  pol2.done_rounded_coords.add(pol2.quantize_pos((100, 92, 100)))
  assert pol2.peek_candidates(4) == []
  with pytest.raises(StopIteration):
    next(pol2)


class _OracleClient(executor.ExecutorClient):

  def __init__(self, blob):
    super().__init__(inference_utils.Counters(), None)
    self.blob = blob

  def start(self):
    return 0

  def finish(self):
    pass

  def predict(self, seed, image, fetches):
    return {'logits': ffn_oracle.forward(image, seed, self.blob, 12)[..., None]}


@pytest.mark.parametrize('device', [False, True])
def test_canvas_reproduces_reference_run(fib25_blob, device):
  """ffn_amd's Canvas (host arrays, predict contract) and DeviceCanvas (with
  the emulated device) == the reference's Canvas.segment_all on cells56."""
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells56.npz'))
  r = _request()
  info = _info()
  image = synthetic.normalize(g['volume'])
  if device:
    client = EmulatedDeviceClient(inference_utils.Counters(), fib25_blob, 12,
                                  (33, 33, 33), (8, 8, 8))
    canvas = inference.make_canvas(info, client, image, r.inference_options,
                                   movement_policy_fn=movement.get_policy_fn(
                                       r, info))
    assert isinstance(canvas, inference.DeviceCanvas)
  else:
    canvas = inference.make_canvas(info, _OracleClient(fib25_blob), image,
                                   r.inference_options,
                                   movement_policy_fn=movement.get_policy_fn(
                                       r, info))
    assert type(canvas) is inference.Canvas
  canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                   coords=g['seeds']))
  if device:  # the between-segment turns went through ffn_canvas_segment_turn
    assert canvas.turns > 0 and canvas._handle.turns == canvas.turns
  assert np.array_equal(np.asarray(canvas.segmentation), g['segmentation'])
  assert np.array_equal(np.asarray(canvas.seed), g['seed_logits'],
                        equal_nan=True)
  ref = json.loads(str(g['counters']))
  for key in ('update_at-calls', 'voxels-segmented', 'voxels-overlapping',
              'skip_invalid_pos', 'inference-calls', 'segment_at-loop-calls'):
    assert canvas.counters[key].value == ref[key], key
  origins = json.loads(str(g['origins']))
  assert {int(k): [list(v.start_zyx), v.iters]
          for k, v in canvas.origins.items()} == {
              int(k): v for k, v in origins.items()}


def _flood_forward(image, seed, blob, depth):
  """A cheap stand-in for the conv stack: the bright component(s) of the FoV
  that the seed touches (enough to drive whole segment_all runs through the HOST
  logic in milliseconds per step)."""
  del blob, depth
  from scipy import ndimage
  lab, _ = ndimage.label(image > 0.5)
  hit = np.unique(lab[(seed > 0.0) & (lab > 0)])
  return np.where(np.isin(lab, hit) & (lab > 0), 4.0, -4.0).astype(np.float32)


def _blob_volume(shape, centres, radius):
  zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing='ij')
  vol = np.zeros(shape, np.float32)
  for cz, cy, cx in centres:
    vol[(zz - cz) ** 2 + (yy - cy) ** 2 + (xx - cx) ** 2 <= radius ** 2] = 1.0
  return vol


@pytest.mark.parametrize('min_size', [200, 3500])
def test_segment_turn_answers_the_seed_loop_like_single_questions(monkeypatch,
                                                                  min_size):
  """DeviceCanvas with the between-segment turn as ONE device call
  (ffn_canvas_segment_turn: commit, -1 markers, the next seeds tested ahead,
  init_seed) == the same canvas asking one question per call, as the reference's
  loop does (inference.py:573-660): segmentation including its -1 markers, seed,
  counters, origins, overlaps -- on a run with committed objects, objects that
  are too small, weak seeds, seeds inside objects and seeds next to them."""
  monkeypatch.setattr(ffn_oracle, 'forward', _flood_forward)
  shape = (72, 96, 96)
  centres = [(30, 30, 30), (34, 60, 34), (40, 40, 64), (36, 66, 66)]
  image = _blob_volume(shape, centres, 9)
  r = _request()
  r.inference_options.min_segment_size = min_size
  info = _info()
  # a dense grid of seeds, bright and dark ones, in raster order
  grid = np.array([(z, y, x) for z in range(18, 54, 6) for y in range(18, 78, 6)
                   for x in range(18, 78, 6)], np.int32)
  runs = []
  for cands in (inference.DeviceCanvas.TURN_CANDIDATES, 7, 0):
    client = EmulatedDeviceClient(inference_utils.Counters(), None, 12,
                                  (33, 33, 33), (8, 8, 8))
    canvas = inference.make_canvas(info, client, image, r.inference_options,
                                   movement_policy_fn=movement.get_policy_fn(
                                       r, info))
    canvas.TURN_CANDIDATES = cands
    canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                     coords=grid))
    h = canvas._handle
    assert (canvas.turns > 0) == (cands > 0)
    counters = {k: c.value for k, c in canvas.counters
                if not k.endswith('-ms')}
    runs.append((np.asarray(canvas.segmentation).copy(),
                 np.asarray(canvas.seed).copy(), counters,
                 {k: (tuple(v.start_zyx), v.iters)
                  for k, v in canvas.origins.items()},
                 {k: np.asarray(v).tolist() for k, v in canvas.overlaps.items()},
                 h.point_reads))
  ref = runs[-1]
  assert (len(ref[3]) >= 2 if min_size == 200 else not ref[3])  # objects
  assert (ref[0] == -1).any()
  for run in runs[:-1]:
    assert np.array_equal(run[0], ref[0])
    assert np.array_equal(run[1], ref[1], equal_nan=True)
    assert run[2] == ref[2] and run[3] == ref[3] and run[4] == ref[4]
  # ... with far fewer device questions
  assert runs[0][5] < ref[5]


def test_segment_turn_and_timed_checkpoints(monkeypatch, tmp_path):
  """A canvas with a timed checkpoint configured (every Runner canvas: the sample
  config asks for one every 1800 s) still uses the device-side turn -- except
  when the checkpoint is about to be taken, which then happens between the
  single questions of the reference's loop, as before."""
  monkeypatch.setattr(ffn_oracle, 'forward', _flood_forward)
  shape = (72, 96, 96)
  image = _blob_volume(shape, [(30, 30, 30), (34, 60, 34), (40, 40, 64)], 9)
  r = _request()
  r.inference_options.min_segment_size = 200
  info = _info()
  grid = np.array([(z, y, x) for z in range(18, 54, 12) for y in range(18, 78, 6)
                   for x in range(18, 78, 6)], np.int32)
  runs = {}
  for interval in (1800.0, 1e-6):
    path = str(tmp_path / ('cp_%g.npz' % interval))
    client = EmulatedDeviceClient(inference_utils.Counters(), None, 12,
                                  (33, 33, 33), (8, 8, 8))
    canvas = inference.make_canvas(info, client, image, r.inference_options,
                                   movement_policy_fn=movement.get_policy_fn(
                                       r, info),
                                   checkpoint_path=path,
                                   checkpoint_interval_sec=interval)
    canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                     coords=grid))
    runs[interval] = (np.asarray(canvas.segmentation).copy(), canvas.turns,
                      os.path.exists(path))
  assert runs[1800.0][1] > 0 and not runs[1800.0][2]
  assert runs[1e-6][1] == 0 and runs[1e-6][2]
  assert np.array_equal(runs[1800.0][0], runs[1e-6][0])
  assert len(np.unique(runs[1800.0][0])) >= 3
  # The guard is deterministic in both directions: a checkpoint that
  # becomes due while a turn's answers are still in use waits for the next
  # opportunity, and the seed policy may list a seed twice.
  path = str(tmp_path / 'cp_due.npz')
  client = EmulatedDeviceClient(inference_utils.Counters(), None, 12, (33, 33, 33),
                                (8, 8, 8))
  canvas = inference.make_canvas(info, client, image, r.inference_options,
                                 movement_policy_fn=movement.get_policy_fn(r, info),
                                 checkpoint_path=path, checkpoint_interval_sec=1800.0)
  saves = []
  monkeypatch.setattr(type(canvas), 'save_checkpoint',
                      lambda self, p, **kw: saves.append(self._turn_rec))
  inner = canvas._turn

  def turn_then_due(*a, **kw):  # the interval runs out right behind a turn
    out = inner(*a, **kw)
    canvas.checkpoint_last = -1e9
    return out

  canvas._turn = turn_then_due
  doubled = np.concatenate([grid[:40], grid[30:]])
  canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                   coords=doubled))
  assert canvas.turns > 0 and saves and all(rec is None for rec in saves)
  assert np.array_equal(np.asarray(canvas.segmentation), runs[1800.0][0])


def test_keep_history_host_and_device_canvas_agree(fib25_blob):
  """keep_history (inference.py:420-423, 520-521): the device canvas gets the
  per-step deleted-voxel count from the step result."""
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells56.npz'))
  r = _request()
  info = _info()
  image = synthetic.normalize(g['volume'])
  runs = []
  for device in (False, True):
    client = (EmulatedDeviceClient(inference_utils.Counters(), fib25_blob, 12,
                                   (33, 33, 33), (8, 8, 8)) if device else
              _OracleClient(fib25_blob))
    canvas = inference.make_canvas(info, client, image, r.inference_options,
                                   movement_policy_fn=movement.get_policy_fn(
                                       r, info), keep_history=True)
    n = canvas.segment_at(tuple(int(v) for v in g['seeds'][0]))
    assert len(canvas.history) == n == len(canvas.history_deleted)
    runs.append(([tuple(int(v) for v in p) for p in canvas.history],
                 [int(v) for v in canvas.history_deleted]))
  assert runs[0] == runs[1] and len(runs[0][0]) > 0


def test_device_canvas_checkpoint_roundtrip(fib25_blob, tmp_path):
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells56.npz'))
  r = _request()
  info = _info()
  image = synthetic.normalize(g['volume'])

  def make():
    client = EmulatedDeviceClient(inference_utils.Counters(), fib25_blob, 12,
                                  (33, 33, 33), (8, 8, 8))
    return inference.make_canvas(info, client, image, r.inference_options,
                                 movement_policy_fn=movement.get_policy_fn(
                                     r, info))

  a = make()
  a.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                              coords=g['seeds']))
  path = str(tmp_path / 'x' / 'seg.cpoint')
  a.save_checkpoint(path, 0)
  b = make()
  assert b.restore_checkpoint(path) == 0
  assert np.array_equal(np.asarray(b.segmentation), g['segmentation'])
  assert np.array_equal(np.asarray(b.seed), g['seed_logits'], equal_nan=True)
  assert b._max_id == a._max_id and set(b.origins) == set(a.origins)
  # the device arrays are VIEWS of the canvas' memory: a host copy outlives the
  # canvas, the view says so instead of dangling
  seg_view, host_copy = b.segmentation, np.asarray(b.segmentation)
  b.close()
  assert np.array_equal(host_copy, g['segmentation'])
  with pytest.raises(RuntimeError, match='canvas closed'):
    seg_view[0, 0, 0]
  with pytest.raises(RuntimeError, match='canvas closed'):
    np.asarray(seg_view)


def test_mid_segment_checkpoint_resume_emulated_device(fib25_blob, tmp_path):
  """Killed in the middle of a segment, continued from the .cpoint in a fresh
  canvas: same final state as the uninterrupted run (inference.py:728-843)."""
  from tests import resume_case
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))
  r = _request()
  info = _info()
  image = synthetic.normalize(g['volume'])

  def make(path, interval):
    client = EmulatedDeviceClient(inference_utils.Counters(), fib25_blob, 12,
                                  (33, 33, 33), (8, 8, 8))
    return inference.make_canvas(info, client, image, r.inference_options,
                                 movement_policy_fn=movement.get_policy_fn(
                                     r, info),
                                 checkpoint_path=path,
                                 checkpoint_interval_sec=interval)

  # (the uninterrupted run is the reference-minted fixture itself)
  _, b, meta = resume_case.run_resume_case(make, g['seeds'], 30, tmp_path,
                                           run_uninterrupted=False)
  assert meta['partial_segment_iters'] > 0  # it really was mid-segment
  assert np.array_equal(np.asarray(b.segmentation), g['segmentation'])
  assert np.array_equal(np.asarray(b.seed), g['seed_logits'], equal_nan=True)
  origins = json.loads(str(g['origins']))
  assert {int(k): [list(v.start_zyx), v.iters]
          for k, v in b.origins.items()} == {int(k): v
                                             for k, v in origins.items()}
  assert b.counters['update_at-calls'].value == len(g['steps'])


def test_normalized_u8_image_equals_host_normalisation():
  """NormalizedU8Image stands for (u8 -> f32 - mean) / stddev (runner.py:383-385)."""
  rng = np.random.RandomState(0)
  raw = rng.randint(0, 256, (9, 10, 11)).astype(np.uint8)
  for mean, std in ((128, 33), (127.3, 28.9)):
    want = (raw.astype(np.float32) - mean) / std
    img = inference.NormalizedU8Image(raw, mean, std)
    assert img.shape == raw.shape and img.dtype == np.float32
    assert np.array_equal(np.asarray(img), want)
    assert np.array_equal(img[2:5, 1:, ::2], want[2:5, 1:, ::2])
    assert want.dtype == np.float32


def test_threaded_executor_protocol_batches_and_terminates():
  """The reference's queue protocol: N clients > batch_size, partial batches are
  not padded, server exits once every expected client came and went."""
  info = _info()
  counters = inference_utils.Counters()
  iface = executor.ExecutorInterface()
  batches = []

  class FakeExec(executor.ThreadingBatchExecutor):

    def _schedule_batch(self, client_ids, fetches):
      n = len(client_ids)
      batches.append(n)
      out = self.input_seed[:n] + self.input_image[:n]
      self._deliver(client_ids, [{'logits': out[i].copy()} for i in range(n)])

  exe = FakeExec(iface, None, info, None, counters, batch_size=2,
                 expected_clients=3)
  exe.start_server()
  results = {}

  def worker(k):
    cl = exe.get_client(counters)
    cl.start()
    seed = np.full((33, 33, 33), float(k), np.float32)
    for _ in range(5):
      out = cl.predict(seed, seed, ['logits'])['logits']
      assert out.shape == (33, 33, 33, 1)
      assert float(out[0, 0, 0, 0]) == 2.0 * k
    cl.finish()
    results[k] = True

  threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=60)
  exe.th_executor.join(timeout=30)
  assert not exe.th_executor.is_alive()
  assert sorted(results) == [0, 1, 2]
  assert max(batches) <= 2 and sum(batches) == 15
  exe.stop_server()


def test_storage_formats_match_reference(tmp_path):
  with open(os.path.join(GOLDEN, 'ref_misc.json')) as f:
    k = json.load(f)
  q = storage.quantize_probability(np.array([0, .001, .5, .6, .95, 1, np.nan]))
  assert [int(x) for x in q] == k['quantize_out']
  dq = storage.dequantize_probability(np.array([0, 1, 128, 255]))
  assert [None if np.isnan(x) else float(x) for x in dq] == k['dequantize_out']
  assert storage.subvolume_path('out', (3, 2, 1), 'npz') == k['subvolume_path']
  assert storage.checkpoint_path('out', (3, 2, 1)) == k['checkpoint_path']
  assert storage.object_prob_path('out', (3, 2, 1)) == k['object_prob_path']
  for m, dt in k['reduce_id_bits'].items():
    assert str(segmentation.reduce_id_bits(np.array([0, int(m)])).dtype) == dt
  seg = np.zeros((4, 5, 6), np.int32)
  seg[1:3] = 7
  origins = {7: storage.OriginInfo((1, 2, 3), 11, 0.5)}
  path = storage.segmentation_path(str(tmp_path), (0, 0, 0))
  storage.save_subvolume(seg, origins, path, request=b'x', counters='{}',
                         overlaps={})
  with np.load(path, allow_pickle=True) as d:
    assert sorted(d.files) == ['counters', 'origins', 'overlaps', 'request',
                               'segmentation']
    assert d['segmentation'].dtype == np.uint8
  out, org = storage.load_segmentation(str(tmp_path), (0, 0, 0),
                                           split_cc=False)
  assert np.array_equal(out, seg) and org[7].iters == 11
  arr = storage.NumpyArray(shape=(2, 2, 2), dtype=np.float32,
                           default_value=np.nan)
  assert np.isnan(arr).all()
  arr[0, 0, 0] = 1
  arr.clear()
  assert np.isnan(arr).all()


def test_grid_seed_policy_and_alignment():
  with open(os.path.join(GOLDEN, 'ref_misc.json')) as f:
    k = json.load(f)

  class C:
    pass

  c = C()
  c.image = np.zeros((50, 56, 60), np.uint8)
  c.shape = c.image.shape
  c.margin = np.array([16, 16, 16])
  pol = seed_lib.PolicyGrid3d(c, step=16, offsets=(0, 8))
  assert [list(p) for p in pol] == k['grid3d_seeds']
  al = align.Aligner().generate_alignment((0, 0, 0), (4, 4, 4))
  src = np.arange(64).reshape(4, 4, 4)
  assert al.align_and_crop((0, 0, 0), src, (0, 0, 0), (4, 4, 4)) is src
  out = al.align_and_crop((1, 1, 1), src, (0, 0, 0), (4, 4, 4), fill=-1)
  assert out[0, 0, 0] == -1 and out[1, 1, 1] == src[0, 0, 0]


def test_model_geometry_and_weight_blob(fib25_variables):
  m = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8],
                                       batch_size=1, depth=12)
  assert tuple(m.info.pred_mask_size) == (33, 33, 33)
  m.set_variables(fib25_variables)
  blob = m.weights_blob()
  assert blob.size == 638433
  assert np.array_equal(blob, ffn_oracle.weights_blob(fib25_variables, 12))
  bad = dict(fib25_variables)
  bad['seed_update/conv3_a/weights'] = np.zeros((3, 3, 3, 16, 32), np.float32)
  with pytest.raises(ValueError):
    m.set_variables(bad)


def test_step_struct_layout_matches_header():
  """ctypes mirrors of the C structs: sizes must match include/ffn_hip.h."""
  import ctypes
  assert ctypes.sizeof(_lib.StepParams) == 16
  assert ctypes.sizeof(_lib.StepRequest) == 4 * (3 + 3 + 1 + 3 * 16)
  assert ctypes.sizeof(_lib.StepResult) == 4 * (6 + 6 + 6 + 3 + 16 + 16 + 2)
  assert ctypes.sizeof(_lib.CommitCounts) == 24


def test_policy_peaks_matches_reference_with_real_skimage():
  """The PolicyPeaks oracle (scipy) and the product's PolicyPeaks host logic
  (sorting, margin filter; device emulated) == the reference's PolicyPeaks run
  with scikit-image 0.18.3 (fixture minted by tools/make_golden_peaks.py)."""
  from oracle import seeds_oracle
  from tests.emulated_device import EmulatedSeeder
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
    got = np.array([p for p in seed_lib.PolicyPeaks(
        c, seeder=EmulatedSeeder())]).reshape(-1, 3)
    assert len(got) > 50
    assert np.array_equal(got, g[n + '_seeds']), n
    # the explicit-arithmetic restatement the HIP kernels follow == scipy
    stages = {}
    seeds_oracle.policy_peaks(c.image, c.segmentation > 0, stages=stages)
    edges = seeds_oracle.gradient_magnitude_exact(c.image)
    assert np.array_equal(edges, stages['edges'])
    assert np.array_equal(seeds_oracle.gaussian_exact(edges), stages['thresh'])


def test_multi_canvas_driver_interleaves_without_changing_results(fib25_blob):
  """Single-threaded batching scheduler: canvases advanced round-robin in
  batches reproduce, each on its own, the reference runs."""
  r = _request()
  info = _info()
  client = EmulatedDeviceClient(inference_utils.Counters(), fib25_blob, 12,
                                (33, 33, 33), (8, 8, 8))

  class EmulatedEngine:
    max_batch = 2
    calls = []

    def __init__(self):
      self.calls = []
      self.pending = {}
      self.max_in_flight = 0

    def step_submit(self, handles, reqs, params):
      # like the device, the emulation executes steps in submission order;
      # results are handed out at wait time
      self.calls.append(len(handles))
      ticket = len(self.calls)
      assert len(self.pending) < 2, 'more than two steps in flight'
      busy = {id(h) for hs, _ in self.pending.values() for h in hs}
      assert not busy & {id(h) for h in handles}, 'canvas in two steps'
      self.pending[ticket] = (list(handles), [
          client.step(h, q, params) for h, q in zip(handles, reqs)])
      self.max_in_flight = max(self.max_in_flight, len(self.pending))
      return ticket

    def step_wait(self, ticket):
      return self.pending.pop(ticket)[1]

  for overlap, batch, names in (
      (True, 2, ['cells56', 'cells72', 'cells56', 'cells56', 'cells56']),
      (False, 2, ['cells56', 'cells56', 'cells56']),
      (True, 8, ['cells56', 'cells72', 'cells56', 'cells56', 'cells56'])):
    jobs, canvases = [], []
    for n in names:
      g = np.load(os.path.join(GOLDEN, 'ref_canvas_%s.npz' % n))
      c = inference.make_canvas(info, client, synthetic.normalize(g['volume']),
                                r.inference_options,
                                movement_policy_fn=movement.get_policy_fn(r,
                                                                         info))
      canvases.append((c, g))
      jobs.append((c, functools.partial(seed_lib.PolicyFixed,
                                        coords=g['seeds'])))
    eng = EmulatedEngine()
    drv = inference.MultiCanvasDriver(eng, batch_size=batch, overlap=overlap)
    drv.run(jobs)
    for c, g in canvases:
      assert np.array_equal(np.asarray(c.segmentation), g['segmentation'])
      assert c.counters['update_at-calls'].value == len(g['steps'])
    assert drv.steps == sum(len(g['steps']) for _, g in canvases)
    assert not eng.pending
    assert max(eng.calls) <= batch
    assert eng.max_in_flight == (2 if overlap else 1)
    if overlap and batch == 8:  # 5 live canvases -> groups of 3 and 2
      assert eng.calls[:2] == [3, 2]


def test_update_at_override_is_honoured(fib25_blob):
  """A subclass hooking update_at (as the reference's loop allows) still sees
  every FoV step."""
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells56.npz'))
  r = _request()
  info = _info()
  client = EmulatedDeviceClient(inference_utils.Counters(), fib25_blob, 12,
                                (33, 33, 33), (8, 8, 8))
  seen = []

  class Hooked(inference.DeviceCanvas):

    def update_at(self, pos):
      seen.append(tuple(pos))
      return super().update_at(pos)

  c = Hooked(info, client, synthetic.normalize(g['volume']),
             r.inference_options,
             movement_policy_fn=movement.get_policy_fn(r, info))
  c.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                              coords=g['seeds']))
  assert np.array_equal(np.array(seen), g['steps'])
  assert np.array_equal(np.asarray(c.segmentation), g['segmentation'])


def test_resegmentation_request_messages_round_trip():
  """ResegmentationRequest / ResegmentationPoint (inference.proto:296-357):
  repeated message fields with add(), presence of the optional id_b, and the
  SerializeToString / ParseFromString pair the result files rely on."""
  import copy
  rq = req_lib.ResegmentationRequest()
  assert len(rq.points) == 0 and not rq.HasField('analysis_radius')
  p = rq.points.add()
  p.id_a, p.id_b = 7, 9
  p.point.x, p.point.y, p.point.z = 1, 2, 3
  q = rq.points.add(id_a=11)
  assert not q.HasField('id_b') and q.id_b == 0
  rq.radius.x = rq.radius.y = rq.radius.z = 24
  rq.inference.inference_options.move_threshold = 0.9
  rq.output_directory = '/tmp/out dir'
  assert rq.max_retry_iters == 1 and rq.subdir_digits == 0  # proto defaults
  back = req_lib.ResegmentationRequest()
  assert back.ParseFromString(rq.SerializeToString()) > 0
  assert back == rq and len(back.points) == 2
  assert back.points[0].point.z == 3 and not back.points[1].HasField('id_b')
  assert isinstance(back.points, list) and hasattr(back.points, 'add')
  clone = copy.deepcopy(rq)
  clone.points.add(id_a=1)
  assert len(rq.points) == 2 and len(clone.points) == 3
  # a plain list assigned to a repeated field gains add()
  rq.points = [req_lib.ResegmentationPoint(id_a=5)]
  rq.points.add(id_a=6)
  assert [pt.id_a for pt in rq.points] == [5, 6]


def test_update_seed_pads_a_smaller_prediction_around_the_centre():
  """FFNModel.update_seed (reference model.py:168-183): pred == seed adds in
  place; pred < seed zero-pads the update dz // 2 in front, the rest behind."""
  info = ffn_model.ModelInfo(np.array([8, 8, 8]), np.array([5, 7, 4]),  # xyz
                             np.array([8, 8, 7]), np.array([8, 8, 7]))
  m = ffn_model.FFNModel(info)
  seed = np.zeros((1, 7, 8, 8, 1), np.float32)
  upd = np.ones((1, 4, 7, 5, 1), np.float32)
  out = m.update_seed(seed, upd)
  assert out is seed and out.sum() == upd.size
  # dz = 3 -> 1 in front, 2 behind; dy = 1 -> 0 / 1; dx = 3 -> 1 / 2
  assert np.all(seed[0, 1:5, 0:7, 1:6, 0] == 1)
  assert seed[0, 0].sum() == 0 and seed[0, 5:].sum() == 0
  assert seed[0, :, 7].sum() == 0 and seed[0, :, :, 0].sum() == 0
  same = ffn_model.FFNModel(_info())
  s2 = np.full((33, 33, 33), 2.0, np.float32)
  assert np.all(same.update_seed(s2, np.ones_like(s2)) == 3.0)


@pytest.mark.parametrize('name', ['cells56_pred25', 'cells72_pred27'])
@pytest.mark.parametrize('device', [False, True])
def test_pred_smaller_than_seed_reproduces_reference_run(fib25_blob, name, device):
  """ModelInfo.pred_mask_size < input_seed_size (reference model.py:168-183,
  inference.py:218,410-411): the reference's Canvas, driven with a network that
  hands back the centred pred box of (seed + update), minted
  tests/golden/ref_canvas_<name>.npz (tools/make_golden.py --only predcrop); the
  host Canvas and the DeviceCanvas (emulated device: the specification of
  ffn_engine_set_pred_size) reproduce it."""
  path = os.path.join(GOLDEN, 'ref_canvas_%s.npz' % name)
  if not os.path.exists(path):
    pytest.skip('fixture not minted')
  g = np.load(path)
  pred = tuple(int(v) for v in g['pred_zyx'])
  assert pred != (33, 33, 33)
  r = _request()
  info = ffn_model.ModelInfo(np.array([8, 8, 8]), np.array(pred[::-1]),
                             np.array([33, 33, 33]), np.array([33, 33, 33]))
  image = synthetic.normalize(g['volume'])
  if device:
    client = EmulatedDeviceClient(inference_utils.Counters(), fib25_blob, 12,
                                  (33, 33, 33), (8, 8, 8), pred_zyx=pred)
  else:
    lo = [(33 - p) // 2 for p in pred]
    box = tuple(slice(l, l + p) for l, p in zip(lo, pred))

    class Cropping(_OracleClient):

      def predict(self, seed, image, fetches):
        out = ffn_oracle.forward(image, seed, self.blob, 12)
        return {'logits': np.ascontiguousarray(out[box])[..., None]}

    client = Cropping(fib25_blob)
  canvas = inference.make_canvas(info, client, image, r.inference_options,
                                 movement_policy_fn=movement.get_policy_fn(r, info))
  assert isinstance(canvas, inference.DeviceCanvas) == device
  canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                   coords=g['seeds']))
  assert np.array_equal(np.asarray(canvas.segmentation), g['segmentation'])
  assert np.array_equal(np.asarray(canvas.seed), g['seed_logits'], equal_nan=True)
  ref = json.loads(str(g['counters']))
  for key in ('update_at-calls', 'voxels-segmented', 'voxels-overlapping',
              'skip_invalid_pos', 'segment_at-loop-calls'):
    assert canvas.counters[key].value == ref[key], key
