"""The library's host-side segment loop (ffn_amd/csrc/ffn_host_loop.h ->
ffn_canvas_segment_at) against the Python loop and the reference-minted runs.

The loop is a C++ template over the device; tests/host_loop_shim.cpp (built
here with g++) instantiates it over callbacks into the emulated device, so the
very code libffn_hip.so runs over the HIP canvas is exercised on the CPU."""
import ctypes
import functools
import os

import numpy as np
import pytest

from ffn_amd import _lib
from ffn_amd import synthetic
from ffn_amd.inference import inference
from ffn_amd.inference import inference_utils
from ffn_amd.inference import movement
from ffn_amd.inference import request as request_lib
from ffn_amd.inference import seed as seed_lib
from ffn_amd.training.model import ModelInfo
from tests.emulated_device import EmulatedDeviceClient

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')

from tests.native_shim import ShimClient, ShimHandle


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
  from tests import native_shim
  return native_shim.build_shim(tmp_path_factory.mktemp('shim'))


def _request():
  r = request_lib.InferenceRequest()
  o = r.inference_options
  o.init_activation, o.pad_value, o.move_threshold = 0.95, 0.05, 0.9
  o.segment_threshold, o.min_segment_size = 0.6, 1000
  o.min_boundary_dist.x = o.min_boundary_dist.y = o.min_boundary_dist.z = 1
  return r


def _info():
  return ModelInfo(deltas=(8, 8, 8), pred_mask_size=(33, 33, 33),
                   input_seed_size=(33, 33, 33), input_image_size=(33, 33, 33))


def _canvas(shim, blob, image, native, **kwargs):
  ShimHandle.shim = shim
  r = _request()
  info = _info()
  cls = ShimClient if native else EmulatedDeviceClient
  client = cls(inference_utils.Counters(), blob, 12, (33, 33, 33), (8, 8, 8))
  counters = inference_utils.Counters()
  c = inference.make_canvas(info, client, image, r.inference_options,
                            counters=counters,
                            movement_policy_fn=movement.get_policy_fn(r, info),
                            **kwargs)
  return c


def test_struct_layout_matches_the_header(shim):
  assert ctypes.sizeof(_lib.SegmentParams) == shim.shim_sizeof_params() == 104
  assert ctypes.sizeof(_lib.SegmentResult) == shim.shim_sizeof_result() == 88


@pytest.mark.parametrize('name', ['cells56', 'cells72'])
def test_native_loop_reproduces_reference_run(shim, fib25_blob, name):
  """segment_all with every segment_at inside the C++ loop == the reference's
  own Canvas.segment_all (steps, segmentation, counters)."""
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_%s.npz' % name))
  c = _canvas(shim, fib25_blob, synthetic.normalize(g['volume']), True)
  assert c._native_loop_ok()
  c.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                              coords=g['seeds']))
  h = c._handle
  assert h.native_calls > 0
  assert np.array_equal(np.array(h.steps_seen).reshape(-1, 3), g['steps'])
  assert np.array_equal(np.asarray(c.segmentation), g['segmentation'])
  import json
  ref = json.loads(str(g['counters']))
  for key in ('update_at-calls', 'voxels-segmented', 'skip_invalid_pos',
              'skip_threshold', 'seed_got_too_weak', 'segment_at-loop-calls'):
    if key in ref:
      assert c.counters[key].value == ref[key], key


def test_hinted_positions_are_what_the_loop_pops_next(shim, fib25_blob):
  """The hint the loop gives the device before every step (SegmentLoop::
  guess_next -> HipLoopDevice::hint_next: conv0_a of the next step is queued for
  the first hinted position that is valid on the canvas once this step has
  pasted): whenever the next step is made at a hinted position it is THAT one
  (asserted step by step in tests/native_shim.py), and most steps are."""
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))
  ShimHandle.spec_stats[:] = [0, 0]
  c = _canvas(shim, fib25_blob, synthetic.normalize(g['volume']), True)
  c.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                              coords=g['seeds']))
  assert np.array_equal(np.array(c._handle.steps_seen).reshape(-1, 3), g['steps'])
  hinted, used = ShimHandle.spec_stats
  steps = len(g['steps'])
  print('%d steps, %d hints, %d steps made at a hinted position' % (
      steps, hinted, used))
  assert hinted > steps // 2
  assert used > 0.6 * hinted


def test_native_and_python_loops_leave_the_same_canvas_state(shim, fib25_blob):
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))
  image = synthetic.normalize(g['volume'])
  seed0 = (16, 16, 32)  # 13 FoV steps in the reference run
  runs = []
  for native in (False, True):
    c = _canvas(shim, fib25_blob, image, native, keep_history=True)
    assert c._native_loop_ok() == native
    n = c.segment_at(seed0)
    runs.append(dict(
        n=n, history=[tuple(int(v) for v in p) for p in c.history],
        deleted=[int(v) for v in c.history_deleted],
        mn=[int(v) for v in c._min_pos], mx=[int(v) for v in c._max_pos],
        seed=np.array(np.asarray(c.seed)), rejects=c.gate_rejects,
        counters={k: c.counters[k].value for k in (
            'update_at-calls', 'skip_threshold', 'skip_invalid_pos',
            'seed_got_too_weak', 'segment_at-loop-calls',
            'movement_policy-calls')},
        start=c._start_logit(seed0)))
  a, b = runs
  assert a['n'] == b['n'] > 7
  for key in ('history', 'deleted', 'mn', 'mx', 'rejects', 'counters', 'start'):
    assert a[key] == b[key], key
  assert np.array_equal(a['seed'], b['seed'], equal_nan=True)


@pytest.mark.parametrize('fail_code', [_lib.ERR_RANGE, _lib.ERR_FLOW])
def test_budgeted_calls_resume_to_the_same_result(shim, fib25_blob, fail_code):
  """max_steps + resume (what bench.py uses to time exactly K steps), and a
  voided step (FFN_ERR_RANGE: fp16 range; FFN_ERR_FLOW: the resident launch timed
  out) repeated on resume."""
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))
  image = synthetic.normalize(g['volume'])
  seed0 = (16, 16, 32)  # 13 FoV steps in the reference run
  full = _canvas(shim, fib25_blob, image, True, keep_history=True)
  n_full = full.segment_at(seed0)
  assert n_full > 7

  part = _canvas(shim, fib25_blob, image, True, keep_history=True)
  ShimHandle.fail_step = 4
  ShimHandle.fail_code = fail_code
  try:
    got = part._segment_at_native(seed0, max_steps=3)
    assert got == 3 and part._native_active
    while part._native_active:
      got += part._segment_at_native(seed0, max_steps=3, resume=True)
  finally:
    ShimHandle.fail_step = None
    ShimHandle.fail_code = _lib.ERR_RANGE
  assert got == n_full
  assert part._handle.steps_seen == full._handle.steps_seen
  assert part.history == full.history
  assert part.history_deleted == full.history_deleted
  assert np.array_equal(np.asarray(part.seed), np.asarray(full.seed),
                        equal_nan=True)
  for key in ('update_at-calls', 'skip_threshold', 'skip_invalid_pos',
              'segment_at-loop-calls'):
    assert part.counters[key].value == full.counters[key].value, key


def test_python_loop_kept_for_hooked_canvases(shim, fib25_blob):
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells56.npz'))
  image = synthetic.normalize(g['volume'])

  class Hooked(inference.DeviceCanvas):
    seen = 0

    def update_at(self, pos):
      Hooked.seen += 1
      return super().update_at(pos)

  ShimHandle.shim = shim
  r, info = _request(), _info()
  client = ShimClient(inference_utils.Counters(), fib25_blob, 12, (33, 33, 33),
                      (8, 8, 8))
  c = Hooked(info, client, image, r.inference_options,
             movement_policy_fn=movement.get_policy_fn(r, info))
  assert not c._native_loop_ok()
  n = c.segment_at(tuple(int(v) for v in g['seeds'][0]))
  assert n == Hooked.seen > 0 and c._handle.native_calls == 0


# -- randomized differential test: C++ loop vs Python loop on a scripted device ----------
class ScriptedClient(EmulatedDeviceClient):
  """Device whose step results are a hash of (position, visit count): face
  scores drawn from a few levels (many exact ties and sub-threshold faces),
  duplicate corner voxels, occasional positive segment ids, a start logit that
  decays.  No conv stack -- thousands of steps per second -- but every value the
  loops look at is deterministic device state."""

  in_thread = False

  def __init__(self, deltas, fov, seed):
    super().__init__(inference_utils.Counters(), None, 0, fov, deltas)
    self.rs_seed = seed
    self.trace = []

  def step(self, h, req, params):
    pos = tuple(req.pos)
    self.trace.append(pos)
    rng = np.random.RandomState(
        (hash((pos, self.rs_seed, len(self.trace) // 7)) & 0x7fffffff))
    d = [int(v) for v in self.deltas]
    res = _lib.StepResult()
    levels = np.float32([1.5, 2.2, 2.5, 2.5, 3.0, 3.0, 4.25])
    k = 0
    for axis in range(3):
      others = [a for a in range(3) if a != axis]
      for sign in (-1, 1):
        nrows, ncols = 2 * d[others[0]] + 1, 2 * d[others[1]] + 1
        mode = rng.randint(4)
        if mode == 0:  # a corner shared with other faces
          fi, fj = rng.choice([0, nrows - 1]), rng.choice([0, ncols - 1])
        else:
          fi, fj = rng.randint(nrows), rng.randint(ncols)
        score = levels[rng.randint(len(levels))]
        res.face_score[k] = score
        res.face_index[k] = int(fi * ncols + fj)
        rel = [0, 0, 0]
        rel[axis] = sign * d[axis]
        rel[others[0]] = int(fi) - d[others[0]]
        rel[others[1]] = int(fj) - d[others[1]]
        coord = tuple(int(p + r) for p, r in zip(pos, rel))
        if all(0 <= c < s for c, s in zip(coord, h.shape)):
          h.seed[coord] = score  # what the paste would have written
          res.face_seg[k] = int(h.seg[coord])
        k += 1
    start = tuple(req.start_pos)
    h.seed[start] = np.float32(h.seed[start] - rng.choice([0.0, 0.0, 0.02]))
    res.start_logit = h.seed[start]
    res.num_deleted = int(rng.randint(50))
    for j in range(req.num_candidates):
      cpos = tuple(req.candidates[j])
      res.cand_seed[j] = h.seed[cpos]
      res.cand_seg[j] = h.seg[cpos]
    return res


class ScriptedShimClient(ScriptedClient):
  in_thread = True

  def create_canvas(self, image):
    ShimHandle.client = self
    return ShimHandle(image)


@pytest.mark.parametrize('deltas,fov', [((8, 8, 8), (33, 33, 33)),
                                        ((2, 5, 3), (9, 21, 13)),
                                        ((0, 4, 4), (1, 17, 17))])
def test_randomized_loops_agree_on_scripted_device(shim, deltas, fov):
  ShimHandle.shim = shim
  shape = (48, 72, 64) if deltas[0] else (1, 72, 64)
  rng = np.random.RandomState(5)
  info = ModelInfo(deltas=deltas[::-1], pred_mask_size=fov[::-1],
                   input_seed_size=fov[::-1], input_image_size=fov[::-1])
  r = _request()
  r.inference_options.disco_seed_threshold = 0.002
  total_steps = 0
  stats = {}
  for trial in range(6):
    runs = []
    seg0 = (rng.random_sample(shape) < 0.03).astype(np.int32) * 7
    starts = [tuple(int(rng.randint(m, s - m)) if s - 2 * m > 0 else 0
                    for m, s in zip([f // 2 for f in fov], shape))
              for _ in range(4)]
    for cls in (ScriptedClient, ScriptedShimClient):
      client = cls(deltas, fov, seed=trial)
      c = inference.make_canvas(
          info, client, np.zeros(shape, np.float32), r.inference_options,
          counters=inference_utils.Counters(), keep_history=True,
          movement_policy_fn=movement.get_policy_fn(r, info))
      c._handle.seg[...] = seg0
      assert c._native_loop_ok() == (cls is ScriptedShimClient)
      out = []
      for start in starts:
        c._handle.seg[start] = 0
        n = c.segment_at(start)
        out.append((n, list(c.history), list(c.history_deleted),
                    c._min_pos.tolist(), c._max_pos.tolist(),
                    c._start_logit(start)))
      runs.append(dict(
          out=out, trace=list(client.trace), rejects=c.gate_rejects,
          seed=np.array(c._handle.seed),
          counters={k: c.counters[k].value for k in (
              'update_at-calls', 'skip_threshold', 'skip_invalid_pos',
              'seed_got_too_weak', 'segment_at-loop-calls')}))
    a, b = runs
    assert a['trace'] == b['trace']
    assert a['out'] == b['out']
    assert a['counters'] == b['counters'] and a['rejects'] == b['rejects']
    assert np.array_equal(a['seed'], b['seed'], equal_nan=True)
    total_steps += len(a['trace'])
    stats = {k: stats.get(k, 0) + v for k, v in a['counters'].items()}
    stats['rejects'] = stats.get('rejects', 0) + a['rejects']
  print('\nscripted device %r: %d steps, %r' % (deltas, total_steps, stats))
  assert total_steps > 200
  assert stats['skip_threshold'] and stats['skip_invalid_pos'] and stats['rejects']


def test_timed_checkpoints_with_the_native_loop(shim, fib25_blob, tmp_path):
  """checkpoint_interval > 0 keeps the native loop; checkpoints are written at
  segment boundaries and restore to the same result."""
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))
  image = synthetic.normalize(g['volume'])
  path = str(tmp_path / 'ck' / 'seg.cpoint')
  policy = functools.partial(seed_lib.PolicyFixed, coords=g['seeds'])

  class StopAfterTwo(Exception):
    pass

  a = _canvas(shim, fib25_blob, image, True, checkpoint_path=path,
              checkpoint_interval_sec=1e-9)
  assert a._native_loop_ok()
  saves = []
  orig = a.save_checkpoint

  def counting(p, partial_segment_iters=0):
    saves.append(partial_segment_iters)
    orig(p, partial_segment_iters=partial_segment_iters)
    if len(a.origins) >= 2:
      raise StopAfterTwo()

  a.save_checkpoint = counting
  with pytest.raises(StopAfterTwo):
    a.segment_all(seed_policy=policy)
  assert a._handle.native_calls > 0 and saves and not any(saves)

  b = _canvas(shim, fib25_blob, image, True)
  assert b.restore_checkpoint(path) == 0
  b.segment_all(seed_policy=policy)
  assert np.array_equal(np.asarray(b.segmentation), g['segmentation'])


def _many_canvases(shim, blob, names, **kwargs):
  """Emulated canvases of ONE client + the engine that advances them together."""
  from tests import native_shim
  ShimHandle.shim = shim
  r = _request()
  info = _info()
  client = ShimClient(inference_utils.Counters(), blob, 12, (33, 33, 33), (8, 8, 8))
  gold = {n: np.load(os.path.join(GOLDEN, 'ref_canvas_%s.npz' % n))
          for n in set(names)}
  canvases = []
  for n in names:
    c = inference.make_canvas(info, client, synthetic.normalize(gold[n]['volume']),
                              r.inference_options,
                              counters=inference_utils.Counters(),
                              movement_policy_fn=movement.get_policy_fn(r, info),
                              **kwargs)
    canvases.append(c)
  return client, native_shim.ShimEngine(client, 4), canvases, gold


@pytest.mark.parametrize('fail_round,carry,fail_code', [
    (None, True, _lib.ERR_RANGE), (37, True, _lib.ERR_FLOW),
    ('short', False, _lib.ERR_RANGE), ('short', 'deferred', _lib.ERR_FLOW)])
def test_segment_many_reproduces_reference_runs(shim, fib25_blob, fail_round, carry,
                                                fail_code):
  """ffn_host::segment_many (ffn_canvas_segment_many's loop) under the
  MultiCanvasDriver: five canvases, at most four per engine call, whole segments
  inside the C++ loop, Python only between segments.  Every canvas repeats the
  reference's own run step for step whatever it shared its calls with -- also
  when a batched round is voided once (FFN_ERR_RANGE) and the call resumed;
  'short' voids the round in which a canvas' loop has just ENDED (that canvas
  keeps its result and is not resumed: resuming it would be FFN_ERR_STATE).
  carry: the running loops' next step stays in flight when a loop ends
  (ffn_canvas_segment_many_carry), 'deferred': and a voided one shows only when
  the next call waits for it."""
  import json
  names = ['cells72', 'cells56', 'cells72', 'cells56', 'cells56']
  client, engine, canvases, gold = _many_canvases(shim, fib25_blob, names)
  engine.fail_round = fail_round
  engine.fail_code = fail_code
  engine.defer_error = carry == 'deferred'
  drv = inference.MultiCanvasDriver(engine, batch_size=4, native=True,
                                    carry=bool(carry))
  assert drv.native
  done = []
  drv.run(((c, functools.partial(seed_lib.PolicyFixed, coords=gold[n]['seeds']))
           for c, n in zip(canvases, names)), on_done=done.append)
  assert len(done) == len(names)
  total = 0
  for c, n in zip(canvases, names):
    g = gold[n]
    assert np.array_equal(np.array(c._handle.steps_seen).reshape(-1, 3), g['steps']), n
    assert np.array_equal(np.asarray(c.segmentation), g['segmentation']), n
    ref = json.loads(str(g['counters']))
    for key in ('update_at-calls', 'voxels-segmented', 'skip_invalid_pos',
                'skip_threshold', 'seed_got_too_weak', 'segment_at-loop-calls'):
      if key in ref:
        assert c.counters[key].value == ref[key], (n, key)
    total += len(g['steps'])
  assert drv.steps == total
  # the steps really shared engine rounds, and Python was entered per SEGMENT
  assert max(engine.batch_sizes) == 4 and engine.rounds < total
  assert engine.many_calls < total / 3
  voided = 0 if fail_round is None else 1
  assert engine.range_fallbacks == (voided if fail_code == _lib.ERR_RANGE else 0)
  assert engine.flow_fallbacks == (voided if fail_code == _lib.ERR_FLOW else 0)
  # steps were left in flight across returns, and none is left at the end
  assert (engine.carried > 0) == bool(carry)
  assert not shim.shim_carry_active()


def test_segment_many_step_budget_per_canvas(shim, fib25_blob):
  """max_steps_per_canvas under the native driver: every canvas is dropped after
  exactly its budget, and what it did until then is the start of its full run."""
  names = ['cells72', 'cells56', 'cells72']
  client, engine, canvases, gold = _many_canvases(shim, fib25_blob, names)
  drv = inference.MultiCanvasDriver(engine, batch_size=4, native=True,
                                    max_steps_per_canvas=20)
  drv.run((c, functools.partial(seed_lib.PolicyFixed, coords=gold[n]['seeds']))
          for c, n in zip(canvases, names))
  for c, n in zip(canvases, names):
    want = gold[n]['steps'][:20]
    got = np.array(c._handle.steps_seen).reshape(-1, 3)
    assert np.array_equal(got, want), n
  assert drv.steps == sum(min(20, len(gold[n]['steps'])) for n in names)


def test_segment_many_small_batches_and_argument_checks(shim, fib25_blob):
  """Two canvases per engine call for four jobs (a finished canvas' slot goes to
  the next job); canvases with different step parameters never share a call."""
  names = ['cells56', 'cells72', 'cells56', 'cells72']
  client, engine, canvases, gold = _many_canvases(shim, fib25_blob, names)
  engine.max_batch = 2
  drv = inference.MultiCanvasDriver(engine, batch_size=2, native=True, carry=False)
  drv.run((c, functools.partial(seed_lib.PolicyFixed, coords=gold[n]['seeds']))
          for c, n in zip(canvases, names))
  for c, n in zip(canvases, names):
    assert np.array_equal(np.asarray(c.segmentation), gold[n]['segmentation']), n
  assert max(engine.batch_sizes) == 2
  # one engine call = one set of step parameters (FFN_ERR_ARG otherwise)
  from tests import native_shim
  client2, engine2, cs, _ = _many_canvases(shim, fib25_blob, ['cells56', 'cells56'])
  sp = [c._segment_params() for c in cs]
  a = _lib.SegmentParams.from_buffer_copy(sp[0])
  b = _lib.SegmentParams.from_buffer_copy(sp[1])
  b.step.move_threshold = a.step.move_threshold + 1.0
  n = 2
  sarr = (ctypes.c_int32 * 3 * n)()
  parr = (_lib.SegmentParams * n)(a, b)
  rarr = (ctypes.c_int32 * n)(0, 0)
  res = (_lib.SegmentResult * n)()
  fin = (ctypes.c_int32 * n)()
  rc = engine2._segment_many_once([c._handle for c in cs], sarr, parr, rarr, res, fin)
  assert rc == -1  # FFN_ERR_ARG


def test_native_driver_steps_python_loop_canvases_next_to_native_ones(shim, fib25_blob):
  """A canvas whose loop has to stay in Python (here: a MovementRestrictor with
  masks) yields single FoV steps; the native driver makes them one by one and
  runs the other canvases' segments in the library -- every canvas ends with
  its reference-minted run."""
  from tests import native_shim
  from tests import test_masks
  from ffn_amd.training import model as ffn_model
  gm = np.load(test_masks.FIX)
  names = ['cells56', 'cells72']
  client, engine, canvases, gold = _many_canvases(shim, fib25_blob, names)
  r = test_masks._options()
  info = ffn_model.ModelInfo(np.array([8, 8, 8]), np.array([33, 33, 33]),
                             np.array([33, 33, 33]), np.array([33, 33, 33]))
  masked = inference.make_canvas(
      info, client, synthetic.normalize(gm['run_volume']), r.inference_options,
      counters=inference_utils.Counters(), restrictor=test_masks._restrictor(gm),
      movement_policy_fn=movement.get_policy_fn(r, info))
  assert not masked._native_loop_ok()
  jobs = [(canvases[0], functools.partial(seed_lib.PolicyFixed,
                                          coords=gold['cells56']['seeds'])),
          (masked, functools.partial(seed_lib.PolicyFixed, coords=gm['run_seeds'])),
          (canvases[1], functools.partial(seed_lib.PolicyFixed,
                                          coords=gold['cells72']['seeds']))]
  drv = inference.MultiCanvasDriver(engine, batch_size=4, native=True)
  drv.run(jobs)
  for c, n in zip(canvases, names):
    assert np.array_equal(np.asarray(c.segmentation), gold[n]['segmentation']), n
  assert np.array_equal(np.asarray(masked.segmentation), gm['run_segmentation'])
  import json
  ref = json.loads(str(gm['run_counters']))
  for key in ('update_at-calls', 'skip_restriced_pos', 'voxels-segmented'):
    assert masked.counters[key].value == ref[key], key
  assert drv.steps == (len(gm['run_steps']) + len(gold['cells56']['steps']) +
                       len(gold['cells72']['steps']))


def test_segment_many_two_groups_in_two_threads(shim, fib25_blob):
  """groups=2: the open canvases form two groups, each advanced by its own host
  thread with its own library calls (on the GPU: one group's steps run while
  the other's ended segments are committed and re-seeded).  Every canvas still
  repeats its reference-minted run; every job is done exactly once."""
  import json
  names = ['cells72', 'cells56', 'cells56', 'cells72', 'cells56', 'cells56',
           'cells72']
  client, engine, canvases, gold = _many_canvases(shim, fib25_blob, names)
  engine.max_batch = 2
  drv = inference.MultiCanvasDriver(engine, batch_size=2, native=True, groups=2)
  assert drv.groups == 2
  done = []
  drv.run(((c, functools.partial(seed_lib.PolicyFixed, coords=gold[n]['seeds']))
           for c, n in zip(canvases, names)), on_done=done.append)
  assert sorted(id(c) for c in done) == sorted(id(c) for c in canvases)
  total = 0
  for c, n in zip(canvases, names):
    g = gold[n]
    assert np.array_equal(np.array(c._handle.steps_seen).reshape(-1, 3), g['steps']), n
    assert np.array_equal(np.asarray(c.segmentation), g['segmentation']), n
    ref = json.loads(str(g['counters']))
    for key in ('update_at-calls', 'voxels-segmented', 'segment_at-loop-calls'):
      if key in ref:
        assert c.counters[key].value == ref[key], (n, key)
    total += len(g['steps'])
  assert drv.steps == total
  assert max(engine.batch_sizes) <= 2


@pytest.mark.parametrize('name,native', [('nodisco', True), ('mbd3', False),
                                         ('seg08_probmap', True)])
def test_non_default_inference_options(shim, fib25_blob, name, native):
  """InferenceOptions away from the sample configuration (disco bias off /
  needing a fraction of active voxels, min_boundary_dist > 1, other segment
  thresholds and size filters, quantised probability maps kept; reference
  inference.py:416-439,556,624-660): the reference's own runs reproduced by the
  device-canvas host logic over the emulated device, with the Python loop or
  with the library's segment loop (all seven cases, both loops: the GPU suite,
  tests/test_gpu_round5.py)."""
  from tests import option_cases
  g = option_cases.load(name)
  ShimHandle.shim = shim
  r = option_cases.request_for(g)
  info = _info()
  cls = ShimClient if native else EmulatedDeviceClient
  client = cls(inference_utils.Counters(), fib25_blob, 12, (33, 33, 33), (8, 8, 8))
  c = inference.make_canvas(info, client, g['image'], r.inference_options,
                            counters=inference_utils.Counters(),
                            movement_policy_fn=movement.get_policy_fn(r, info),
                            keep_probability_maps=g['probmap'])
  assert c._native_loop_ok() == native
  option_cases.run(c, g)
  option_cases.check(c, g, steps=c._handle.steps_seen if native else None)
