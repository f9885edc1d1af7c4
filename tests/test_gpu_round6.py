"""GPU tests added in round 6: the paced resident stack (timing only: same bits), the
80-voxel / two-chains family (conv_variant 10) against the oracle and against its own
per-layer launches, the CAUSE of a void step travelling with the step, the turn-around
figures.  All through the C-ABI.
"""

import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from tests.test_gpu_round2 import (_assert_shipped_default, _device_canvas, hip_exe,  # noqa: F401
                                   seed_lib_fixed)

pytestmark = pytest.mark.gpu

FAULT = 2048  # flow_debug: main chunk 3 stops publishing after the first conv


def _engine(depth=12, fib25=True, seed=5):
  from ffn_amd import engine as hip_engine
  from ffn_amd.training.models import convstack_3d
  from oracle import ffn_oracle
  m = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8], depth=depth)
  if fib25:
    with np.load(os.path.join(GOLDEN, 'fib25_weights.npz')) as d:
      variables = {k: d[k] for k in d.files}
  else:
    variables = ffn_oracle.random_weights(depth, seed=seed, stddev=0.05)
  m.set_variables(variables)
  return hip_engine.HipEngine.from_model(m, max_batch=1), ffn_oracle.weights_blob(variables, depth)


def _inputs(seed=3):
  rng = np.random.RandomState(seed)
  return (rng.normal(0, 1.5, [1, 33, 33, 33]).astype(np.float32),
          rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32))


def test_pacing_is_timing_only():
  """The beat of the resident stack (ConvStackTab::pace) is measured when the weights are
  set and changes WHEN a conv starts, never what it computes (convstack_3d.py:38-54)."""
  eng, _ = _engine()
  seed, img = _inputs()
  beat = eng.get_option('flow_pace_now')
  assert eng.get_option('flow_pace') == -1
  assert beat == 0 or 560 <= beat <= 820, beat
  free_us = eng.get_option('flow_pace_free_ns') / 1e3
  best_us = eng.get_option('flow_pace_best_ns') / 1e3
  assert 100 < best_us < 400 and 100 < free_us < 400, (best_us, free_us)
  if beat:  # kept only where it beat the free-running stack twice in a row
    assert best_us < 0.985 * free_us
  paced = eng.predict(seed, img)
  for pace, spread in ((0, -1), (700, -1), (700, 0), (640, 320)):
    eng.set_option('flow_pace', pace)
    eng.set_option('flow_pace_spread', spread)
    assert eng.get_option('flow_pace_now') == pace
    assert np.array_equal(eng.predict(seed, img), paced), (pace, spread)
  assert eng.get_option('stat_flow_timeouts') == 0
  with pytest.raises(Exception):
    eng.set_option('flow_pace', 6000)
  eng.close()


@pytest.mark.parametrize('depth,fib25', [(12, True), (2, False), (5, False)])
def test_conv_variant_10_matches_the_oracle_and_its_own_per_layer_launches(depth, fib25):
  """conv32hs / conv32h (80-voxel workgroups, 16x16x32 tiles, a fifth tile split over the
  waves by tap): the resident stack = its per-layer launches bit for bit, both within
  1e-4 of the C oracle (convstack_3d.py:26-56, 86-95)."""
  from oracle import ffn_oracle
  eng, blob = _engine(depth, fib25)
  seed, img = _inputs(depth)
  want = ffn_oracle.forward(img[0], seed[0], blob, depth)
  v9 = eng.predict(seed, img)
  eng.set_option('conv_variant', 10)
  assert eng.get_option('conv_variant') == 10
  eng.set_option('flow', 2)
  resident = eng.predict(seed, img)
  eng.set_option('flow', 0)
  per_layer = eng.predict(seed, img)
  assert np.array_equal(resident, per_layer)
  assert np.abs(resident[0] - want).max() <= 1e-4
  assert np.abs(resident - v9).max() <= 2e-5
  # paced: the same bits again
  eng.set_option('flow', 2)
  eng.set_option('flow_pace', 760)
  assert np.array_equal(eng.predict(seed, img), resident)
  assert eng.get_option('stat_flow_timeouts') == 0
  eng.close()


def test_conv_variant_10_on_a_canvas(hip_exe, fib25_model):  # noqa: F811
  """The reference-minted cells72 run (ref_canvas_cells72.npz, 94 FoV steps) under
  conv_variant 10: same positions, same segmentation (inference.py:460-683)."""
  from ffn_amd import synthetic
  from tests.test_gpu_round2 import _run_recorded
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))
  eng = hip_exe.engine
  try:
    eng.set_option('conv_variant', 10)
    canvas = _device_canvas(hip_exe, fib25_model, synthetic.normalize(g['volume']))
    got_steps, _ = _run_recorded(canvas, g['seeds'].astype(np.int32))
    want_steps = [tuple(int(v) for v in p) for p in g['steps']]
    assert got_steps == want_steps
    assert np.array_equal(np.asarray(canvas.segmentation), g['segmentation'])
    canvas.close()
  finally:
    eng.restore_default_variant()
    eng.set_option('flow', 2)
  _assert_shipped_default(eng)


def test_void_cause_travels_with_the_step(hip_exe, fib25_model):  # noqa: F811
  """with two steps in flight a global time-out counter cannot say WHICH step
  timed out.  The step's own record does (ffn_step_result.range_error 3): a canvas step
  under the injected fault comes back FFN_ERR_FLOW, is repeated per layer with the same
  result as a healthy engine's, and a step that is fine reports nothing."""
  from ffn_amd import _lib
  from ffn_amd import synthetic
  from oracle import ffn_oracle
  eng = hip_exe.engine
  vol = synthetic.normalize(synthetic.cells_volume((48, 48, 48), seed=3))

  def one_step(fault):
    canvas = eng.create_canvas(vol)
    pos = (24, 24, 24)
    canvas.init_seed(pos, ffn_oracle.f32_logit(0.95))
    req = _lib.StepRequest()
    req.pos[:] = pos
    req.start_pos[:] = pos
    req.num_candidates = 0
    params = _lib.StepParams(ffn_oracle.f32_logit(0.05), ffn_oracle.f32_logit(0.9), 0.0)
    eng.set_option('flow_debug', FAULT if fault else 0)
    res = eng.step1(canvas, req, params)
    eng.set_option('flow_debug', 0)
    seed = np.array(canvas.read_seed())
    canvas.close()
    return res, seed

  try:
    eng.set_option('flow', 2)
    voids0 = eng.get_option('stat_flow_voids')
    good, seed_good = one_step(False)
    assert good.range_error == 0 and eng.get_option('stat_flow_voids') == voids0
    bad, seed_bad = one_step(True)  # (step1 repeats a step that came back FFN_ERR_FLOW)
    assert eng.get_option('stat_flow_voids') == voids0 + 1
    assert eng.get_option('conv_variant') == 9  # a time-out is NOT a range error
    assert bad.range_error == 0
    assert list(bad.face_index) == list(good.face_index)
    assert np.array_equal(np.isnan(seed_bad), np.isnan(seed_good))
    assert np.array_equal(seed_bad[~np.isnan(seed_bad)], seed_good[~np.isnan(seed_good)])
  finally:
    eng.set_option('flow_debug', 0)
    eng.restore_default_variant()
    eng.set_option('flow', 2)
  _assert_shipped_default(eng)


def test_turn_around_figures(hip_exe, fib25_model):  # noqa: F811
  """stat_turn_*: the GPU's and the host's view of the time between two single-FoV steps
  inside a segment (bench.py: turn_around_us) -- present, ordered, microseconds."""
  from ffn_amd import synthetic
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))  # 94 FoV steps
  eng = hip_exe.engine
  eng.set_option('stat_reset', 0)
  import functools
  canvas = _device_canvas(hip_exe, fib25_model, synthetic.normalize(g['volume']))
  # (the library's own segment loop: a Python loop between the steps takes longer than
  # the 100 us these figures are clamped to)
  canvas.segment_all(seed_policy=functools.partial(seed_lib_fixed(),
                                                   coords=g['seeds'].astype(np.int32)))
  assert np.array_equal(np.asarray(canvas.segmentation), g['segmentation'])
  canvas.close()
  n = eng.get_option('stat_turn_count')
  gpu_ns = eng.get_option('stat_turn_gpu_ns')
  host_ns = eng.get_option('stat_turn_host_ns')
  launch_ns = eng.get_option('stat_launch_host_ns')
  assert n > 40, n
  assert 1000 < launch_ns <= host_ns < 100000, (launch_ns, host_ns)
  assert 2000 < gpu_ns < 100000, gpu_ns
  _assert_shipped_default(eng)


def test_stack_queued_ahead_bookkeeping(hip_exe, fib25_model):  # noqa: F811
  """stack_ahead (the default): every step that runs on a conv0_a made ahead also finds its
  resident stack queued behind it; a stack that no step used either ended after its first
  conv (its conv0_a found no valid position) or is one of the few the loop stepped past; a
  host that idles 20 us in front of every step's launches changes nothing -- and the run is
  the reference-minted one either way."""
  import functools
  from ffn_amd import synthetic
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))  # 94 FoV steps
  eng = hip_exe.engine
  _assert_shipped_default(eng)
  try:
    for delay in (0, 20000):
      eng.set_option('debug_submit_delay_ns', delay)
      eng.set_option('stat_reset', 0)
      canvas = _device_canvas(hip_exe, fib25_model, synthetic.normalize(g['volume']))
      canvas.segment_all(seed_policy=functools.partial(seed_lib_fixed(),
                                                       coords=g['seeds'].astype(np.int32)))
      assert np.array_equal(np.asarray(canvas.segmentation), g['segmentation'])
      assert canvas.counters['update_at-calls'].value == len(g['steps'])
      canvas.close()
      launched = eng.get_option('stat_spec_launched')
      hits = eng.get_option('stat_spec_hits')
      used = eng.get_option('stat_ahead_used')
      wasted = eng.get_option('stat_ahead_wasted')
      aborted = eng.get_option('stat_ahead_aborted')
      print('delay %d ns: conv0_a launched ahead %d, used %d; stacks queued ahead used %d, not '
            'used %d (ended after the first conv: %d)' % (delay, launched, hits, used, wasted,
                                                          aborted))
      assert used == hits > 0.6 * len(g['steps'])
      assert used + wasted == launched  # every launch made ahead had its stack behind it
      assert 0 <= aborted <= wasted
      assert eng.get_option('stat_spec_mismatch') == 0
  finally:
    eng.set_option('debug_submit_delay_ns', 0)
