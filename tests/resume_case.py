"""Mid-segment checkpoint / resume scenario shared by the CPU test (emulated
device) and the GPU test (test tooling).

Reference behaviour being pinned: Canvas.save_checkpoint / restore_checkpoint
(ffn/inference/inference.py:728-843) and Runner.run's resume
(ffn/inference/runner.py:505-519): a run that is interrupted in the middle of
a segment and continued from its `.cpoint` in a FRESH canvas must end in
exactly the state of the uninterrupted run.
"""

import functools
import os

import numpy as np


class _Interrupted(Exception):
  pass


def run_resume_case(make_canvas, seeds, stop_after_steps, tmp_path,
                    run_uninterrupted=True):
  """make_canvas(checkpoint_path, interval) -> a fresh canvas on the same
  volume.  Returns (uninterrupted or None, resumed, info)."""
  from ffn_amd.inference import seed as seed_lib
  policy = functools.partial(seed_lib.PolicyFixed, coords=np.asarray(seeds))

  # A: uninterrupted
  a = None
  if run_uninterrupted:
    a = make_canvas(None, 0)
    a.segment_all(seed_policy=policy)

  # B1: checkpoint after every FoV step, killed mid-segment
  path = os.path.join(str(tmp_path), 'ck', 'seg-0_0_0.cpoint')
  b1 = make_canvas(path, 1e-9)
  steps = [0]
  inner = b1.update_at

  def hooked(pos):  # instance hook: the Python loop runs, one checkpoint per step
    if steps[0] >= stop_after_steps:
      raise _Interrupted()
    steps[0] += 1
    return inner(pos)

  b1.update_at = hooked
  try:
    b1.segment_all(seed_policy=policy)
    raise AssertionError('the run ended before step %d' % stop_after_steps)
  except _Interrupted:
    pass
  assert os.path.exists(path)
  if hasattr(b1, 'close'):
    b1.close()  # the device canvas is gone; only the file survives
  del b1

  # B2: a fresh canvas continues from the file
  b2 = make_canvas(None, 0)
  partial = b2.restore_checkpoint(path)
  b2.segment_all(seed_policy=policy, partial_segment_iters=partial)
  return a, b2, {'partial_segment_iters': partial, 'checkpoint': path}


def assert_same_final_state(a, b):
  assert np.array_equal(np.asarray(a.segmentation), np.asarray(b.segmentation))
  assert np.array_equal(np.asarray(a.seed), np.asarray(b.seed), equal_nan=True)
  assert a._max_id == b._max_id
  assert ({k: (tuple(v.start_zyx), v.iters) for k, v in a.origins.items()} ==
          {k: (tuple(v.start_zyx), v.iters) for k, v in b.origins.items()})
  assert set(a.overlaps) == set(b.overlaps)
  for k in a.overlaps:
    assert np.array_equal(a.overlaps[k], b.overlaps[k])
  # ('voxels-segmented' is re-derived as sum(seg != 0) on restore, -1 markers
  # included, exactly as the reference does: inference.py:752-753)
  assert (a.counters['update_at-calls'].value ==
          b.counters['update_at-calls'].value)
