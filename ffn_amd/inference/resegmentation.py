"""Resegmentation of object pairs / endpoints around decision points.

Entry points and output format of reference ffn/inference/resegmentation.py
(`get_starting_location` :38-46, `get_target_path` :49-80, `get_canvas`
:83-108, `process_point` :111-293, `process` :296-300): for every decision
point a small canvas is cut out of the volume, the one or two objects in
question are cleared from the initial segmentation, and each is re-grown by
the FFN from the point deepest inside it (maximum of the Euclidean distance
transform), with retries; the object probability maps and the FoV histories
are saved for resegmentation_analysis.

The FoV loop is the ordinary device-resident canvas (`Canvas.segment_at` with
`keep_history`); the distance transform runs on the GPU
(`ffn_amd.seeding.Seeder.edt`, exact).  Three things of the reference do not
run at its HEAD and are fixed here: the canvas is built with
`keep_probability_maps=True` (the reference assigns into `canvas.seg_prob`,
which is otherwise None), `process` passes `voxel_size` on to `process_point`,
and the ragged per-object result lists are saved as object arrays (what numpy
< 1.24 made of them implicitly).
"""

from __future__ import annotations

import hashlib
import logging
import os

import numpy as np
from scipy.special import expit

from . import storage
from .inference_utils import timer_counter


def get_starting_location(dists, exclusion_radius):
  """Position of the maximum of `dists` (first in C order); an area of
  `exclusion_radius` around it is zeroed so that a retry picks another one."""
  z, y, x = np.unravel_index(np.argmax(dists), tuple(dists.shape))
  er = exclusion_radius
  dists[max(z - er.z, 0):z + er.z + 1,
        max(y - er.y, 0):y + er.y + 1,
        max(x - er.x, 0):x + er.x + 1] = 0
  return z, y, x


def get_target_path(request, point_num):
  """Output path for a point, or None if the output already exists."""
  output_dir = request.output_directory
  id_a = request.points[point_num].id_a
  id_b = request.points[point_num].id_b
  if request.subdir_digits > 1:
    m = hashlib.md5()
    m.update(str(id_a).encode())
    m.update(str(id_b).encode())
    output_dir = os.path.join(output_dir, m.hexdigest()[:request.subdir_digits])
  os.makedirs(output_dir, exist_ok=True)
  dp = request.points[point_num].point
  target_path = os.path.join(output_dir, '%d-%d_at_%d_%d_%d.npz' % (
      id_a, id_b, dp.x, dp.y, dp.z))
  if os.path.exists(target_path):
    logging.info('Output already exists: %s', target_path)
    return None
  return target_path


def get_canvas(point, radius, runner):
  """Canvas of size 2 * radius + 1 centred on `point` (z, y, x), or
  (None, None) if the volume does not give that much context."""
  origin = np.array(point)
  radius = np.array(radius)
  corner = origin - radius
  subvol_size = radius * 2 + 1
  end = subvol_size + corner
  shape = runner.init_seg_volume.shape
  if (np.any(corner < 0) or shape[1] <= end[0] or shape[2] <= end[1] or
      shape[3] <= end[2]):
    logging.error('Not enough context for: %d, %d, %d; corner: %r; end: %r',
                  point[2], point[1], point[0], corner, end)
    return None, None
  return runner.make_canvas(corner, subvol_size, keep_history=True,
                            keep_probability_maps=True)


def _ragged(items):
  """Object array with one entry per item (np.array of numpy < 1.24)."""
  try:
    return np.array(items)
  except ValueError:
    out = np.empty(len(items), dtype=object)
    for k, v in enumerate(items):
      out[k] = v
    return out


class _PointTask:
  """One decision point between `_prepare_point` and `_save_point`."""

  __slots__ = ('request', 'point_num', 'target_path', 'canvas', 'alignment',
               'point', 'radius', 'todo', 'is_shift', 'raw_probs', 'probs',
               'deletes', 'histories', 'start_points')


def _default_seeder(canvas):
  from .. import seeding  # pylint:disable=g-import-not-at-top
  handle = getattr(canvas, '_handle', None)
  engine = getattr(handle, 'engine', None)
  return seeding.default_seeder(getattr(engine, 'device_id', 0))


def _prepare_point(request, runner, point_num):
  """Target path, canvas and the cleared initial segmentation for a point
  (resegmentation.py:118-175).  None if there is nothing to do."""
  target_path = get_target_path(request, point_num)
  if target_path is None:
    return None
  curr = request.points[point_num]
  point = curr.point.z, curr.point.y, curr.point.x
  radius = (request.radius.z, request.radius.y, request.radius.x)
  canvas, alignment = get_canvas(point, radius, runner)
  if canvas is None:
    logging.warning('Could not get a canvas object.')
    return None

  is_shift = (canvas.restrictor is not None and
              np.any(getattr(canvas.restrictor, 'shift_mask', None)))
  is_endpoint = not curr.HasField('id_b')

  segmentation = np.array(np.asarray(canvas.segmentation))
  seg_a = segmentation == canvas.local_id(curr.id_a)
  size_a = np.sum(seg_a)
  if is_endpoint:
    size_b = -1
    todo = [seg_a]
  else:
    seg_b = segmentation == canvas.local_id(curr.id_b)
    size_b = np.sum(seg_b)
    todo = [seg_a, seg_b]
  if size_a == 0 or size_b == 0:
    logging.warning('Segments (%d, %d) local ids (%d, %d) not found in input '
                    'at %r.  Current values are: %r.', curr.id_a, curr.id_b,
                    canvas.local_id(curr.id_a), canvas.local_id(curr.id_b),
                    point, np.unique(segmentation))
    _release(canvas)
    return None

  if is_endpoint:
    canvas.seg_prob[:] = 0
    segmentation[:] = 0
  else:
    # Clear the two segments in question, keep everything else as context.
    segmentation[seg_a] = 0
    segmentation[seg_b] = 0
    canvas.seg_prob[seg_a] = 0
    canvas.seg_prob[seg_b] = 0
  canvas.segmentation[...] = segmentation

  task = _PointTask()
  task.request, task.point_num, task.target_path = request, point_num, target_path
  task.canvas, task.alignment = canvas, alignment
  task.point, task.radius, task.todo, task.is_shift = point, radius, todo, is_shift
  # First index enumerates the original segments, second (where present) the
  # segmentation attempts.
  task.raw_probs, task.probs, task.deletes, task.histories = [], [], [], []
  task.start_points = [[], []]
  return task


def _release(canvas):
  canvas._deregister_client()  # pylint:disable=protected-access
  if hasattr(canvas, 'close'):
    canvas.close()


def _point_steps(task, voxel_size, seeder):
  """The seeded re-growth of a point's objects (resegmentation.py:177-279) as
  a step generator: yields the canvas' FoV-step requests, so that one thread can
  advance many points in batched engine calls (`process_many`)."""
  request, canvas, alignment = task.request, task.canvas, task.alignment
  radius, todo = task.radius, task.todo
  if seeder is None:
    seeder = _default_seeder(canvas)

  def unalign_prob(prob):
    return alignment.align_and_crop(canvas.corner_zyx, prob, alignment.corner,
                                    alignment.size, forward=False)

  transformed_point = alignment.transform(np.array([task.point]).T)
  tz, ty, tx = transformed_point[:, 0]
  oz, oy, ox = canvas.corner_zyx
  tz, ty, tx = int(tz - oz), int(ty - oy), int(tx - ox)

  if request.HasField('analysis_radius'):
    ar = request.analysis_radius
    lo = (radius[0] - ar.z, radius[1] - ar.y, radius[2] - ar.x)
    analysis = tuple(slice(l, l + 2 * r + 1)
                     for l, r in zip(lo, (ar.z, ar.y, ar.x)))
  else:
    analysis = (slice(None),) * 3

  options = request.inference.inference_options
  margin = canvas.margin
  for i, seg in enumerate(todo):
    logging.info('processing object %d', i)
    with timer_counter(canvas.counters, 'edt'):
      dists = seeder.edt(seg, voxel_size)
      # Do not seed where not enough context is available.
      dists[:margin[0], :, :] = 0
      dists[:, :margin[1], :] = 0
      dists[:, :, :margin[2]] = 0
      dists[-margin[0]:, :, :] = 0
      dists[:, -margin[1]:, :] = 0
      dists[:, :, -margin[2]:] = 0
    if request.HasField('init_exclusion_radius'):
      ier = request.init_exclusion_radius
      dists[tz - ier.z:tz + ier.z + 1, ty - ier.y:ty + ier.y + 1,
            tx - ier.x:tx + ier.x + 1] = 0

    seg_prob = None
    recovered = False
    crop_prob = None
    for _ in range(request.max_retry_iters):
      z0, y0, x0 = get_starting_location(dists, request.exclusion_radius)
      if not seg[z0, y0, x0]:
        continue
      canvas.log_info('.. starting segmentation at (xyz): %d %d %d', x0, y0, z0)
      yield from canvas._segment_at_gen(  # pylint:disable=protected-access
          (int(z0), int(y0), int(x0)))
      seg_prob = expit(np.asarray(canvas.seed))
      task.start_points[i].append((x0, y0, z0))

      # Was an acceptable fraction of the seeded segment recovered?
      recovered = True
      crop_seg = seg[analysis]
      crop_prob = seg_prob[analysis]
      start_size = np.sum(crop_seg)
      segmented_voxels = np.sum((crop_prob >= options.segment_threshold) &
                                crop_seg)
      if request.segment_recovery_fraction > 0:
        if segmented_voxels / start_size >= request.segment_recovery_fraction:
          break
      elif segmented_voxels >= options.min_segment_size:
        break
      recovered = False

    if seg_prob is not None:
      qprob = storage.quantize_probability(seg_prob)
      task.raw_probs.append(qprob)
      task.probs.append(unalign_prob(qprob))
      task.deletes.append(np.array(canvas.history_deleted))
      task.histories.append(np.array(canvas.history))

    if request.terminate_early:
      if not recovered:
        break
      if (request.segment_recovery_fraction > 0 and i == 0 and
          len(todo) > 1):
        crop_seg = todo[1][analysis]
        size2 = np.sum(crop_seg)
        segmented_voxels2 = np.sum(
            (crop_prob >= options.segment_threshold) & crop_seg)
        if segmented_voxels2 / size2 < request.segment_recovery_fraction:
          break


def _save_point(task):
  """Writes the point's .npz (resegmentation.py:281-293) and frees the canvas."""
  canvas = task.canvas
  canvas.log_info('saving results to %s', task.target_path)
  with storage.atomic_file(task.target_path) as fd:
    np.savez_compressed(fd,
                        probs=np.array(task.probs),
                        raw_probs=np.array(task.raw_probs),
                        deletes=_ragged(task.deletes),
                        histories=_ragged(task.histories),
                        start_points=_ragged(task.start_points),
                        request=task.request.SerializeToString(),
                        counters=canvas.counters.dumps(),
                        corner_zyx=canvas.corner_zyx,
                        is_shift=task.is_shift)
  canvas.log_info('.. save complete')
  _release(canvas)


def process_point(request, runner, point_num, voxel_size, seeder=None):
  """Runs resegmentation for one point of a ResegmentationRequest.

  Args:
    request: ResegmentationRequest
    runner: started inference Runner (with `init_segmentation` configured)
    point_num: index into request.points
    voxel_size: (z, y, x) voxel size in physical units
    seeder: distance-transform provider (default: the GPU `Seeder`)
  """
  with timer_counter(runner.counters, 'resegmentation'):
    task = _prepare_point(request, runner, point_num)
    if task is None:
      return
    task.canvas._drive(  # pylint:disable=protected-access
        _point_steps(task, voxel_size, seeder))
  _save_point(task)


def process(request, runner, voxel_size=(1, 1, 1)):
  num_points = len(request.points)
  for i in range(num_points):
    logging.info('processing %d/%d', i, num_points)
    process_point(request, runner, i, voxel_size)


def process_many(request, runner, voxel_size=(1, 1, 1), batch_size=None,
                 window=None, seeder=None, engine=None):
  """`process` with the points advanced CONCURRENTLY on one GPU.

  A resegmentation request is thousands of independent small canvases of a few
  FoV steps each -- at one canvas at a time every step is a batch-1 launch
  chain.  Here one thread keeps `window` points open and every round is one
  batched `ffn_canvas_step(n, ...)` over up to `batch_size` of them
  (`inference.MultiCanvasDriver`; needs a Runner started with direct=True).
  Results are the files `process` writes, point for point.

  Args:
    batch_size: FoV steps per engine call (default: the engine's max batch)
    window: points open at once (default 4 x batch_size); bounds host memory
  """
  from . import inference  # pylint:disable=g-import-not-at-top
  if engine is None:
    engine = runner.executor.engine
  batch_size = batch_size or engine.max_batch
  window = window or 4 * batch_size
  num_points = len(request.points)
  with timer_counter(runner.counters, 'resegmentation', increment=num_points):
    for first in range(0, num_points, window):
      tasks = []
      for i in range(first, min(first + window, num_points)):
        task = _prepare_point(request, runner, i)
        if task is not None:
          tasks.append(task)
      driver = inference.MultiCanvasDriver(engine, batch_size)
      driver.run([(t.canvas, _point_steps(t, voxel_size, seeder))
                  for t in tasks])
      for task in tasks:
        _save_point(task)
