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
    target_path = get_target_path(request, point_num)
    if target_path is None:
      return
    curr = request.points[point_num]
    point = curr.point.z, curr.point.y, curr.point.x
    radius = (request.radius.z, request.radius.y, request.radius.x)
    canvas, alignment = get_canvas(point, radius, runner)
    if canvas is None:
      logging.warning('Could not get a canvas object.')
      return
    if seeder is None:
      from .. import seeding  # pylint:disable=g-import-not-at-top
      handle = getattr(canvas, '_handle', None)
      engine = getattr(handle, 'engine', None)
      seeder = seeding.default_seeder(getattr(engine, 'device_id', 0))

    def unalign_prob(prob):
      return alignment.align_and_crop(canvas.corner_zyx, prob,
                                      alignment.corner, alignment.size,
                                      forward=False)

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
      canvas._deregister_client()  # pylint:disable=protected-access
      return

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

    transformed_point = alignment.transform(np.array([point]).T)
    tz, ty, tx = transformed_point[:, 0]
    oz, oy, ox = canvas.corner_zyx
    tz, ty, tx = int(tz - oz), int(ty - oy), int(tx - ox)

    # First index enumerates the original segments, second (where present)
    # the segmentation attempts.
    raw_probs, probs, deletes, histories = [], [], [], []
    start_points = [[], []]
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
        canvas.log_info('.. starting segmentation at (xyz): %d %d %d',
                        x0, y0, z0)
        canvas.segment_at((int(z0), int(y0), int(x0)))
        seg_prob = expit(np.asarray(canvas.seed))
        start_points[i].append((x0, y0, z0))

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
        raw_probs.append(qprob)
        probs.append(unalign_prob(qprob))
        deletes.append(np.array(canvas.history_deleted))
        histories.append(np.array(canvas.history))

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

  canvas.log_info('saving results to %s', target_path)
  with storage.atomic_file(target_path) as fd:
    np.savez_compressed(fd,
                        probs=np.array(probs),
                        raw_probs=np.array(raw_probs),
                        deletes=_ragged(deletes),
                        histories=_ragged(histories),
                        start_points=_ragged(start_points),
                        request=request.SerializeToString(),
                        counters=canvas.counters.dumps(),
                        corner_zyx=canvas.corner_zyx,
                        is_shift=is_shift)
  canvas.log_info('.. save complete')
  canvas._deregister_client()  # pylint:disable=protected-access
  if hasattr(canvas, 'close'):
    canvas.close()


def process(request, runner, voxel_size=(1, 1, 1)):
  num_points = len(request.points)
  for i in range(num_points):
    logging.info('processing %d/%d', i, num_points)
    process_point(request, runner, i, voxel_size)
