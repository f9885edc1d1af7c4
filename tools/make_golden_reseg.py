#!/usr/bin/env python3
"""Mints tests/golden/ref_reseg.npz with the reference's own
resegmentation.process_point (ffn/inference/resegmentation.py:111-293).

Runs in the build container only.  The reference modules are imported through
tools/ref_shims; the TF forward is supplied by oracle/ffn_oracle.forward.  Two
things of the reference do not run at HEAD and are worked around HERE (they are
documented in DESIGN.md, and the product makes the same two fixes):
  * `process_point` assigns into `canvas.seg_prob`, which is None unless the
    canvas is built with keep_probability_maps=True; `get_canvas`
    (resegmentation.py:83-108) does not pass it.  The stand-in runner below
    builds its canvases with keep_probability_maps=True.
  * `process()` calls `process_point` without `voxel_size` (:296-300);
    `process_point` is called directly.
  * `np.array(deletes)` / `np.array(histories)` / `start_points` (:281-286) are
    ragged (one entry per object, different lengths); numpy >= 1.24 refuses to
    build them implicitly.  The reference's `np` is wrapped so that a ragged
    `np.array` yields the object array older numpy produced.
Everything else -- seed selection on the EDT, retries, recovery test, output
arrays -- is the reference's unmodified code.
"""
import os
import sys

os.environ['PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION'] = 'python'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('FFN_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, 'tools', 'ref_shims'))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from scipy import ndimage  # noqa: E402

from ffn.inference import align as ref_align  # noqa: E402
from ffn.inference import executor as ref_executor  # noqa: E402
from ffn.inference import inference as ref_inference  # noqa: E402
from ffn.inference import inference_pb2  # noqa: E402
from ffn.inference import inference_utils as ref_utils  # noqa: E402
from ffn.inference import movement as ref_movement  # noqa: E402
from ffn.inference import resegmentation as ref_reseg  # noqa: E402
from ffn.training import model as ref_model  # noqa: E402

from ffn_amd import synthetic  # noqa: E402
from oracle import ffn_oracle  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


class _LegacyNumpy:
  """numpy with the pre-1.24 behaviour for ragged np.array(...)."""

  def __getattr__(self, name):
    return getattr(np, name)

  @staticmethod
  def array(obj, *args, **kwargs):
    try:
      return np.array(obj, *args, **kwargs)
    except ValueError:
      out = np.empty(len(obj), dtype=object)
      for k, v in enumerate(obj):
        out[k] = v
      return out

  @staticmethod
  def savez_compressed(fd, **kw):
    fixed = {}
    for k, v in kw.items():
      try:
        fixed[k] = np.asarray(v)
      except ValueError:
        fixed[k] = _LegacyNumpy.array(v)
    return np.savez_compressed(fd, **fixed)


ref_reseg.np = _LegacyNumpy()


class OracleClient(ref_executor.ExecutorClient):

  def __init__(self, blob, depth):
    self.blob, self.depth = blob, depth

  def start(self):
    return 0

  def finish(self):
    pass

  def predict(self, seed, image, fetches):
    return {'logits': ffn_oracle.forward(image, seed, self.blob,
                                         self.depth)[..., None]}


class StandInRunner:
  """What process_point needs of Runner (runner.py:307-414): make_canvas with
  identity alignment, init_seg_volume, counters."""

  def __init__(self, volume_u8, init_seg, blob, depth, request):
    self.volume = volume_u8
    self.init_seg_volume = init_seg[np.newaxis]
    self.counters = ref_utils.Counters()
    self.blob, self.depth = blob, depth
    self.request = request
    self.info = ref_model.ModelInfo(
        deltas=np.array([8, 8, 8]), pred_mask_size=np.array([33, 33, 33]),
        input_seed_size=np.array([33, 33, 33]),
        input_image_size=np.array([33, 33, 33]))

  def make_canvas(self, corner, subvol_size, **kwargs):
    corner = np.array(corner)
    end = corner + np.array(subvol_size)
    sel = tuple(slice(int(c), int(e)) for c, e in zip(corner, end))
    image = (self.volume[sel].astype(np.float32) - 128.0) / 33.0
    canvas = ref_inference.Canvas(
        self.info, OracleClient(self.blob, self.depth), image,
        self.request.inference_options,
        counters=self.counters.get_sub_counters(),
        movement_policy_fn=ref_movement.get_policy_fn(self.request, self.info),
        corner_zyx=corner, keep_probability_maps=True, **kwargs)
    canvas.init_segmentation_from_volume(self.init_seg_volume, corner, end)
    alignment = ref_align.Aligner().generate_alignment(corner, subvol_size)
    return canvas, alignment


def main():
  weights = dict(np.load(os.path.join(GOLD, 'fib25_weights.npz')))
  blob = ffn_oracle.weights_blob(weights, 12)
  shape = (80, 80, 80)
  vol = synthetic.cells_volume(shape, seed=41, membrane_dilate=2)
  # initial segmentation: 6-connected components of the cell interiors
  labels, _ = ndimage.label(vol > 110)
  sizes = np.bincount(labels.ravel())
  sizes[0] = 0
  # two large components that come close to each other near the centre
  centre = np.array(shape) // 2
  best = None
  for a in np.argsort(sizes)[::-1][:12]:
    if a == 0 or sizes[a] < 3000:
      continue
    grown = ndimage.binary_dilation(labels == a, iterations=8)
    for b in np.unique(labels[grown]):
      if b in (0, a) or sizes[b] < 3000:
        continue
      zone = np.argwhere(grown & (labels == b))
      d = np.abs(zone - centre).max(axis=1)
      k = int(np.argmin(d))
      if d[k] <= 15 and (best is None or d[k] < best[0]):
        best = (int(d[k]), int(a), int(b), tuple(int(v) for v in zone[k]))
  assert best is not None, 'no suitable object pair in the phantom'
  _, id_a, id_b, point = best
  init_seg = labels.astype(np.uint64) * 7 + 100  # sparse, non-contiguous ids
  init_seg[labels == 0] = 0
  gid_a, gid_b = id_a * 7 + 100, id_b * 7 + 100

  request = inference_pb2.ResegmentationRequest()
  o = request.inference.inference_options
  o.init_activation = 0.95
  o.pad_value = 0.05
  o.move_threshold = 0.9
  o.segment_threshold = 0.6
  o.min_segment_size = 1000
  o.min_boundary_dist.x = o.min_boundary_dist.y = o.min_boundary_dist.z = 1
  request.radius.x = request.radius.y = request.radius.z = 24
  request.output_directory = '/tmp/ref_reseg_out'
  request.max_retry_iters = 2
  request.exclusion_radius.x = request.exclusion_radius.y = 4
  request.exclusion_radius.z = 4
  request.segment_recovery_fraction = 0.5
  request.analysis_radius.x = request.analysis_radius.y = 8
  request.analysis_radius.z = 8
  p = request.points.add()
  p.id_a, p.id_b = gid_a, gid_b
  p.point.z, p.point.y, p.point.x = point
  p2 = request.points.add()  # endpoint request: id_b omitted
  p2.id_a = gid_b
  p2.point.z, p2.point.y, p2.point.x = point

  out = {'volume': vol, 'init_seg': init_seg,
         'point': np.array(point), 'ids': np.array([gid_a, gid_b], np.uint64)}
  os.system('rm -rf /tmp/ref_reseg_out')
  runner = StandInRunner(vol, init_seg, blob, 12, request.inference)
  for n in range(2):
    path = ref_reseg.get_target_path(request, n)
    ref_reseg.process_point(request, runner, n, voxel_size=(1, 1, 1))
    with np.load(path, allow_pickle=True) as d:
      for key in ('probs', 'raw_probs', 'deletes', 'histories',
                  'corner_zyx', 'is_shift'):
        out['p%d_%s' % (n, key)] = d[key]
      sp = d['start_points']
      out['p%d_start_points_a' % n] = np.array(sp[0]).reshape(-1, 3)
      out['p%d_start_points_b' % n] = np.array(sp[1]).reshape(-1, 3)
      out['p%d_name' % n] = os.path.basename(path)
    print('point', n, os.path.basename(path), 'probs', out['p%d_probs' % n].shape,
          'starts', sp, 'histories',
          [len(h) for h in out['p%d_histories' % n]])
  dst = os.path.join(GOLD, 'ref_reseg.npz')
  np.savez_compressed(dst, **out)
  print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
  main()
