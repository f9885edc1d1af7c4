#!/usr/bin/env python3
"""Mints the golden fixtures under tests/golden/ (TEST TOOLING).

Runs in the build container only (needs /root/reference).  The reference's own
*unmodified* Python modules are imported through tools/ref_shims and driven
with fixed synthetic inputs; the conv-stack forward (TensorFlow in the
reference, unavailable here) is supplied by oracle/ffn_oracle.forward.

Outputs (committed):
  tests/golden/fib25_weights.npz   the reference's shipped FIB-25 checkpoint
                                   (models/fib25/model.ckpt-27465036) as f32
                                   arrays keyed by TF variable name
  tests/golden/ref_movement.npz    movement.get_scored_move_offsets KATs
  tests/golden/ref_misc.json       storage / segmentation / threshold KATs
  tests/golden/ref_canvas_*.npz    Canvas.segment_all runs: per-step FoV
                                   positions + queued moves, final
                                   segmentation, counters, origins
  tests/golden/ref_canvas_options.npz   (--only options) Canvas.segment_all runs
                                   under InferenceOptions away from the sample
                                   configuration: disco off / positive,
                                   min_boundary_dist > 1, other segment
                                   thresholds and size filters, probability maps
  tests/golden/ref_masks.npz       storage.build_mask KATs and a Canvas run
                                   under a MovementRestrictor (mask, seed
                                   mask, shift mask)
  tests/golden/ref_canvas_cells250.npz   (--only cells250, ~15 min) the same at
                                   the BASELINE size: 3,658 FoV steps on the
                                   250^3 bench volume
  tests/golden/ref_canvas_cells250_{onednn,f64}.npz   (--only cells250
                                   --forward onednn|f64) the same run with the
                                   torch-CPU f32 / f64 conv stack behind the
                                   reference Canvas (3,725 steps)

Usage:  python tools/make_golden.py [--only weights|movement|misc|canvas|masks]
"""

import argparse
import json
import os
import sys

os.environ['PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION'] = 'python'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('FFN_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, 'tools', 'ref_shims'))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
from scipy.special import logit  # noqa: E402

from ffn.inference import executor as ref_executor  # noqa: E402
from ffn.inference import inference as ref_inference  # noqa: E402
from ffn.inference import inference_pb2  # noqa: E402
from ffn.inference import inference_utils as ref_utils  # noqa: E402
from ffn.inference import movement as ref_movement  # noqa: E402
from ffn.inference import seed as ref_seed  # noqa: E402
from ffn.inference import segmentation as ref_segmentation  # noqa: E402
from ffn.inference import storage as ref_storage  # noqa: E402
from ffn.training import model as ref_model  # noqa: E402

from ffn_amd import synthetic  # noqa: E402
from ffn_amd.training import tf_checkpoint  # noqa: E402
from oracle import ffn_oracle  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
CKPT = os.path.join(REF, 'models', 'fib25', 'model.ckpt-27465036')


def make_weights():
  v = tf_checkpoint.load_checkpoint(CKPT)
  keep = {k: a for k, a in v.items() if k.startswith('seed_update/')}
  np.savez_compressed(os.path.join(GOLD, 'fib25_weights.npz'), **keep)
  print('weights:', len(keep), 'tensors',
        sum(a.size for a in keep.values()), 'floats')


def make_movement():
  thr = float(logit(float(np.float32(0.9))))
  out = {}
  rng = np.random.RandomState(7)
  cases = [
      ('iso', (8, 8, 8), rng.normal(0, 2, (33, 33, 33)).astype(np.float32)),
      ('const', (8, 8, 8), np.full((33, 33, 33), 3.0, np.float32)),
      ('aniso', (5, 10, 10), rng.normal(0, 2, (21, 41, 41)).astype(np.float32)),
      ('below', (8, 8, 8), np.full((33, 33, 33), 1.0, np.float32)),
  ]
  for i in range(6):
    cases.append(('rand%d' % i, (8, 8, 8),
                  rng.normal(1.5, 1.5, (33, 33, 33)).astype(np.float32)))
  for name, deltas, pm in cases:
    res = sorted(
        ref_movement.get_scored_move_offsets(deltas, pm, threshold=thr),
        reverse=True)
    out[name + '_deltas'] = np.array(deltas)
    out[name + '_map'] = pm
    out[name + '_scores'] = np.array([r[0] for r in res], np.float32)
    out[name + '_offsets'] = np.array([r[1] for r in res],
                                      np.int64).reshape(-1, 3)
  out['threshold'] = np.float64(thr)
  np.savez_compressed(os.path.join(GOLD, 'ref_movement.npz'), **out)
  print('movement KATs:', len(cases))


class _FakeCanvas:
  pass


def make_misc():
  out = {}
  q = ref_storage.quantize_probability(
      np.array([0, .001, .5, .6, .95, 1, np.nan]))
  out['quantize_in'] = [0, .001, .5, .6, .95, 1, 'nan']
  out['quantize_out'] = [int(x) for x in q]
  dq = ref_storage.dequantize_probability(np.array([0, 1, 128, 255]))
  out['dequantize_out'] = [None if np.isnan(x) else float(x) for x in dq]
  out['reduce_id_bits'] = {
      str(m): str(ref_segmentation.reduce_id_bits(np.array([0, m])).dtype)
      for m in (255, 256, 65535, 65536, 70000)
  }
  out['subvolume_path'] = ref_storage.subvolume_path('out', (3, 2, 1), 'npz')
  out['checkpoint_path'] = ref_storage.checkpoint_path('out', (3, 2, 1))
  out['object_prob_path'] = ref_storage.object_prob_path('out', (3, 2, 1))
  # Logit-space options as the Canvas stores them (inference.py:189-195).
  opts = inference_pb2.InferenceOptions()
  opts.init_activation = 0.95
  opts.pad_value = 0.05
  opts.move_threshold = 0.9
  opts.segment_threshold = 0.6
  for attr in ('init_activation', 'pad_value', 'move_threshold',
               'segment_threshold'):
    setattr(opts, attr, logit(getattr(opts, attr)))
    out['logit_' + attr] = float(getattr(opts, attr))
  out['disco_seed_threshold_default'] = float(opts.disco_seed_threshold)
  out['policy_threshold'] = float(logit(float(np.float32(0.9))))
  # quantize_pos (movement.py:200-208)
  pol = ref_movement.FaceMaxMovementPolicy.__new__(
      ref_movement.FaceMaxMovementPolicy)
  pol.deltas = np.array([8, 8, 8])
  pol._start_pos = (100, 100, 100)
  out['quantize_pos'] = {
      str(p): [int(v) for v in pol.quantize_pos(p)]
      for p in [(104, 100, 100), (103, 100, 100), (96, 100, 100),
                (95, 100, 100), (108, 92, 116)]
  }
  # PolicyGrid3d coordinates after the margin filter (seed.py:63-95,411-430).
  fc = _FakeCanvas()
  fc.image = np.zeros((50, 56, 60), np.uint8)
  fc.shape = fc.image.shape
  fc.margin = np.array([16, 16, 16])

  class _C:  # weakref.proxy needs a weakref-able object
    pass

  c = _C()
  c.__dict__.update(fc.__dict__)
  pol = ref_seed.PolicyGrid3d(c, step=16, offsets=(0, 8))
  out['grid3d_seeds'] = [[int(v) for v in p] for p in pol]
  with open(os.path.join(GOLD, 'ref_misc.json'), 'w') as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print('misc KATs written')


class OracleClient(ref_executor.ExecutorClient):
  """ExecutorClient (executor.py:85-108) backed by the oracle forward."""

  def __init__(self, blob, depth, log, forward_fn=None):
    self.blob = blob
    self.depth = depth
    self.log = log
    self.forward_fn = forward_fn  # f(image, seed) -> logits; default: C oracle

  def start(self):
    return 0

  def finish(self):
    pass

  def predict(self, seed, image, fetches):
    if self.forward_fn is not None:
      out = self.forward_fn(image[..., 0] if image.ndim == 5 else image,
                            seed[..., 0] if seed.ndim == 5 else seed)
    else:
      out = ffn_oracle.forward(image, seed, self.blob, self.depth)
    return {'logits': out[..., None]}


def run_reference_canvas(image_f32, blob, depth, fov, deltas, seeds,
                         min_segment_size=1000, forward_fn=None,
                         restrictor=None, pred=None, options=None,
                         keep_probability_maps=False):
  """Drives the reference Canvas exactly as Runner does (runner.py:392-408).
  pred (zyx): a prediction smaller than the seed FoV (ModelInfo.pred_mask_size <
  input_seed_size; inference.py:218,410-411)."""
  info = ref_model.ModelInfo(
      deltas=np.array(deltas[::-1]),
      pred_mask_size=np.array((pred or fov)[::-1]),
      input_seed_size=np.array(fov[::-1]), input_image_size=np.array(fov[::-1]))
  request = inference_pb2.InferenceRequest()
  o = request.inference_options
  o.init_activation = 0.95
  o.pad_value = 0.05
  o.move_threshold = 0.9
  o.segment_threshold = 0.6
  o.min_segment_size = min_segment_size
  o.min_boundary_dist.x = 1
  o.min_boundary_dist.y = 1
  o.min_boundary_dist.z = 1
  for key, value in (options or {}).items():  # InferenceOptions the sample config leaves alone
    if key == 'min_boundary_dist':
      o.min_boundary_dist.x, o.min_boundary_dist.y, o.min_boundary_dist.z = value
    else:
      setattr(o, key, value)
  counters = ref_utils.Counters()
  trace = []
  canvas = ref_inference.Canvas(
      info, OracleClient(blob, depth, trace, forward_fn), image_f32, o,
      counters=counters, restrictor=restrictor,
      movement_policy_fn=ref_movement.get_policy_fn(request, info),
      keep_probability_maps=keep_probability_maps)

  # Record each FoV step: position + the moves the policy queued.
  orig_update = canvas.movement_policy.update

  def recording_update(prob_map, position):
    before = len(canvas.movement_policy.scored_coords)
    orig_update(prob_map, position)
    new = list(canvas.movement_policy.scored_coords)[before:]
    trace.append((tuple(int(p) for p in position),
                  [(float(s), tuple(int(c) for c in xyz)) for s, xyz in new]))

  canvas.movement_policy.update = recording_update

  class FixedSeeds(ref_seed.BaseSeedPolicy):

    def init_coords(self):
      self.coords = np.array(seeds)

  canvas.segment_all(seed_policy=FixedSeeds)
  return canvas, trace, counters


def make_canvas_case(name, shape, seed, depth_weights, grid_step, grid_offsets,
                     dilate, pred=None):
  vol = synthetic.cells_volume(shape, seed=seed, membrane_dilate=dilate)
  image = synthetic.normalize(vol)
  blob, depth = depth_weights
  fov = (33, 33, 33)
  deltas = (8, 8, 8)
  seeds = ffn_oracle.grid_seeds(shape, (16, 16, 16), step=grid_step,
                                offsets=grid_offsets)
  forward_fn = None
  if pred is not None:
    # the model's `logits` = (seed + update) of the centred pred box: what a
    # network with pred_mask_size < input_seed_size hands to Canvas.update_at
    lo = [(f - p) // 2 for f, p in zip(fov, pred)]
    sel = tuple(slice(l, l + p) for l, p in zip(lo, pred))

    def forward_fn(img, sd):
      return np.ascontiguousarray(ffn_oracle.forward(img, sd, blob, depth)[sel])

  canvas, trace, counters = run_reference_canvas(image, blob, depth, fov,
                                                 deltas, seeds,
                                                 forward_fn=forward_fn, pred=pred)
  seg = np.array(canvas.segmentation)
  steps = np.array([t[0] for t in trace], np.int32).reshape(-1, 3)
  n_moves = np.array([len(t[1]) for t in trace], np.int32)
  move_scores = np.array([s for t in trace for s, _ in t[1]], np.float32)
  move_coords = np.array([c for t in trace for _, c in t[1]],
                         np.int32).reshape(-1, 3)
  origins = {
      int(k): [list(int(x) for x in v.start_zyx), int(v.iters)]
      for k, v in canvas.origins.items()
  }
  cdict = {k: c.value for k, c in counters}
  keep = {k: v for k, v in cdict.items() if not k.endswith('-time-ms')}
  np.savez_compressed(
      os.path.join(GOLD, 'ref_canvas_%s.npz' % name),
      volume=vol, seeds=seeds, segmentation=seg, seed_logits=np.array(
          canvas.seed), steps=steps, n_moves=n_moves, move_scores=move_scores,
      move_coords=move_coords, origins=json.dumps(origins),
      counters=json.dumps(keep), depth=depth,
      pred_zyx=np.array(pred or fov, np.int32))
  print(name, 'steps', len(steps), 'segments', len(origins), 'counters', keep)


# InferenceOptions (inference.proto:131-168) away from the sample configuration's
# values: each case is a whole reference Canvas.segment_all run
OPTION_CASES = [
    # disco bias off (inference.py:416)
    ('nodisco', 'cells72', {'disco_seed_threshold': -1.0}, False),
    # ... and with a positive active-voxel fraction (inference.py:427)
    ('disco002', 'cells72', {'disco_seed_threshold': 0.002}, False),
    ('disco30', 'cells72', {'disco_seed_threshold': 0.3}, False),
    # seeds further from what is segmented already (inference.py:556-562)
    ('mbd2', 'cells72', {'min_boundary_dist': (2, 2, 2)}, False),
    ('mbd3', 'cells72', {'min_boundary_dist': (3, 2, 4)}, False),
    # another mask threshold / size filter (inference.py:624,639) + quantised
    # probabilities kept (inference.py:229-232,656)
    ('seg05_probmap', 'cells72', {'segment_threshold': 0.5, 'min_segment_size': 100}, True),
    ('seg08_probmap', 'cells72', {'segment_threshold': 0.8, 'min_segment_size': 3000,
                                  'move_threshold': 0.8}, True),
]
PHANTOMS = {'cells56': ((56, 56, 56), 11, 16, (0, 8), 1),
            'cells72': ((72, 64, 80), 5, 16, (0,), 2)}


def make_option_cases(blob, depth):
  out = {}
  for name, phantom, options, probmap in OPTION_CASES:
    shape, vseed, step, offsets, dilate = PHANTOMS[phantom]
    vol = synthetic.cells_volume(shape, seed=vseed, membrane_dilate=dilate)
    seeds = ffn_oracle.grid_seeds(shape, (16, 16, 16), step=step, offsets=offsets)
    canvas, trace, counters = run_reference_canvas(
        synthetic.normalize(vol), blob, depth, (33, 33, 33), (8, 8, 8), seeds,
        options=options, keep_probability_maps=probmap)
    cdict = {k: c.value for k, c in counters}
    keep = {k: v for k, v in cdict.items() if not k.endswith('-time-ms')}
    out[name + '/phantom'] = phantom
    out[name + '/options'] = json.dumps(options)
    out[name + '/seeds'] = seeds
    out[name + '/steps'] = np.array([t[0] for t in trace], np.int32).reshape(-1, 3)
    out[name + '/move_scores'] = np.array([s for t in trace for s, _ in t[1]], np.float32)
    out[name + '/segmentation'] = np.array(canvas.segmentation).astype(np.int16)
    out[name + '/seed_logits'] = np.array(canvas.seed)
    out[name + '/counters'] = json.dumps(keep)
    out[name + '/origins'] = json.dumps({
        int(k): [list(int(x) for x in v.start_zyx), int(v.iters)]
        for k, v in canvas.origins.items()})
    if probmap:
      out[name + '/seg_prob'] = np.array(canvas.seg_prob)
    print(name, phantom, options, 'steps', len(trace), 'segments', len(canvas.origins),
          keep)
  np.savez_compressed(os.path.join(GOLD, 'ref_canvas_options.npz'), **out)


def make_masks(blob, depth):
  """Exclusion masks: storage.build_mask KATs (reference storage.py:323-411) and
  a Canvas run restricted by a MovementRestrictor with a mask, a seed mask and a
  shift mask (reference movement.py:247-336, inference.py:497-499,573-577)."""
  from ffn.utils import bounding_box as ref_bbox
  out = {}
  rng = np.random.RandomState(5)
  corner, size = (5, 7, 9), (20, 24, 28)
  image = rng.randint(0, 256, size).astype(np.uint8)
  labels = rng.randint(0, 6, (2,) + tuple(c + s + 3 for c, s in zip(corner, size))
                       ).astype(np.uint64)  # a 4-d "volume" mask source
  configs = []
  c = inference_pb2.MaskConfig()
  c.coordinate_expression.expression = '(x + 2 * y > 60) & (z % 3 == 0)'
  configs.append(c)
  c = inference_pb2.MaskConfig()
  ch = c.image.channels.add()
  ch.channel = 0
  ch.min_value = 100
  ch.max_value = 140
  ch = c.image.channels.add()
  ch.channel = 0
  ch.values.extend([3, 250])
  ch.invert = False
  configs.append(c)
  c = inference_pb2.MaskConfig()
  c.volume.mask.hdf5 = 'unused:unused'
  ch = c.volume.channels.add()
  ch.channel = 1
  ch.values.extend([2, 5])
  ch = c.volume.channels.add()
  ch.channel = 0
  ch.min_value = 0
  ch.max_value = 1
  ch.invert = True
  c.invert = True
  configs.append(c)
  vol_map = {configs[2].volume.mask.SerializeToString(): labels}
  for i, sel in enumerate(([0], [1], [2], [0, 1, 2])):
    m = ref_storage.build_mask([configs[k] for k in sel], corner, size,
                               dict(vol_map), image)
    out['build_mask_%d' % i] = np.asarray(m, bool)
  out['bm_corner'] = np.array(corner)
  out['bm_size'] = np.array(size)
  out['bm_image'] = image
  out['bm_labels'] = labels

  # a restricted Canvas run on the cells72 phantom (94 steps unrestricted)
  shape = (72, 64, 80)
  vol = synthetic.cells_volume(shape, seed=5, membrane_dilate=2)
  img = synthetic.normalize(vol)
  seeds = ffn_oracle.grid_seeds(shape, (16, 16, 16), step=16, offsets=(0,))
  zz, yy, xx = np.mgrid[0:72, 0:64, 0:80]
  mask = (xx > 50) & (yy < 30)              # no FoV centred here
  seed_mask = (zz < 24) & (xx < 30)         # no segment started here
  shift = np.zeros((2, 72, 32, 40), np.float32)  # scale 2 in y, x
  shift[0, 40:43, 14:16, 18:20] = 7.0
  shift[1, 24:26, 9:10, 24:26] = -5.0
  restrictor = ref_movement.MovementRestrictor(
      mask=mask, seed_mask=seed_mask, shift_mask=shift,
      shift_mask_fov=ref_bbox.BoundingBox(start=(-6, -6, -4), size=(13, 13, 9)),
      shift_mask_threshold=4, shift_mask_scale=2)
  canvas, trace, counters = run_reference_canvas(
      img, blob, depth, (33, 33, 33), (8, 8, 8), seeds, restrictor=restrictor)
  cdict = {k: c.value for k, c in counters}
  out.update(
      run_volume=vol, run_seeds=seeds, run_mask=mask, run_seed_mask=seed_mask,
      run_shift=shift, run_segmentation=np.array(canvas.segmentation),
      run_steps=np.array([t[0] for t in trace], np.int32).reshape(-1, 3),
      run_counters=json.dumps({k: v for k, v in cdict.items()
                               if not k.endswith('-time-ms')}))
  np.savez_compressed(os.path.join(GOLD, 'ref_masks.npz'), **out)
  print('masks: run of', len(trace), 'steps; counters',
        {k: v for k, v in cdict.items() if k.startswith('skip')})


def _forward_fn(forward, blob, variables, depth, threads):
  import functools
  if forward == 'f64c':
    return functools.partial(ffn_oracle.forward_f64c, blob=blob, depth=depth,
                             threads=threads)
  return functools.partial(ffn_oracle.forward_torch, variables=variables, depth=depth,
                           threads=threads, f64=forward == 'f64')


def make_cells250(blob, depth, forward='oracle', variables=None, num_seeds=14,
                  threads=None, tag='', volume_seed=1234):
  """BASELINE configs[1] at its full size: the 250^3 cells phantom of bench.py
  (synthetic.cells_volume seed 1234), the first row of the bench's seed grid
  (14 seeds, 5 of which start a segment) -> 3,658 FoV steps through the
  reference's Canvas with the oracle forward.  The volume is NOT stored (it is
  regenerated from its seed; a checksum pins it).  About 15 minutes on 8 cores."""
  import hashlib
  import time
  shape = (250, 250, 250)
  vol = synthetic.cells_volume(shape, seed=volume_seed)  # (1234: bench.py's volume)
  image = synthetic.normalize(vol)
  seeds = ffn_oracle.grid_seeds(shape, (16, 16, 16))
  if num_seeds > 0:  # 14 = the first row of the grid; 0 = the WHOLE volume
    seeds = seeds[:num_seeds]
  threads = threads or os.cpu_count() or 1
  ffn_oracle.set_threads(threads)
  t0 = time.time()
  forward_fn, suffix = None, ''
  if forward != 'oracle':
    # the same run with another CORRECT implementation of the conv stack:
    # 'onednn' = torch-CPU f32 (blocked / vectorised sums, the kind of kernel
    # TensorFlow's CPU path also uses), 'f64' = double precision throughout
    # (torch), 'f64c' = the same double-precision arithmetic from
    # oracle/convstack_f64.c (its f32-rounded logits equal torch's bit for bit
    # on the sample FoV; 0.19 s instead of 1.8 s per FoV on 4 cores)
    forward_fn = _forward_fn(forward, blob, variables, depth, threads)
    suffix = '_' + ('f64' if forward == 'f64c' else forward)
  suffix += tag
  canvas, trace, counters = run_reference_canvas(image, blob, depth,
                                                 (33, 33, 33), (8, 8, 8), seeds,
                                                 forward_fn=forward_fn)
  wall = time.time() - t0
  seg = np.array(canvas.segmentation)
  steps = np.array([t[0] for t in trace], np.int16).reshape(-1, 3)
  n_moves = np.array([len(t[1]) for t in trace], np.int8)
  move_scores = np.array([s for t in trace for s, _ in t[1]], np.float32)
  move_coords = np.array([c for t in trace for _, c in t[1]],
                         np.int16).reshape(-1, 3)
  origins = {
      int(k): [list(int(x) for x in v.start_zyx), int(v.iters)]
      for k, v in canvas.origins.items()
  }
  cdict = {k: c.value for k, c in counters}
  keep = {k: v for k, v in cdict.items() if not k.endswith('-time-ms')}
  final_seed = np.array(canvas.seed)
  np.savez_compressed(
      os.path.join(GOLD, 'ref_canvas_cells250%s.npz' % suffix),
      forward=forward, volume_sha256=hashlib.sha256(vol.tobytes()).hexdigest(), seeds=seeds,
      volume_seed=volume_seed,
      segmentation=seg.astype(np.int8 if seg.max() < 128 else np.int16),
      steps=steps, n_moves=n_moves, num_seeds=len(seeds),
      move_scores=move_scores, move_coords=move_coords,
      origins=json.dumps(origins), counters=json.dumps(keep), depth=depth,
      # the seed array after the LAST segment, around its start (a 33^3 sample
      # of logits; the full 62 MB array is not stored)
      final_seed_sample=final_seed[0:33, 0:33, 192:225],
      mint_wall_seconds=wall)
  print('cells250 steps', len(steps), 'segments', len(origins), 'wall %.0f s'
        % wall, 'counters', keep)


def make_phantoms(blob, depth, variables, vol_seeds, size, threads, forward='onednn',
                  tag=''):
  """An ENSEMBLE of whole-volume runs: `size`^3 cells phantoms of several seeds, every
  grid seed, through the reference's Canvas behind the torch-CPU / oneDNN f32
  forward (the configuration of the 250^3 whole-volume fixtures, at a size that
  takes minutes instead of an hour).  What it is for: how OFTEN two correct
  implementations of the forward leave each other's trajectory on this kind of
  volume, and what the segmentations then still share (tests/test_gpu_round5.py::
  test_phantom_ensemble, profiles/r05_phantom_ensemble.txt)."""
  import functools
  import time
  out = {'vol_seeds': np.array(vol_seeds, np.int32), 'size': np.int32(size)}
  shape = (size,) * 3
  forward_fn = _forward_fn(forward, blob, variables, depth, threads)
  for vs in vol_seeds:
    vol = synthetic.cells_volume(shape, seed=int(vs))
    seeds = ffn_oracle.grid_seeds(shape, (16, 16, 16))
    t0 = time.time()
    canvas, trace, counters = run_reference_canvas(
        synthetic.normalize(vol), blob, depth, (33, 33, 33), (8, 8, 8), seeds,
        forward_fn=forward_fn)
    seg = np.array(canvas.segmentation)
    k = 's%d/' % vs
    out[k + 'seeds'] = seeds.astype(np.int16)
    out[k + 'steps'] = np.array([t[0] for t in trace], np.int16).reshape(-1, 3)
    out[k + 'segmentation'] = seg.astype(np.int8 if seg.max() < 128 else np.int16)
    out[k + 'objects'] = np.int32(len(canvas.origins))
    out[k + 'voxels'] = np.int64((seg > 0).sum())
    print('phantom seed %d: %d steps, %d objects, %d voxels, %.0f s' % (
        vs, len(trace), len(canvas.origins), int((seg > 0).sum()), time.time() - t0),
          flush=True)
  np.savez_compressed(os.path.join(GOLD, 'ref_canvas_phantoms%d%s.npz' % (size, tag)), **out)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--only', default='')
  ap.add_argument('--forward', default='oracle',
                  choices=['oracle', 'onednn', 'f64', 'f64c'],
                  help='cells250: the conv-stack implementation behind the '
                  "reference Canvas (default: the C oracle's sequential f32 "
                  'fmaf chain)')
  ap.add_argument('--num-seeds', type=int, default=14,
                  help='cells250: seeds of the grid to run (14 = first row, '
                  '0 = all of them: the whole volume)')
  ap.add_argument('--threads', type=int, default=0)
  ap.add_argument('--tag', default='', help='cells250: file-name suffix')
  ap.add_argument('--phantom-seeds', type=int, nargs='+',
                  default=[101, 102, 103, 104, 105, 106])
  ap.add_argument('--phantom-size', type=int, default=128)
  ap.add_argument('--volume-seed', type=int, default=1234,
                  help="cells250: seed of the synthetic phantom (1234 = bench.py's)")
  args = ap.parse_args()
  os.makedirs(GOLD, exist_ok=True)
  if args.only in ('', 'weights'):
    make_weights()
  if args.only in ('', 'movement'):
    make_movement()
  if args.only in ('', 'misc'):
    make_misc()
  if args.only in ('', 'canvas'):
    v = tf_checkpoint.load_checkpoint(CKPT)
    blob = ffn_oracle.weights_blob(v, 12)
    make_canvas_case('cells56', (56, 56, 56), 11, (blob, 12), 16, (0, 8), 1)
    make_canvas_case('cells72', (72, 64, 80), 5, (blob, 12), 16, (0,), 2)
  if args.only in ('', 'predcrop'):
    # pred_mask_size (25^3) < input_seed_size (33^3)
    v = tf_checkpoint.load_checkpoint(CKPT)
    blob = ffn_oracle.weights_blob(v, 12)
    make_canvas_case('cells56_pred25', (56, 56, 56), 11, (blob, 12), 16, (0, 8), 1,
                     pred=(25, 25, 25))
    make_canvas_case('cells72_pred27', (72, 64, 80), 5, (blob, 12), 16, (0,), 2,
                     pred=(27, 29, 25))
  if args.only in ('', 'options'):
    v = tf_checkpoint.load_checkpoint(CKPT)
    make_option_cases(ffn_oracle.weights_blob(v, 12), 12)
  if args.only in ('', 'masks'):
    v = tf_checkpoint.load_checkpoint(CKPT)
    make_masks(ffn_oracle.weights_blob(v, 12), 12)
  if args.only == 'phantoms':  # ~40 minutes: only on request
    v = tf_checkpoint.load_checkpoint(CKPT)
    make_phantoms(ffn_oracle.weights_blob(v, 12), 12, v, args.phantom_seeds,
                  args.phantom_size, args.threads or 8,
                  'onednn' if args.forward == 'oracle' else args.forward, args.tag)
  if args.only == 'cells250':  # slow: only on request
    v = tf_checkpoint.load_checkpoint(CKPT)
    make_cells250(ffn_oracle.weights_blob(v, 12), 12, args.forward, v,
                  args.num_seeds, args.threads or None, args.tag, args.volume_seed)


if __name__ == '__main__':
  main()
