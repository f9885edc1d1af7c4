"""The FoV inference loop: `Canvas` (mirror of reference ffn/inference/inference.py).

Two implementations share the reference's public surface (`is_valid_pos`,
`predict`, `update_at`, `init_seed`, `segment_at`, `segment_all`,
`save_checkpoint`, `restore_checkpoint`, `seed`, `segmentation`, `origins`,
`overlaps`, `counters`):

* `Canvas` keeps its state in host numpy arrays and calls
  `exec_client.predict(seed, image, ['logits'])` once per step -- the
  reference's literal contract (inference.py:356-441), usable with ANY
  ExecutorClient (the stateless `ffn_predict` path of HipBatchExecutor, or a
  test double).
* `DeviceCanvas` keeps image / seed / segmentation resident in HBM
  (libffn_hip.so `ffn_canvas_*`).  One C call per FoV step does gather -> conv
  stack -> disco -> paste-back -> 6-face argmax on the GPU and returns ~30
  scalars; the BFS move queue, the seed iteration and the accept/reject logic of
  segments stay in Python exactly as in the reference.

`make_canvas(...)` picks `DeviceCanvas` when the client can host device
canvases (`create_canvas` + `step`), else `Canvas`.
"""

from __future__ import annotations

import collections
import inspect
import ctypes

import logging
import os
import struct
import sys
import threading
import weakref
import time

import numpy as np
from scipy.special import expit
from scipy.special import logit

from .. import _lib
from .. import engine as hip_engine
from ..training import model as ffn_model
from . import executor
from . import movement
from . import seed as seed_lib
from . import storage
from .inference_utils import Counters
from .inference_utils import TimedIter
from .inference_utils import timer_counter
from .request import InferenceOptions

MSEC_IN_SEC = 1000
MAX_SELF_CONSISTENT_ITERS = 32


def _logit_options(options) -> InferenceOptions:
  """Copies `options` and converts the four probability fields to logits,
  rounded to f32 like the reference's proto fields (inference.py:187-195)."""
  out = InferenceOptions()
  out.CopyFrom(options)
  for attr in ('init_activation', 'pad_value', 'move_threshold',
               'segment_threshold'):
    setattr(out, attr, logit(getattr(out, attr)))
  return out


class NormalizedU8Image:
  """A raw uint8 image that stands for `(raw.astype(np.float32) - mean) /
  stddev`, the normalised image Runner.make_canvas hands to the Canvas
  (reference runner.py:383-385).

  A DeviceCanvas uploads `raw_u8` as it is (1 byte per voxel in HBM) and the
  normalisation happens in the FoV gather (ffn_canvas_create_u8), bit-identical
  to the f32 path; host-side consumers (seed policies, a host Canvas, save_raw)
  index or convert it like the f32 array and get the same values."""

  dtype = np.dtype(np.float32)
  ndim = 3

  def __init__(self, raw_u8, mean, stddev):
    self.raw_u8 = np.asarray(raw_u8)
    if self.raw_u8.dtype != np.uint8 or self.raw_u8.ndim != 3:
      raise ValueError('raw_u8 must be a 3d uint8 array')
    self.mean = float(np.float32(mean))
    self.stddev = float(np.float32(stddev))
    self.shape = tuple(self.raw_u8.shape)

  def _norm(self, raw):
    return ((np.asarray(raw).astype(np.float32) - np.float32(self.mean)) /
            np.float32(self.stddev))

  def __getitem__(self, key):
    return self._norm(self.raw_u8[key])

  def __array__(self, dtype=None, copy=None):
    out = self._norm(self.raw_u8)
    return out if dtype is None else out.astype(dtype)

  def __len__(self):
    return self.shape[0]


class Canvas:
  """Tracks state of the inference progress and results within a subvolume."""

  io_lock = threading.Lock()

  def __init__(self, model_info: ffn_model.ModelInfo,
               exec_client: executor.ExecutorClient, image, options,
               voxel_size_zyx=(1, 1, 1), counters=None, restrictor=None,
               movement_policy_fn=None, keep_history=False,
               checkpoint_path=None, checkpoint_interval_sec=0,
               corner_zyx=None, storage_cls=storage.NumpyArray,
               keep_probability_maps=False):
    self.image = image
    self._exec_client = exec_client
    self._exec_client_id = None
    self.voxel_size_zyx = voxel_size_zyx
    self.options = _logit_options(options)
    self.counters = counters if counters is not None else Counters()
    self.checkpoint_interval_sec = checkpoint_interval_sec
    self.checkpoint_path = checkpoint_path
    self.checkpoint_last = time.time()
    self._keep_history = keep_history
    self.corner_zyx = corner_zyx
    self.shape = tuple(image.shape)
    self.restrictor = (movement.MovementRestrictor()
                       if restrictor is None else restrictor)

    # zyx
    self._pred_size = np.array(model_info.pred_mask_size[::-1])
    self._input_seed_size = np.array(model_info.input_seed_size[::-1])
    self._input_image_size = np.array(model_info.input_image_size[::-1])
    self.margin = self._input_image_size // 2
    self._pred_delta = (self._input_seed_size - self._pred_size) // 2
    assert np.all(self._pred_delta >= 0)
    self._margin_t = tuple(int(v) for v in self.margin)

    self.keep_probability_maps = keep_probability_maps
    self._alloc_state(storage_cls)

    self.global_to_local_ids = {}
    self.local_to_global_ids = {}
    self.seed_policy = None
    self._seed_policy_state = None
    self._max_id = 0
    self.origins = {}
    self.overlaps = {}
    self.reset_seed_per_segment = True

    if movement_policy_fn is None:
      self.movement_policy = movement.FaceMaxMovementPolicy(
          self, deltas=model_info.deltas[::-1],
          score_threshold=self.options.move_threshold)
    else:
      self.movement_policy = movement_policy_fn(self)

    self._hosts = []
    self.history = []
    self.history_deleted = []
    self.reset_state((0, 0, 0))
    self.t_last_predict = None
    self.log_info('Constructed canvas with corner %s (zyx) and shape %s',
                  self.corner_zyx, self.shape)

  # -- state allocation (overridden by DeviceCanvas) ---------------------------
  def _alloc_state(self, storage_cls):
    self.seed = storage_cls(shape=self.shape, dtype=np.float32,
                            default_value=np.nan)
    self.segmentation = storage_cls(shape=self.shape, dtype=np.int32)
    if self.keep_probability_maps:
      self.seg_prob = storage_cls(shape=self.shape, dtype=np.uint8)
    else:
      self.seg_prob = None

  # -- executor registration -----------------------------------------------------
  def _register_client(self):
    if self._exec_client_id is None:
      self._exec_client_id = self._exec_client.start()
      logging.info('Registered as client %d.', self._exec_client_id)

  def _deregister_client(self):
    if self._exec_client_id is not None:
      logging.info('Deregistering client %d', self._exec_client_id)
      self._exec_client.finish()
      self._exec_client_id = None

  def __del__(self):
    try:
      self._deregister_client()
    except Exception:  # pylint:disable=broad-except
      pass

  def local_id(self, segment_id: int):
    return self.global_to_local_ids.get(segment_id, segment_id)

  def reset_state(self, start_pos, reset_extents=True):
    """Prepares the canvas for a new inference run (inference.py:291-310)."""
    self.movement_policy.reset_state(start_pos)
    self.history = []
    self.history_deleted = []
    if reset_extents:
      self._min_pos = np.array(start_pos)
      self._max_pos = np.array(start_pos)
    self._register_client()

  # -- validity --------------------------------------------------------------------
  def _in_bounds(self, pos) -> bool:
    m = self._margin_t
    s = self.shape
    return (pos[0] - m[0] >= 0 and pos[0] + m[0] < s[0] and
            pos[1] - m[1] >= 0 and pos[1] + m[1] < s[1] and
            pos[2] - m[2] >= 0 and pos[2] + m[2] < s[2])

  def is_valid_pos(self, pos, ignore_move_threshold=False) -> bool:
    """True if segmentation should be attempted at `pos` (inference.py:312-346)."""
    if not ignore_move_threshold:
      if self.seed[pos] < self.options.move_threshold:
        self.counters['skip_threshold'].Increment()
        return False
    if not self._in_bounds(pos):
      self.counters['skip_invalid_pos'].Increment()
      return False
    if self.segmentation[pos] > 0:
      self.counters['skip_invalid_pos'].Increment()
      return False
    return True

  # -- one FoV step ----------------------------------------------------------------
  def _get_image(self, pos) -> np.ndarray:
    start = np.array(pos) - self.margin
    end = start + self._input_image_size
    return self.image[tuple(slice(s, e) for s, e in zip(start, end))]

  def predict(self, pos, logit_seed: np.ndarray) -> np.ndarray:
    """Runs a single step of FFN prediction (inference.py:356-384)."""
    with timer_counter(self.counters, 'predict'):
      with timer_counter(self.counters, 'get-image'):
        img = self._get_image(pos)
      if self.t_last_predict is not None:
        delta_t = time.time() - self.t_last_predict
        self.counters['inference-not-predict-ms'].IncrementBy(
            delta_t * MSEC_IN_SEC)
      with timer_counter(self.counters, 'inference'):
        fetches = self._exec_client.predict(logit_seed, img, ['logits'])
      self.t_last_predict = time.time()
    logits = fetches.pop('logits')
    return logits[..., 0]

  def update_at(self, pos):
    """Updates the object mask prediction at `pos` (inference.py:386-441)."""
    with timer_counter(self.counters, 'update_at'):
      off = self._input_seed_size // 2
      start = np.array(pos) - off
      end = start + self._input_seed_size
      logit_seed = np.array(
          self.seed[tuple(slice(s, e) for s, e in zip(start, end))])
      init_prediction = np.isnan(logit_seed)
      logit_seed[init_prediction] = np.float32(self.options.pad_value)

      logits = self.predict(pos, logit_seed)
      start += self._pred_delta
      end = start + self._pred_size
      sel = tuple(slice(s, e) for s, e in zip(start, end))

      # Disco bias: never reverse a disconnectedness prediction.
      if self.options.disco_seed_threshold >= 0:
        th_max = logit(0.5)
        old_seed = self.seed[sel]
        if self._keep_history:
          self.history_deleted.append(
              np.sum((old_seed >= logit(0.8)) & (logits < th_max)))
        if (np.mean(logits >= self.options.move_threshold) >
            self.options.disco_seed_threshold):
          with np.errstate(invalid='ignore'):
            mask = (old_seed < th_max) & (logits > old_seed)
          logits[mask] = old_seed[mask]
      self.seed[sel] = logits
    return logits

  def init_seed(self, pos):
    """Reinitialises the object mask with a seed (inference.py:443-450)."""
    self.seed.clear()
    self.seed[pos] = self.options.init_activation

  def get_next_segment_id(self) -> int:
    self._max_id += 1
    while self._max_id in self.origins:
      self._max_id += 1
    return self._max_id

  def _start_logit(self, start_pos):
    return self.seed[start_pos]

  def segment_at(self, start_pos, dynamic_image=None, vis_update_every=10,
                 vis_fixed_z=False, partial_segment_iters=0):
    """Runs FFN segmentation from `start_pos` (inference.py:460-533)."""
    del dynamic_image, vis_update_every, vis_fixed_z  # notebook-only in the ref
    start_pos = tuple(int(v) for v in start_pos)
    if not partial_segment_iters:
      if self.reset_seed_per_segment:
        self.init_seed(start_pos)
      self.reset_state(start_pos, reset_extents=self.reset_seed_per_segment)
      if not self.movement_policy:
        item = (self.movement_policy.score_threshold * 2, start_pos)
        self.movement_policy.append(item)

    num_iters = partial_segment_iters
    with timer_counter(self.counters, 'segment_at-loop'):
      for pos in self.movement_policy:
        if self._start_logit(start_pos) < self.options.move_threshold:
          self.counters['seed_got_too_weak'].Increment()
          break
        if not self.restrictor.is_valid_pos(pos):
          self.counters['skip_restriced_pos'].Increment()
          continue
        pred = self.update_at(pos)
        self._min_pos = np.minimum(self._min_pos, pos)
        self._max_pos = np.maximum(self._max_pos, pos)
        num_iters += 1
        with timer_counter(self.counters, 'movement_policy'):
          self._policy_update(pred, pos)
        if self._keep_history:
          self.history.append(pos)
        self._maybe_save_checkpoint(partial_segment_iters=num_iters)
    return num_iters

  def _policy_update(self, pred, pos):
    self.movement_policy.update(pred, pos)

  def log_info(self, string: str, *args, **kwargs):
    logging.info('[cl %s] ' + string, self._exec_client_id, *args, **kwargs)

  # -- segment bookkeeping hooks (overridden by DeviceCanvas) -------------------------
  def _seg_point(self, pos) -> int:
    return int(self.segmentation[pos])

  def _mark_excluded(self, pos):
    if self.segmentation[pos] == 0:
      self.segmentation[pos] = -1

  def _too_close(self, pos, mbd) -> bool:
    low = np.array(pos) - mbd
    high = np.array(pos) + mbd + 1
    sel = tuple(slice(max(int(s), 0), int(e)) for s, e in zip(low, high))
    if np.any(self.segmentation[sel] > 0):
      self.segmentation[pos] = -1
      return True
    return False

  def _commit(self, sel_lo, sel_hi, pos):
    """mask/overlap/assign of one finished object (inference.py:614-661).

    Returns (raw, actual, overlapped_ids, counts, sid or None)."""
    sel = tuple(slice(l, h) for l, h in zip(sel_lo, sel_hi))
    mask = self.seed[sel] >= self.options.segment_threshold
    raw = int(np.sum(mask))
    overlapped_ids, counts = np.unique(self.segmentation[sel][mask],
                                       return_counts=True)
    valid = overlapped_ids > 0
    overlapped_ids = overlapped_ids[valid]
    counts = counts[valid]
    mask &= self.segmentation[sel] <= 0
    actual = int(np.sum(mask))
    if actual < self.options.min_segment_size:
      return raw, actual, overlapped_ids, counts, None
    sid = self.get_next_segment_id()
    self.segmentation[sel][mask] = sid
    if self.keep_probability_maps:
      self.seg_prob[sel][mask] = storage.quantize_probability(
          expit(self.seed[sel][mask]))
    return raw, actual, overlapped_ids, counts, sid

  # The seed loop is written ONCE, as a generator: wherever the device-resident
  # canvas needs a FoV step it yields the request and receives the result, so
  # the same code runs blocking (`segment_all`) or interleaved with other
  # canvases by a single-threaded scheduler (`MultiCanvasDriver`).  The
  # host-array Canvas never yields: its steps happen inside `segment_at`.
  def _drive(self, gen):
    """Runs a step generator to completion with blocking executor calls."""
    outer = self.__dict__.get('_blocking_drive', False)
    self._blocking_drive = True  # nobody interleaves this canvas' steps
    try:
      req = next(gen)
      while True:
        req = gen.send(self._blocking_step(req))
    except StopIteration as stop:
      return stop.value
    finally:
      self._blocking_drive = outer

  def _blocking_step(self, req):
    raise RuntimeError('host-array Canvas does not yield step requests')

  def _segment_at_gen(self, start_pos, partial_segment_iters=0):
    if False:  # pylint:disable=using-constant-test
      yield None
    return self.segment_at(start_pos,
                           partial_segment_iters=partial_segment_iters)

  def segment_all(self, seed_policy=seed_lib.PolicyPeaks,
                  partial_segment_iters=0):
    """Segments the input image from every seed (inference.py:538-683)."""
    return self._drive(self._segment_all_gen(seed_policy,
                                             partial_segment_iters))

  def _segment_all_gen(self, seed_policy=seed_lib.PolicyPeaks,
                       partial_segment_iters=0):
    self.seed_policy = seed_policy(self)
    if self._seed_policy_state is not None:
      self.seed_policy.set_state(self._seed_policy_state)
      self._seed_policy_state = None

    with timer_counter(self.counters, 'segment_all'):
      mbd = self.options.min_boundary_dist
      mbd = np.array([mbd.z, mbd.y, mbd.x])

      for pos in TimedIter(self.seed_policy, self.counters, 'seed-policy'):
        if not (self.is_valid_pos(pos, ignore_move_threshold=True) and
                self.restrictor.is_valid_pos(pos) and
                self.restrictor.is_valid_seed(pos)):
          assert not partial_segment_iters
          continue
        if not partial_segment_iters:
          self._maybe_save_checkpoint(partial_segment_iters=0)

        if self._too_close(pos, mbd):
          assert not partial_segment_iters
          continue

        self.log_info('Starting segmentation at %r (zyx)', pos)
        seg_start = time.time()
        num_iters = yield from self._segment_at_gen(
            pos, partial_segment_iters=partial_segment_iters)
        partial_segment_iters = 0
        t_seg = time.time() - seg_start

        if num_iters <= 0:
          self.counters['invalid-other-time-ms'].IncrementBy(
              t_seg * MSEC_IN_SEC)
          self.log_info('Failed: num iters was %d', num_iters)
          continue

        if self._start_logit(pos) < self.options.move_threshold:
          self._mark_excluded(pos)
          self.log_info('Failed: weak seed')
          self.counters['invalid-weak-time-ms'].IncrementBy(
              t_seg * MSEC_IN_SEC)
          continue

        # Bounding box of the area the FFN actually changed.
        half = self._pred_size // 2
        lo = [max(int(s), 0) for s in self._min_pos - half]
        hi = [min(int(e) + 1, d)
              for e, d in zip(self._max_pos + half, self.shape)]
        raw, actual, overlapped_ids, counts, sid = self._commit(lo, hi, pos)

        if sid is None:
          self._mark_excluded(pos)
          self.log_info('Failed: too small: %d', actual)
          self.counters['invalid-small-time-ms'].IncrementBy(
              t_seg * MSEC_IN_SEC)
          continue

        self.counters['voxels-segmented'].IncrementBy(actual)
        self.counters['voxels-overlapping'].IncrementBy(raw - actual)
        self.log_info('Created supervoxel:%d  seed(zyx):%s  size:%d  iters:%d',
                      self._max_id, pos, actual, num_iters)
        self.overlaps[self._max_id] = np.array([overlapped_ids, counts])
        self.origins[self._max_id] = storage.OriginInfo(pos, num_iters, t_seg)
        self.counters['valid-time-ms'].IncrementBy(t_seg * MSEC_IN_SEC)
        self._maybe_save_checkpoint(partial_segment_iters=0)

    self.log_info('Segmentation done.')
    self._deregister_client()

  # -- initial segmentation / checkpoints -------------------------------------------------
  def _set_segmentation(self, seg: np.ndarray):
    self.segmentation[:] = seg

  def _set_seed(self, seed: np.ndarray):
    self.seed[:] = seed

  def init_segmentation_from_volume(self, volume, corner, end,
                                    align_and_crop=None):
    """Starts from an existing segmentation (inference.py:685-726)."""
    init_seg = volume[:, corner[0]:end[0], corner[1]:end[1], corner[2]:end[2]]
    init_seg = np.asarray(init_seg[0, ...])
    ids = np.unique(init_seg)
    new = np.arange(len(ids))
    if len(ids) and ids[0] != 0:
      new = new + 1
    self.global_to_local_ids = {int(k): int(v) for k, v in zip(ids, new)}
    self.local_to_global_ids = {
        v: k for k, v in self.global_to_local_ids.items()}
    init_seg = new[np.searchsorted(ids, init_seg)].astype(np.int32)
    if align_and_crop is not None:
      init_seg = align_and_crop(init_seg)
    self._set_segmentation(init_seg)
    if self.keep_probability_maps:
      self.seg_prob[np.asarray(self.segmentation) > 0] = (
          storage.quantize_probability(np.array([1.0])))
    self._max_id = int(np.max(init_seg)) if init_seg.size else 0

  def restore_checkpoint(self, path: str) -> int:
    """Restores state from a checkpoint (inference.py:728-778)."""
    self.log_info('Restoring inference checkpoint: %s', path)
    with open(path, 'rb') as f:
      data = np.load(f, allow_pickle=True)
      self._set_segmentation(data['segmentation'])
      self._set_seed(data['seed'])
      if self.keep_probability_maps:
        self.seg_prob[:] = data['seg_qprob']
      self.history_deleted = list(data['history_deleted'])
      self.history = list(data['history'])
      self.origins = data['origins'].item()
      if 'overlaps' in data:
        self.overlaps = data['overlaps'].item()
      seg = data['segmentation']
      self.counters['voxels-segmented'].Set(int(np.sum(seg != 0)))
      self._max_id = int(np.max(seg))
      self._min_pos = data['min_pos']
      self._max_pos = data['max_pos']
      self.movement_policy.restore_state(data['movement_policy'])
      self._seed_policy_state = data['seed_policy_state']
      self.counters.loads(data['counters'].item())
      partial = (int(data['partial_segment_iters'])
                 if 'partial_segment_iters' in data else 0)
      if 'hosts' in data:
        self._hosts = list(data['hosts'])
    self.log_info('Inference checkpoint restored.')
    return partial

  def save_checkpoint(self, path: str, partial_segment_iters: int):
    """Saves an inference checkpoint (inference.py:780-821)."""
    self.log_info('Saving inference checkpoint to %s.', path)
    with timer_counter(self.counters, 'save_checkpoint'):
      os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
      with storage.atomic_file(path) as fd:
        seed_policy_state = None
        if self.seed_policy is not None:
          seed_policy_state = self.seed_policy.get_state(
              partial_segment_iters > 0)
        aux = {}
        if self.keep_probability_maps:
          aux['seg_qprob'] = np.asarray(self.seg_prob)
        np.savez_compressed(
            fd,
            movement_policy=np.asarray(self.movement_policy.get_state(),
                                       dtype=object),
            segmentation=np.asarray(self.segmentation),
            seed=np.asarray(self.seed),
            origins=self.origins,
            overlaps=self.overlaps,
            min_pos=self._min_pos,
            max_pos=self._max_pos,
            history=np.array(self.history),
            history_deleted=np.array(self.history_deleted),
            seed_policy_state=np.asarray(seed_policy_state, dtype=object),
            counters=self.counters.dumps(),
            partial_segment_iters=partial_segment_iters,
            hosts=self._hosts,
            **aux)
    self.log_info('Inference checkpoint saved.')

  def _checkpoint_due(self) -> bool:
    return (self.checkpoint_path is not None and self.checkpoint_interval_sec > 0 and
            time.time() - self.checkpoint_last >= self.checkpoint_interval_sec)

  def _maybe_save_checkpoint(self, partial_segment_iters=0):
    if not self._checkpoint_due():
      return
    with Canvas.io_lock:
      self.save_checkpoint(self.checkpoint_path,
                           partial_segment_iters=partial_segment_iters)
    self.checkpoint_last = time.time()


# ---------------------------------------------------------------------------
# Device-resident canvas
# ---------------------------------------------------------------------------


class _DeviceArray:
  """ndarray-like view of one device-resident canvas volume.

  Supports what the Canvas surface and downstream code use: point reads
  `arr[(z, y, x)]`, box reads / writes with slices, `clear()`, `np.asarray`.
  """

  def __init__(self, canvas: 'DeviceCanvas', which: str):
    # a weak reference, not a strong one: canvas <-> array would be a cycle, and
    # a cycle keeps a finished subvolume's HBM (12 B / voxel) alive until the
    # cyclic collector happens to run.  LIFETIME: the array is a VIEW of the
    # canvas' device memory -- `np.asarray(canvas.segmentation)` (a host copy)
    # is what outlives `canvas.close()` / the canvas itself; using the view
    # afterwards raises `RuntimeError('canvas closed')`.
    self._ref = weakref.ref(canvas)
    self._which = which
    self.shape = canvas.shape
    self.dtype = np.dtype(np.float32 if which == 'seed' else np.int32)
    self.ndim = 3

  @property
  def _c(self):
    canvas = self._ref()
    if canvas is None or canvas._handle is None:
      raise RuntimeError(
          'canvas closed: its device arrays are views of HBM that is gone; keep '
          'np.asarray(canvas.%s) (a host copy) instead' % (
              'seed' if self._which == 'seed' else 'segmentation'))
    return canvas

  def canvas_handle(self):
    """The ffn_canvas* behind this array (device-side consumers: the assembly
    of sub-box results copies the segmentation inside HBM)."""
    return self._c._handle._h

  def _box(self, key):
    if key is Ellipsis:
      key = (slice(None),) * 3
    if not isinstance(key, tuple):
      key = (key,)
    key = tuple(key) + (slice(None),) * (3 - len(key))
    lo, hi, squeeze = [], [], []
    for k, n in zip(key, self.shape):
      if isinstance(k, slice):
        s, e, st = k.indices(n)
        if st != 1:
          raise IndexError('device arrays support unit-stride slices only')
        lo.append(s)
        hi.append(max(e, s))
        squeeze.append(False)
      else:
        k = int(k)
        if k < 0:
          k += n
        lo.append(k)
        hi.append(k + 1)
        squeeze.append(True)
    return lo, hi, squeeze

  def __getitem__(self, key):
    if (isinstance(key, tuple) and len(key) == 3 and
        not any(isinstance(k, slice) for k in key)):
      sv, gv = self._c._read_point(key)
      return np.float32(sv) if self._which == 'seed' else np.int32(gv)
    lo, hi, squeeze = self._box(key)
    h = self._c._handle
    fn = h.read_seed if self._which == 'seed' else h.read_segmentation
    out = self._c._call(fn, lo, hi)
    idx = tuple(0 if s else slice(None) for s in squeeze)
    return out[idx]

  def __setitem__(self, key, value):
    self._c._invalidate_cache()
    lo, hi, _ = self._box(key)
    h = self._c._handle
    if self._which == 'seed':
      self._c._call(h.write_seed, lo, hi, np.asarray(value, np.float32))
    else:
      self._c._call(h.write_segmentation, lo, hi, np.asarray(value, np.int32))

  def clear(self):
    default = np.nan if self._which == 'seed' else 0
    self[...] = np.full((), default, self.dtype)

  def __array__(self, dtype=None, copy=None):
    arr = self[...]
    return arr if dtype is None else arr.astype(dtype)

  def __len__(self):
    return self.shape[0]


# ffn_step_result (include/ffn_hip.h): face_score[6] face_index[6] face_seg[6]
# start_logit num_above_move disco_applied cand_seed[16] cand_seg[16]
# num_deleted range_error
_STEP_RESULT = struct.Struct('<6f6i6ifIi16f16iIi')
assert _STEP_RESULT.size == ctypes.sizeof(_lib.StepResult)


class DeviceCanvas(Canvas):
  """Canvas whose image / seed / segmentation live in HBM for its lifetime."""

  #: number of queue-head positions whose post-step values ride along with a step
  #: (the loop pops 1.6 entries per step on average; misses cost one point read)
  PREFETCH = 8

  def __init__(self, model_info, exec_client, image, options, **kwargs):
    if not (hasattr(exec_client, 'create_canvas') and
            hasattr(exec_client, 'step')):
      raise TypeError('DeviceCanvas needs an executor client that can host '
                      'device canvases (HipBatchExecutor.get_client)')
    kwargs.pop('storage_cls', None)
    self._handle = None
    self._cache = {}
    self._cached_start = None
    self._hot = [0, 0.0, 0.0, 0.0, 0, 0.0]
    #: in-bounds queue entries that failed the device-state part of
    #: is_valid_pos (seed below threshold / already segmented): the cases a
    #: speculative next step would have mispredicted
    self.gate_rejects = 0
    #: between-segment turns answered by one device call (`_turn`)
    self.turns = 0
    self._turn_rec = None
    self._turn_armed = False
    self._pending = None
    self._step_req = _lib.StepRequest()
    # the request as a flat int32 view: [pos 3][start 3][n 1][candidates 16 x 3]
    self._req_i32 = np.frombuffer(self._step_req, dtype=np.int32)
    self._step_params = _lib.StepParams()
    super().__init__(model_info, exec_client, image, options, **kwargs)
    # pred_mask_size < input_seed_size (inference.py:218,410-411): the engine
    # scores and pastes the centred box only (ffn_engine_set_pred_size)
    engine_pred = getattr(getattr(exec_client, 'engine', None), 'pred_zyx', None)
    if engine_pred is not None and tuple(engine_pred) != tuple(
        int(v) for v in self._pred_size):
      raise ValueError('the engine predicts %r, the model info says %r' % (
          tuple(engine_pred), tuple(self._pred_size)))
    self._step_params.pad_value = self.options.pad_value
    self._step_params.move_threshold = self.options.move_threshold
    self._step_params.disco_seed_threshold = self.options.disco_seed_threshold
    if self._keep_history and self.options.disco_seed_threshold >= 0:
      # history_deleted (inference.py:420-423) is counted on the device
      self._step_params.deleted_threshold = float(np.float32(logit(0.8)))
    self._pred_size_t = tuple(int(v) for v in self._pred_size)
    self._fast_policy = (
        type(self.movement_policy) is movement.FaceMaxMovementPolicy)

  def _alloc_state(self, storage_cls):
    del storage_cls
    if isinstance(self.image, NormalizedU8Image):
      image = self.image  # uploaded raw; normalised in the FoV gather
    else:
      image = np.ascontiguousarray(self.image, dtype=np.float32)
    self._handle = self._exec_client.create_canvas(image)
    self.seed = _DeviceArray(self, 'seed')
    self.segmentation = _DeviceArray(self, 'seg')
    if self.keep_probability_maps:
      self.seg_prob = storage.NumpyArray(shape=self.shape, dtype=np.uint8)
    else:
      self.seg_prob = None

  def _call(self, fn, *args, **kwargs):
    return self._exec_client.canvas_call(fn, *args, **kwargs)

  def close(self):
    """Frees the canvas' device memory now (it is also freed when the last
    reference to the canvas goes away)."""
    if self._handle is not None:
      self._call(self._handle.close)
      self._handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint:disable=broad-except
      pass
    super().__del__()

  # -- cached point reads ------------------------------------------------------------
  def _invalidate_cache(self):
    self._cache = {}
    self._cached_start = None
    self._turn_rec = None

  # -- the between-segment turn as one device call ---------------------------------------
  #: seeds of the policy the device tests ahead in one turn (0: one question per
  #: call, as the reference asks them)
  TURN_CANDIDATES = int(os.environ.get('FFN_AMD_TURN_CANDIDATES', '128'))

  def _turn_ok(self) -> bool:
    """True if `ffn_canvas_segment_turn` may answer the seed loop's questions
    (inference.py:573-660) ahead of time: nothing between two segments that the
    device does not see -- no restrictor masks, no timed checkpoint about to
    be taken, no probability map, the stock validity test."""
    ok = self.__dict__.get('_turn_static')
    if ok is None:
      ok = (hasattr(self._handle, 'segment_turn') and
            not self.keep_probability_maps and
            getattr(self.restrictor, 'is_trivial', self.restrictor is None))
      self._turn_static = ok
    # A timed checkpoint is taken BETWEEN the loop's questions, and a turn moves
    # the canvas past them (the next seed initialised, too-close seeds marked).
    # So: no turn while a checkpoint is DUE -- the loop then asks its questions
    # one by one and the checkpoint sees the state the reference would save --
    # and no checkpoint while a turn's answers are still being used up
    # (`_maybe_save_checkpoint` below: it waits for the next opportunity, where
    # this test keeps the turn from running).
    return (ok and not self._checkpoint_due() and self.TURN_CANDIDATES > 0 and
            getattr(self.is_valid_pos, '__func__', None)
            is DeviceCanvas.is_valid_pos)

  def _policy_list(self):
    """(coords, idx) of the seed policy if it hands its seeds out the stock way
    (BaseSeedPolicy.__next__: coords[idx], idx += 1), else (None, None)."""
    policy = self.__dict__.get('seed_policy')
    if getattr(type(policy), '__next__', None) is not seed_lib.BaseSeedPolicy.__next__:
      return None, None
    return getattr(policy, 'coords', None), getattr(policy, 'idx', None)

  def _is_current_seed(self, pos) -> bool:
    coords, idx = self._policy_list()
    return (coords is not None and idx is not None and
            1 <= idx <= len(coords) and
            tuple(int(v) for v in coords[idx - 1]) == pos)

  def _upcoming_seeds(self, first=None):
    """The next seeds the policy will hand out (`first`: the one it has just
    handed out, then those), as long as they pass the bounds test."""
    coords, idx = self._policy_list()
    if coords is None or idx is None:
      return [] if first is None else [first]
    if first is not None:
      if not (1 <= idx <= len(coords) and
              tuple(int(v) for v in coords[idx - 1]) == first):
        return [first]
      idx -= 1
    out, listed = [], set()
    for c in np.asarray(coords[idx:idx + self.TURN_CANDIDATES]).tolist():
      c = (int(c[0]), int(c[1]), int(c[2]))
      # (a policy that lists a seed twice: the turn's record answers by
      # coordinate, so the list ends in front of the repeat)
      if not self._in_bounds(c) or c in listed:
        break
      listed.add(c)
      out.append(c)
    return out

  def _maybe_save_checkpoint(self, partial_segment_iters=0):
    if self.__dict__.get('_turn_rec') is not None:
      return  # (became due after the turn was made: taken at the next opportunity)
    super()._maybe_save_checkpoint(partial_segment_iters)

  def _turn(self, commit, mark, first=None):
    """One device-side turn; what it found out is kept for the seed loop's next
    questions (`_turn_rec`), which are then answered without a device call."""
    cands = self._upcoming_seeds(first)
    mbd = self.options.min_boundary_dist
    init = (float(self.options.init_activation)
            if self.reset_seed_per_segment else None)
    self._invalidate_cache()
    out = self._call(self._handle.segment_turn, commit, mark, cands,
                     (int(mbd.z), int(mbd.y), int(mbd.x)), init)
    chosen, flags, cseed, cseg = out[5], out[6], out[7], out[8]
    seen = {}
    for k, c in enumerate(cands):
      f = int(flags[k])
      if f == 3:  # after the chosen one: not looked at
        break
      seen[c] = f
      self._cache[c] = (float(cseed[k]), -1 if f == 2 else int(cseg[k]))
    # the -1 marker was written in mode 1 always, in mode 2 (commit) only when
    # the segment was NOT committed (ffn_canvas_segment_turn)
    marked = mark is not None and (mark[1] == 1 or not out[4])
    self._turn_rec = {
        'mark': tuple(int(v) for v in mark[0]) if marked else None,
        'seen': seen,
        'init': cands[chosen] if chosen >= 0 and init is not None else None}
    self._turn_armed = True
    self.turns += 1
    return out

  #: seeds whose (seed, segmentation) values one device call fetches ahead
  SEED_PREFETCH = 256

  def _read_point(self, pos):
    pos = (int(pos[0]), int(pos[1]), int(pos[2]))
    hit = self._cache.get(pos)
    if hit is not None:
      return hit
    # The seed loop tests one seed after the other (inference.py:573-581), and
    # late in a subvolume it rejects hundreds in a row (already segmented): one
    # device round trip each.  When `pos` is the seed the policy has just
    # handed out, the values of the NEXT seeds ride along; every write to the
    # canvas (a step, a commit) drops them again (`_invalidate_cache`), so
    # what the loop sees is what a read at that moment returns.
    policy = self.__dict__.get('seed_policy')
    coords = getattr(policy, 'coords', None)
    idx = getattr(policy, 'idx', 0)
    if (coords is not None and 1 <= idx <= len(coords) and
        self.SEED_PREFETCH > 1 and
        (int(coords[idx - 1][0]), int(coords[idx - 1][1]),
         int(coords[idx - 1][2])) == pos):
      batch = np.ascontiguousarray(coords[idx - 1:idx - 1 + self.SEED_PREFETCH],
                                   dtype=np.int32)
      seeds, segs = self._call(self._handle.read_points, batch)
      cache = self._cache
      for c, sv, gv in zip(batch.tolist(), seeds.tolist(), segs.tolist()):
        cache.setdefault((c[0], c[1], c[2]), (sv, gv))
      return cache[pos]
    val = self._call(self._handle.read_point, pos)
    self._cache[pos] = val
    return val

  def is_valid_pos(self, pos, ignore_move_threshold=False) -> bool:
    if not ignore_move_threshold:
      if self._read_point(pos)[0] < self.options.move_threshold:
        self.counters['skip_threshold'].Increment()
        if self._in_bounds(pos):
          self.gate_rejects += 1
        return False
    if not self._in_bounds(pos):
      self.counters['skip_invalid_pos'].Increment()
      return False
    if (ignore_move_threshold and self.__dict__.get('_turn_armed') and
        self._turn_ok()):
      # the seed loop (inference.py:573) asking about the seed the policy has
      # just handed out, after the last turn's answers have run out
      p = (int(pos[0]), int(pos[1]), int(pos[2]))
      if p not in self._cache and self._is_current_seed(p):
        self._turn(None, None, first=p)
    if self._read_point(pos)[1] > 0:
      self.counters['skip_invalid_pos'].Increment()
      self.gate_rejects += 1
      return False
    return True

  def _start_logit(self, start_pos):
    if self._cached_start is not None and self._cached_start[0] == start_pos:
      return self._cached_start[1]
    return self._read_point(start_pos)[0]

  # -- one FoV step: a single C call ---------------------------------------------------
  def _flush_hot(self):
    """Adds the per-step tallies kept in plain Python numbers to the counters."""
    h = self._hot
    if not h[0]:
      return
    c = self.counters
    c['update_at-calls'].IncrementBy(h[0])
    c['inference-calls'].IncrementBy(h[0])
    c['predict-calls'].IncrementBy(h[0])
    c['update_at-time-ms'].IncrementBy(h[1] * MSEC_IN_SEC)
    c['inference-time-ms'].IncrementBy(h[2] * MSEC_IN_SEC)
    c['inference-not-predict-ms'].IncrementBy(h[3] * MSEC_IN_SEC)
    if h[4]:
      c['movement_policy-calls'].IncrementBy(h[4])
      c['movement_policy-time-ms'].IncrementBy(h[5] * MSEC_IN_SEC)
    h[:] = [0, 0.0, 0.0, 0.0, 0, 0.0]  # in place: the segment loop holds a reference

  def _prepare_step(self, pos):
    """Fills the step request: FoV centre, segment start, queue-head points."""
    req = self._step_req
    if self._fast_policy:
      policy = self.movement_policy
      sp = policy._start_pos
      cands = policy.peek_candidates(self.PREFETCH)
    else:
      sp = pos
      cands = ()
    n = len(cands)
    flat = [pos[0], pos[1], pos[2], sp[0], sp[1], sp[2], n]
    for c in cands:
      flat += c
    self._req_i32[:7 + 3 * n] = flat  # one buffer write instead of 55 ctypes stores
    self._pending = (pos, sp, cands)
    return req

  def _finish_step(self, res):
    """Caches the post-step point values and wraps the face maxima."""
    pos, sp, cands = self._pending
    # one unpack of the whole record instead of ~50 ctypes attribute reads
    t = _STEP_RESULT.unpack_from(res)
    if self._keep_history:
      if self.options.disco_seed_threshold >= 0:
        self.history_deleted.append(t[53])
    self._cache = dict(zip(cands, zip(t[21:37], t[37:53])))
    self._cached_start = (sp, t[18])
    return movement.FacePrediction(t[0:6], t[6:12], t[12:18],
                                   self._pred_size_t,
                                   read_fn=self._make_reader(pos))

  def _blocking_step(self, req):
    return self._exec_client.step(self._handle, req, self._step_params)

  def _update_at_gen(self, pos):
    """One FoV step as a generator: yields the request, receives the result."""
    t_start = time.time()
    hot = self._hot
    req = self._prepare_step(pos)
    t_call = time.time()
    if self.t_last_predict is not None:
      hot[3] += t_call - self.t_last_predict
    res = yield req
    t_done = time.time()
    self.t_last_predict = t_done
    hot[2] += t_done - t_call
    pred = self._finish_step(res)
    hot[0] += 1
    hot[1] += time.time() - t_start
    return pred

  def update_at(self, pos):
    """gather -> conv stack -> disco -> paste -> face argmax on the GPU."""
    return self._drive(self._update_at_gen(pos))

  def segment_at(self, start_pos, dynamic_image=None, vis_update_every=10,
                 vis_fixed_z=False, partial_segment_iters=0):
    del dynamic_image, vis_update_every, vis_fixed_z
    return self._drive(self._segment_at_gen(start_pos, partial_segment_iters))

  # -- the whole segment loop inside the library ---------------------------------
  #: class-wide switch (FFN_AMD_NATIVE_LOOP=0 in the environment turns it off)
  NATIVE_LOOP = os.environ.get('FFN_AMD_NATIVE_LOOP', '1') != '0'

  def _native_loop_ok(self) -> bool:
    """True if `ffn_canvas_segment_at` may run this canvas' segment loop: the
    default movement policy and validity test, nothing hooked in between the
    steps, and an in-thread client (a loop on the executor's server thread
    would starve the other clients).

    Timed checkpoints (`checkpoint_interval`) are then taken at the first
    SEGMENT boundary after the interval has passed (segment_all checks after
    every segment) instead of after the first FoV step past it: a segment is
    seconds of GPU time against intervals of minutes (1800 s in the sample
    config), and a checkpoint taken between segments restores without a
    partial segment."""
    ok = self.__dict__.get('_native_ok')
    if ok is None:
      cls = type(self)
      ok = (cls.NATIVE_LOOP and
            hasattr(self._handle, 'segment_at') and
            getattr(self._exec_client, 'in_thread', False) and
            type(self.movement_policy) is movement.FaceMaxMovementPolicy and
            getattr(self.restrictor, 'is_trivial', self.restrictor is None) and
            cls._segment_at_gen is DeviceCanvas._segment_at_gen)
      self._native_ok = ok
    # hooks may also be set on the instance (canvas.update_at = ...)
    return (ok and
            getattr(self.update_at, '__func__', None) is DeviceCanvas.update_at
            and getattr(self.is_valid_pos, '__func__', None)
            is DeviceCanvas.is_valid_pos)

  def _segment_params(self):
    sp = self.__dict__.get('_seg_params')
    if sp is None:
      sp = _lib.SegmentParams()
      policy = self.movement_policy
      sp.score_threshold = float(policy.score_threshold)
      for a in range(3):
        sp.deltas_zyx[a] = int(policy.deltas[a])
        sp.margin_zyx[a] = int(self._margin_t[a])
        sp.shape_zyx[a] = int(self.shape[a])
      sp.prefetch = self.PREFETCH
      sp.keep_history = 1 if self._keep_history else 0
      self._seg_params = sp
    sp.step = self._step_params
    return sp

  def _native_prelude(self, start_pos, resume=False):
    """What Canvas.segment_at does before its loop; -> the loop's parameters, or
    None if the loop has to stay in Python (a pre-filled policy queue)."""
    sp = self._segment_params()
    if not resume:
      if self.reset_seed_per_segment:
        self.init_seed(start_pos)
      self.reset_state(start_pos, reset_extents=self.reset_seed_per_segment)
      if self.movement_policy or self._min_pos is None:
        return None  # a pre-filled queue is the caller's business
      for a in range(3):
        sp.init_min_pos[a] = int(self._min_pos[a])
        sp.init_max_pos[a] = int(self._max_pos[a])
      cs = self._cached_start
      sp.initial_start_logit = (cs[1] if cs is not None and cs[0] == start_pos
                                else float('nan'))
    return sp

  def _segment_at_native(self, start_pos, max_steps=0, resume=False):
    """Canvas.segment_at through `ffn_canvas_segment_at`; same state and
    counters afterwards as the Python loop leaves."""
    start_pos = tuple(int(v) for v in start_pos)
    sp = self._native_prelude(start_pos, resume)
    if sp is None:
      return self._drive(self._segment_at_gen_body(start_pos))
    sp.max_steps = int(max_steps)
    t0 = time.time()
    # a resumed leg belongs to the loop call that started the segment
    with timer_counter(self.counters, 'segment_at-loop',
                       increment=0 if resume else 1):
      self._invalidate_cache()
      res = self._call(self._handle.segment_at, start_pos, sp, resume)
    return self._native_postlude(start_pos, res, time.time() - t0)

  def _segment_at_native_many(self, start_pos):
    """The same for a canvas advanced by `MultiCanvasDriver` in native mode: the
    loop runs inside `ffn_canvas_segment_many` next to the other canvases'; the
    driver sends back the result of the WHOLE segment."""
    sp = self._native_prelude(start_pos)
    if sp is None:
      return (yield from self._segment_at_gen_body(start_pos))
    sp.max_steps = 0
    self._invalidate_cache()
    t0 = time.time()
    # a copy: the driver keeps it across calls, _segment_params() is reused
    res = yield NativeSegment(start_pos, _lib.SegmentParams.from_buffer_copy(sp))
    dt = time.time() - t0
    self.counters['segment_at-loop-calls'].Increment()
    self.counters['segment_at-loop-time-ms'].IncrementBy(dt * MSEC_IN_SEC)
    self._invalidate_cache()
    return self._native_postlude(start_pos, res, dt)

  def _native_postlude(self, start_pos, res, dt):
    n = int(res.num_steps)
    c = self.counters
    if n:
      c['update_at-calls'].IncrementBy(n)
      c['inference-calls'].IncrementBy(n)
      c['predict-calls'].IncrementBy(n)
      c['movement_policy-calls'].IncrementBy(n)
      c['update_at-time-ms'].IncrementBy(dt * MSEC_IN_SEC)
      c['inference-time-ms'].IncrementBy(dt * MSEC_IN_SEC)
    if res.skip_threshold:
      c['skip_threshold'].IncrementBy(int(res.skip_threshold))
    if res.skip_invalid_pos:
      c['skip_invalid_pos'].IncrementBy(int(res.skip_invalid_pos))
    if res.seed_got_too_weak:
      c['seed_got_too_weak'].Increment()
    self.gate_rejects += int(res.gate_rejects)
    self._min_pos = np.array([int(v) for v in res.min_pos])
    self._max_pos = np.array([int(v) for v in res.max_pos])
    if res.start_logit_known:
      self._cached_start = (start_pos, float(res.start_logit))
    self.t_last_predict = time.time()
    if self._keep_history and not res.budget_exhausted:
      pos, deleted = self._call(self._handle.segment_history)
      self.history = [tuple(int(v) for v in p) for p in pos]
      if self.options.disco_seed_threshold >= 0:
        self.history_deleted = [int(v) for v in deleted]
    self._native_active = bool(res.active)
    return n

  def _segment_at_gen(self, start_pos, partial_segment_iters=0):
    """Same loop as Canvas.segment_at (reference inference.py:460-533), with
    the per-step tallies kept in plain Python numbers."""
    start_pos = tuple(int(v) for v in start_pos)
    if (not partial_segment_iters and
        self.__dict__.get('_blocking_drive', False) and
        self._native_loop_ok()):
      # driven by blocking calls (segment_at / segment_all, not interleaved by
      # a MultiCanvasDriver): the whole loop runs inside the library
      return self._segment_at_native(start_pos)
    if (not partial_segment_iters and
        self.__dict__.get('_native_many', False) and self._native_loop_ok()):
      # interleaved by a MultiCanvasDriver in native mode
      return (yield from self._segment_at_native_many(start_pos))
    if not partial_segment_iters:
      if self.reset_seed_per_segment:
        self.init_seed(start_pos)
      self.reset_state(start_pos, reset_extents=self.reset_seed_per_segment)
    return (yield from self._segment_at_gen_body(start_pos,
                                                 partial_segment_iters))

  def _segment_at_gen_body(self, start_pos, partial_segment_iters=0):
    """The loop proper, after init_seed / reset_state."""
    if not partial_segment_iters:
      if not self.movement_policy:
        self.movement_policy.append(
            (self.movement_policy.score_threshold * 2, start_pos))
    num_iters = partial_segment_iters
    mn = [int(v) for v in self._min_pos]
    mx = [int(v) for v in self._max_pos]
    thr = self.options.move_threshold
    restrict = (None if getattr(self.restrictor, 'is_trivial', False) else
                self.restrictor)
    hot = self._hot
    checkpointing = (self.checkpoint_path is not None and
                     self.checkpoint_interval_sec > 0)
    # The reference's loop calls self.update_at(pos): a subclass that overrides
    # it keeps being called (such a canvas then cannot be interleaved by the
    # MultiCanvasDriver, it simply blocks for its steps).
    overridden = (getattr(self.update_at, '__func__', None)
                  is not DeviceCanvas.update_at)
    with timer_counter(self.counters, 'segment_at-loop'):
      try:
        for pos in self.movement_policy:
          if self._start_logit(start_pos) < thr:
            self.counters['seed_got_too_weak'].Increment()
            break
          if restrict is not None and not restrict.is_valid_pos(pos):
            self.counters['skip_restriced_pos'].Increment()
            continue
          if overridden:
            pred = self.update_at(pos)  # honour subclass hooks (blocking)
          else:
            pred = yield from self._update_at_gen(pos)
          for a in (0, 1, 2):
            if pos[a] < mn[a]:
              mn[a] = pos[a]
            if pos[a] > mx[a]:
              mx[a] = pos[a]
          num_iters += 1
          if self._keep_history:
            self.history.append(pos)
          t0 = time.time()
          self._policy_update(pred, pos)
          hot[4] += 1
          hot[5] += time.time() - t0
          if checkpointing:
            self._min_pos = np.array(mn)
            self._max_pos = np.array(mx)
            self._flush_hot()
            self._maybe_save_checkpoint(partial_segment_iters=num_iters)
      finally:
        self._min_pos = np.array(mn)
        self._max_pos = np.array(mx)
        self._flush_hot()
    return num_iters

  def _make_reader(self, pos):
    def read():
      # the box update_at wrote (inference.py:410-411): pred_mask_size, centred
      lo = tuple(int(pos[a] - self._input_seed_size[a] // 2 + self._pred_delta[a])
                 for a in range(3))
      hi = tuple(int(l + self._pred_size[a]) for a, l in enumerate(lo))
      return self._call(self._handle.read_seed, lo, hi)
    return read

  def _policy_update(self, pred, pos):
    new = self.movement_policy.update(pred, pos)
    if new:
      # Freshly queued moves: their seed logit is the face maximum just written
      # by the paste kernel, and the kernel also returned segmentation[] there.
      for coord, score, seg in new:
        self._cache.setdefault(coord, (score, seg))

  def init_seed(self, pos):
    rec = self.__dict__.get('_turn_rec')
    served = (rec is not None and rec['init'] is not None and
              rec['init'] == tuple(int(v) for v in pos))
    self._invalidate_cache()
    if not served:  # (else: the turn that chose this seed has done it)
      self._call(self._handle.init_seed, pos, self.options.init_activation)
    self._cached_start = (tuple(int(v) for v in pos),
                          float(np.float32(self.options.init_activation)))

  # -- segment bookkeeping on the device -------------------------------------------------
  def _seg_point(self, pos) -> int:
    return int(self._read_point(pos)[1])

  def _mark_excluded(self, pos):
    rec = self.__dict__.get('_turn_rec')
    p = tuple(int(v) for v in pos)
    if rec is not None and rec['mark'] == p:
      rec['mark'] = None  # the turn that counted this segment has marked it
      return
    if self._turn_ok():
      self._turn(None, (p, 1))
      self._turn_rec['mark'] = None
      return
    if self._seg_point(pos) == 0:
      self._invalidate_cache()
      self._call(self._handle.write_seg_points, [pos], [-1])

  def _too_close(self, pos, mbd) -> bool:
    rec = self.__dict__.get('_turn_rec')
    if rec is not None:
      f = rec['seen'].pop(tuple(int(v) for v in pos), None)
      if f is not None:  # tested (and, if too close, marked) by the last turn
        return f == 2
    low = [int(p - m) for p, m in zip(pos, mbd)]
    high = [int(p + m + 1) for p, m in zip(pos, mbd)]
    if self._call(self._handle.any_segmented, low, high):
      self._invalidate_cache()
      self._call(self._handle.write_seg_points, [pos], [-1])
      return True
    return False

  def _commit(self, sel_lo, sel_hi, pos):
    thr = self.options.segment_threshold
    max_existing = max(self._max_id, max(self.origins) if self.origins else 0)
    if self._turn_ok():
      sid = self._max_id + 1  # what get_next_segment_id will hand out
      while sid in self.origins:
        sid += 1
      out = self._turn(([int(v) for v in sel_lo], [int(v) for v in sel_hi], thr,
                        int(self.options.min_segment_size), sid, max_existing),
                       (pos, 2))
      raw, actual, ids, counts, committed = out[:5]
      if not committed:
        return raw, actual, ids, counts, None
      got = self.get_next_segment_id()
      assert got == sid, (got, sid)
      return raw, actual, ids, counts, sid
    raw, actual, ids, counts = self._call(self._handle.commit_count, sel_lo,
                                          sel_hi, thr, max_existing)
    if actual < self.options.min_segment_size:
      return raw, actual, ids, counts, None
    sid = self.get_next_segment_id()
    if self.keep_probability_maps:
      sel = tuple(slice(l, h) for l, h in zip(sel_lo, sel_hi))
      seed = self._call(self._handle.read_seed, sel_lo, sel_hi)
      seg = self._call(self._handle.read_segmentation, sel_lo, sel_hi)
      mask = (seed >= thr) & (seg <= 0)
      self.seg_prob[sel][mask] = storage.quantize_probability(
          expit(seed[mask]))
    self._invalidate_cache()
    self._call(self._handle.commit_assign, sel_lo, sel_hi, thr, sid)
    return raw, actual, ids, counts, sid

  def _set_segmentation(self, seg):
    self.segmentation[...] = np.asarray(seg, np.int32)

  def _set_seed(self, seed):
    self.seed[...] = np.asarray(seed, np.float32)
    # a restored canvas may be inside a segment (partial_segment_iters): no turn
    # -- it would re-initialise the seed -- before that segment has ended
    self._turn_armed = False


class NativeSegment:
  """What a DeviceCanvas yields instead of single FoV steps when its driver
  runs whole segment loops in the library (`ffn_canvas_segment_many`): "run the
  segment at `start_pos`"; the driver answers with that segment's
  `SegmentResult` once the loop has ended."""

  __slots__ = ('start_pos', 'params', 'started')

  def __init__(self, start_pos, params):
    self.start_pos = start_pos
    self.params = params
    self.started = False


class MultiCanvasDriver:
  """Single-threaded scheduler advancing many DeviceCanvases: every round
  collects the pending FoV-step requests of up to `batch_size` canvases and
  issues ONE batched step for them; each result is fed back into its canvas'
  generator.

  This is what the reference gets from N client threads + 1 server thread
  (executor.py:266-340), without the per-step queue hops and without GIL
  contention between the client threads (measured: with 64 client threads the
  threaded path drops to 1.7k steps/s; see profiles/).

  With `overlap` (default) the live canvases are split into two groups and two
  steps are kept in flight (`ffn_canvas_step_submit` / `_wait`): while the GPU
  runs one group's step, Python digests the other group's results and queues
  its next step, so the GPU never waits for the host.
  """

  def __init__(self, engine, batch_size=None, overlap=True,
               max_steps_per_canvas=None, native=None, groups=1, carry=None):
    """groups (native mode): 2 = the open canvases form two groups, each driven
    by its own host thread and its own library calls, `batch_size` canvases per
    call.  The library interleaves the two calls' steps on the engine's stream
    (one step of each in flight), so while one thread does a canvas'
    between-segment work in Python -- commit, seed policy, the next
    `init_seed` -- the other group's steps keep the GPU busy; with one group
    every ended segment idles it for that long.  What the reference's N client
    threads give it (executor.py:266-340), with two threads.

    native: run whole segment loops inside the library
    (`ffn_canvas_segment_many`: the per-canvas policy queues, validity tests and
    the batching in C++, Python only between segments) instead of one Python
    round trip per batched step.  Default: on when the engine has it
    (FFN_AMD_NATIVE_MANY=0 in the environment turns it off); canvases that
    cannot use the library's loop (custom policies, restrictors, hooks) are
    stepped one by one next to it."""
    self.engine = engine
    hip_engine.pin_batched_arithmetic(engine)
    self.batch_size = batch_size or engine.max_batch
    self.overlap = overlap
    if native is None:
      native = os.environ.get('FFN_AMD_NATIVE_MANY', '1') != '0'
    self.native = bool(native) and hasattr(engine, 'segment_many')
    self.groups = max(1, int(groups)) if self.native else 1
    #: one group: leave the running canvases' next step in flight while a canvas
    #: is between two segments (`ffn_canvas_segment_many_carry`)
    if carry is None:
      carry = os.environ.get('FFN_AMD_MANY_CARRY', '1') != '0'
    self.carry = bool(carry)
    #: benchmarking / bounded runs: a canvas is dropped after this many steps
    self.max_steps_per_canvas = max_steps_per_canvas
    self.calls = 0
    self.steps = 0
    #: native mode: seconds the group threads spent inside the library's
    #: segment_many calls, and segments ended (= between-segment turns in Python)
    self.library_seconds = 0.0
    self.segments_ended = 0

  def run(self, jobs, window=None, on_done=None):
    """jobs: iterable of (DeviceCanvas, seed_policy_factory) -- a whole
    `segment_all` per canvas -- or (DeviceCanvas, generator) for any other
    step-yielding task on that canvas (e.g. a resegmentation point, or a
    `segment_all` resumed from a checkpoint).

    `jobs` is consumed LAZILY: at most `window` canvases are open at any time
    (None = all at once); when one finishes, `on_done(canvas)` is called -- the
    place to save and close it -- and the next job is pulled.  A device canvas
    holds 12 B / voxel of HBM plus its host image, so a long job list must not
    be materialised up front."""
    if self.native:
      return self._run_native(jobs, window, on_done)
    jobs = iter(jobs)
    ready = collections.deque()  # [canvas, generator, pending request, steps]
    state = {'open': 0, 'exhausted': False}

    def finished(canvas):
      state['open'] -= 1
      if on_done is not None:
        on_done(canvas)

    def refill():
      while not state['exhausted'] and (window is None or
                                        state['open'] < window):
        try:
          canvas, task = next(jobs)
        except StopIteration:
          state['exhausted'] = True
          return
        gen = (task if inspect.isgenerator(task) else
               canvas._segment_all_gen(task))
        state['open'] += 1
        try:
          ready.append([canvas, gen, next(gen), 0])
        except StopIteration:
          finished(canvas)

    refill()
    limit = self.max_steps_per_canvas
    inflight = collections.deque()  # (ticket, batch)
    depth = 2 if self.overlap else 1
    engine = self.engine
    while ready or inflight:
      while ready and len(inflight) < depth:
        if self.overlap:
          live = len(ready) + sum(len(b) for _, b in inflight)
          n = min(self.batch_size, max(1, (live + 1) // 2), len(ready))
        else:
          n = min(self.batch_size, len(ready))
        # one engine call = one set of step parameters: canvases that differ
        # (keep_history / other options, e.g. resegmentation canvases next to
        # plain ones) wait for a batch of their own kind
        key = bytes(ready[0][0]._step_params)
        batch, others = [], []
        while ready and len(batch) < n:
          entry = ready.popleft()
          if bytes(entry[0]._step_params) == key:
            batch.append(entry)
          else:
            others.append(entry)
        ready.extendleft(reversed(others))
        n = len(batch)
        ticket = engine.step_submit([b[0]._handle for b in batch],
                                    [b[2] for b in batch],
                                    batch[0][0]._step_params)
        inflight.append((ticket, batch))
        self.calls += 1
        self.steps += n
      ticket, batch = inflight.popleft()
      res = engine.step_wait(ticket)
      for k, entry in enumerate(batch):
        try:
          entry[2] = entry[1].send(res[k])
        except StopIteration:
          finished(entry[0])
          continue
        entry[3] += 1
        if limit is not None and entry[3] >= limit:
          entry[1].close()
          finished(entry[0])
          continue
        ready.append(entry)
      refill()


  def _run_native(self, jobs, window, on_done):
    """`run` with the segment loops in the library: every engine call advances
    the current segments of up to `batch_size` canvases until one of them ends;
    that canvas' generator then does its between-segment work (commit, next
    seed) and yields its next segment."""
    jobs = iter(jobs)
    if window is None or window > self.batch_size:
      # a canvas outside the engine call would only hold memory
      window = self.batch_size
    lock = threading.Lock()
    errors = []  # of the group threads: a failed group ends the deal for all

    def pull():
      """Next (canvas, task) of the job list, or None."""
      with lock:
        if errors:
          return None
        try:
          return next(jobs)
        except StopIteration:
          return None

    def done(canvas):
      if on_done is not None:
        with lock:
          on_done(canvas)

    if self.groups == 1:
      tally = [0, 0, 0.0, 0]
      try:
        self._run_native_group(pull, window, done, tally, errors)
      finally:
        self.calls += tally[0]
        self.steps += tally[1]
        self.library_seconds += tally[2]
        self.segments_ended += tally[3]
      return
    # one thread per group; the GIL is released inside the library calls, and a
    # short switch interval hands it over promptly when one returns (the
    # interval is PROCESS-wide: other Python threads of the caller switch more
    # often while this runs; it is restored on the way out)
    tallies = [[0, 0, 0.0, 0] for _ in range(self.groups)]

    def work(k):
      try:
        self._run_native_group(pull, window, done, tallies[k], errors)
      except BaseException as e:  # pylint:disable=broad-except
        errors.append(e)  # (the other groups see it at their next round)

    interval = sys.getswitchinterval()
    sys.setswitchinterval(min(interval, 2e-4))
    threads = [threading.Thread(target=work, args=(k,), daemon=True,
                                name='ffn-canvas-group-%d' % k)
               for k in range(self.groups)]
    try:
      for t in threads:
        t.start()
      for t in threads:
        t.join()
    finally:
      sys.setswitchinterval(interval)
      for c, n, sec, ended in tallies:
        self.calls += c
        self.steps += n
        self.library_seconds += sec
        self.segments_ended += ended
    if errors:
      raise errors[0]

  def _run_native_group(self, pull, window, on_done, tally, errors=()):
    """One group of at most `window` open canvases: the loop of `_run_native`.
    tally = [engine calls, FoV steps, seconds inside segment_many, segments
    ended] of this group.  `errors`: the job's list of failures -- once another
    group has put one there this group stops at its next round; whichever way
    the group ends early, its open canvases are closed (generator and HBM)."""
    live = []
    try:
      self._native_group_loop(pull, window, on_done, tally, errors, live)
    finally:
      for entry in live:  # empty after a complete run
        try:
          entry[1].close()
        except Exception:  # pylint:disable=broad-except
          pass
        try:
          entry[0].close()
        except Exception:  # pylint:disable=broad-except
          pass

  def _native_group_loop(self, pull, window, on_done, tally, errors, live):
    engine = self.engine
    limit = self.max_steps_per_canvas
    # live: [canvas, generator, pending request, steps of finished segments,
    #        steps of the current segment reported so far]
    state = {'exhausted': False}

    def finished(entry):
      live.remove(entry)
      on_done(entry[0])

    def advance(entry, value):
      """Feeds `value` to the canvas' generator; False once it is through."""
      try:
        entry[2] = entry[1].send(value) if value is not None else next(entry[1])
        return True
      except StopIteration:
        finished(entry)
        return False

    def refill():
      while not state['exhausted'] and len(live) < window:
        job = pull()
        if job is None:
          state['exhausted'] = True
          return
        canvas, task = job
        canvas._native_many = True
        gen = (task if inspect.isgenerator(task) else
               canvas._segment_all_gen(task))
        entry = [canvas, gen, None, 0, 0]
        live.append(entry)
        advance(entry, None)

    refill()
    #: steps a library call may make while canvases on the Python loop wait
    mixed_call_steps = 8
    while live:
      if errors:  # another group failed: the job is over
        return
      # canvases on the Python loop (restrictor masks, custom policies, hooks:
      # no native segment pending) advance by ONE batched step per round, next
      # to the library calls of the others -- never one canvas at a time
      py = [e for e in live if not isinstance(e[2], NativeSegment)]
      while py:
        key = bytes(py[0][0]._step_params)
        batch = [e for e in py if bytes(e[0]._step_params) == key]
        batch = batch[:self.batch_size]
        py = [e for e in py if e not in batch]
        res = engine.step([e[0]._handle for e in batch], [e[2] for e in batch],
                          batch[0][0]._step_params)
        tally[0] += 1
        tally[1] += len(batch)
        for entry, r in zip(batch, res):
          entry[3] += 1
          if limit is not None and entry[3] >= limit:
            entry[1].close()
            finished(entry)
            continue
          advance(entry, r)
      nat = [e for e in live if isinstance(e[2], NativeSegment)]
      if not nat:
        refill()
        continue
      mixed = len(nat) < len(live)
      key = bytes(nat[0][2].params.step)
      batch = [e for e in nat if bytes(e[2].params.step) == key][:self.batch_size]
      for e in batch:
        left = limit - e[3] - e[4] if limit is not None else 0
        if mixed:
          left = min(left, mixed_call_steps) if limit is not None else mixed_call_steps
        e[2].params.max_steps = max(left, 1) if (limit is not None or mixed) else 0
      t_call = time.perf_counter()
      # one group: the others' next step stays in flight while this thread does
      # the between-segment work of the canvas whose loop has just ended (what a
      # second group's steps provide otherwise); step budgets are per call and
      # canvases on the Python loop need the engine's step slots: neither then
      kw = {}
      if (self.carry and self.groups == 1 and limit is None and not mixed and
          len(batch) == len(nat) and getattr(engine, 'can_carry', False)):
        kw['carry'] = True
      results, fin = engine.segment_many(
          [e[0]._handle for e in batch], [e[2].start_pos for e in batch],
          [e[2].params for e in batch], [e[2].started for e in batch], **kw)
      tally[2] += time.perf_counter() - t_call
      tally[3] += sum(1 for d in fin if d)
      tally[0] += 1
      for e, res, done in zip(batch, results, fin):
        tally[1] += int(res.num_steps) - e[4]
        e[4] = int(res.num_steps)
        e[2].started = True
        if not done:
          continue
        if res.budget_exhausted:
          if limit is not None and e[3] + e[4] >= limit:
            e[1].close()
            finished(e)
          # else: only this call's share was spent; the loop is resumed
          continue
        e[3] += e[4]
        e[4] = 0
        if limit is not None and e[3] >= limit:
          e[1].close()
          finished(e)
          continue
        advance(e, res)
      refill()


def make_canvas(model_info, exec_client, image, options, **kwargs) -> Canvas:
  """DeviceCanvas if the client can host device canvases, else host Canvas."""
  if hasattr(exec_client, 'create_canvas') and hasattr(exec_client, 'step'):
    return DeviceCanvas(model_info, exec_client, image, options, **kwargs)
  return Canvas(model_info, exec_client, image, options, **kwargs)
