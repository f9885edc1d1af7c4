"""Python handles over the C-ABI objects (ffn_engine / ffn_canvas).

Thin, allocation-free wrappers: every method is one ctypes call into
libffn_hip.so.  torch is not involved -- device memory belongs to the library.
"""

from __future__ import annotations

import atexit
import ctypes
import threading
from typing import Sequence
import weakref

import numpy as np

from . import _lib
from ._lib import (CommitCounts, StepParams, StepRequest, StepResult, check, i3)


def _f32(a):
  return np.ascontiguousarray(a, dtype=np.float32)


_LIVE_ENGINES = weakref.WeakSet()


@atexit.register
def _close_all_engines():
  # Release device objects while the HIP runtime is still alive, instead of
  # leaving it to arbitrary-order __del__ calls at interpreter teardown.
  for eng in list(_LIVE_ENGINES):
    try:
      eng.close()
    except Exception:  # pylint:disable=broad-except
      pass


def pin_batched_arithmetic(engine):
  """Called by the drivers that put several canvases into one engine call.

  The default kernel choice (conv_variant 9) runs a step of ONE FoV as conv32mt
  and a step of several as conv32m -- the same sums in another order for the
  last voxels of the FoV, ~1e-6 in the logits.  In a batched run the number of
  FoVs per call varies with timing, so a canvas would get either, step by
  step; conv32m for every step keeps a batched run reproducible.  An explicit
  choice -- Runner(conv_variant=...), --conv_variant, engine.set_option(
  'conv_variant', ...) -- wins whenever it was made: the drivers call this on
  every run, and it leaves an engine alone whose kernel somebody chose."""
  if not isinstance(engine, HipEngine):
    return  # a stand-in engine of the host-logic tests
  if engine.variant_is_explicit:
    return
  if engine.max_batch > 1 and engine.get_option('conv_variant') == 9:
    engine.set_option('conv_variant', 8, explicit=False)


#: how often one step may come back void before the error is raised: the fp16
#: range check voids at most once per engine (it then stays on the exact kernel),
#: the resident launch three times in a row (the library then turns it off)
MAX_VOID_REPEATS = 5


def segment_many_with_retry(once, n, sarr, parr, rarr, before, fallback):
  """One `ffn_canvas_segment_many` call plus the handling of a round voided by
  the fp16 range check (FFN_ERR_RANGE), shared by HipEngine and the CPU shim of
  the tests.  once(keys, starts, params, resumes, results, finished) -> rc makes
  the call for the canvases `keys` (indices into the caller's list);
  fallback(rc) is told which kind of void it was (FFN_ERR_RANGE: switch to the
  exact-f32 kernel; FFN_ERR_FLOW: nothing to switch, the library already runs
  the repeat without the resident launch);
  before[k] = steps canvas k's segment had made before this call (results count
  from the start of the segment, budgets are per call).

  The voided round changed nothing on the device and every prepared position
  stays pending.  Loops that ENDED in this call before the voided round (queue
  empty, seed too weak, budget spent) keep their result -- they are no longer
  resumable, and asking the library to resume them is FFN_ERR_STATE.  The others
  are resumed, after `fallback()` has switched kernels, with what is left of
  their budgets."""
  res = (_lib.SegmentResult * n)()
  fin = (ctypes.c_int32 * n)()
  keys = list(range(n))
  sa, pa, ra, cur_res, cur_fin = sarr, parr, rarr, res, fin
  for attempt in range(MAX_VOID_REPEATS + 1):
    rc = once(keys, sa, pa, ra, cur_res, cur_fin)
    if cur_res is not res:
      for j, k in enumerate(keys):
        ctypes.pointer(res[k])[0] = cur_res[j]
        fin[k] = cur_fin[j]
    if rc not in _lib.ERR_VOIDED or attempt == MAX_VOID_REPEATS:
      return rc, res, fin
    fallback(rc)
    keys = [k for k in keys if not fin[k]]
    if not keys:
      return 0, res, fin
    m = len(keys)
    sa = (ctypes.c_int32 * 3 * m)()
    pa = (_lib.SegmentParams * m)()
    ra = (ctypes.c_int32 * m)(*([1] * m))
    cur_res = (_lib.SegmentResult * m)()
    cur_fin = (ctypes.c_int32 * m)()
    for j, k in enumerate(keys):
      for a in range(3):
        sa[j][a] = sarr[k][a]
      ctypes.pointer(pa[j])[0] = parr[k]
      if parr[k].max_steps > 0:
        spent = int(res[k].num_steps) - before[k]
        pa[j].max_steps = max(parr[k].max_steps - spent, 1)
  return rc, res, fin


class HipEngine:
  """One GPU + one stream + the conv-stack weights (include/ffn_hip.h)."""

  def __init__(self, fov_zyx, deltas_zyx, depth: int, features: int = 32,
               max_batch: int = 1, device_id: int = 0):
    self._lib = _lib.load()
    self._h = ctypes.c_void_p()
    self.fov_zyx = tuple(int(v) for v in fov_zyx)
    self.deltas_zyx = tuple(int(v) for v in deltas_zyx)
    self.depth = int(depth)
    self.features = int(features)
    self.max_batch = int(max_batch)
    self.device_id = int(device_id)
    check(self._lib.ffn_engine_create(self.device_id, i3(self.fov_zyx),
                                      i3(self.deltas_zyx), self.depth,
                                      self.features, self.max_batch,
                                      ctypes.byref(self._h)))
    self._canvas_arr = (ctypes.c_void_p * self.max_batch)()
    self._req_arr = (StepRequest * self.max_batch)()
    self._res_arr = (StepResult * self.max_batch)()
    # per-slot argument / result arrays of the split submit / wait calls
    self._slot_canvas_arr = [(ctypes.c_void_p * self.max_batch)()
                             for _ in range(2)]
    self._slot_req_arr = [(StepRequest * self.max_batch)() for _ in range(2)]
    self._slot_res_arr = [(StepResult * self.max_batch)() for _ in range(2)]
    self._submit_slot = 0
    self._ticket_slot = {}
    # `step` / `step1` share argument arrays: one caller at a time (the library
    # itself is thread safe; `segment_many` brings its own arrays)
    self._step_lock = threading.RLock()
    #: steps repeated because a split-product kernel met a value outside the fp16 range
    self.range_fallbacks = 0
    #: steps repeated because the resident conv launch timed out (FFN_ERR_FLOW)
    self.flow_fallbacks = 0
    #: True once a caller has chosen the conv kernel (pin_batched_arithmetic
    #: then keeps its hands off)
    self.variant_is_explicit = False
    self._default_variant = self.get_option('conv_variant')
    #: the model's prediction box inside the FoV (set_pred_size), zyx slices
    self.pred_zyx = self.fov_zyx
    self._pred_sel = None
    self._canvases = weakref.WeakSet()
    _LIVE_ENGINES.add(self)

  @classmethod
  def from_model(cls, model, max_batch: int = 1, device_id: int = 0):
    """Builds an engine from a ConvStack3DFFNModel (xyz geometry -> zyx)."""
    info = model.info
    if not np.array_equal(info.input_seed_size, info.input_image_size):
      raise ValueError('seed and image sizes must be equal for the conv stack')
    if np.any(np.asarray(info.pred_mask_size) > np.asarray(info.input_seed_size)):
      raise ValueError('pred_mask_size exceeds input_seed_size')
    if np.any((np.asarray(info.input_seed_size) - np.asarray(info.pred_mask_size)) % 2):
      # the reference's update_at (inference.py:218,410-411) would build a box of
      # pred + 1 voxels and fail on the shape: not a geometry it can run
      raise ValueError('input_seed_size - pred_mask_size must be even on every axis: '
                       'the prediction is centred in the seed FoV')
    eng = cls(tuple(int(v) for v in info.input_seed_size[::-1]),
              tuple(int(v) for v in info.deltas[::-1]), model.depth,
              model.features, max_batch, device_id)
    eng.set_weights(model.weights_blob())
    if not np.array_equal(info.pred_mask_size, info.input_seed_size):
      eng.set_pred_size(tuple(int(v) for v in info.pred_mask_size[::-1]))
    return eng

  def close(self):
    if self._h:
      for c in list(self._canvases):
        c.close()
      self._lib.ffn_engine_destroy(self._h)
      self._h = ctypes.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint:disable=broad-except
      pass

  # -- setup -----------------------------------------------------------------
  def set_weights(self, blob: np.ndarray):
    blob = _f32(blob).ravel()
    check(self._lib.ffn_engine_set_weights(self._h, blob.ctypes.data,
                                           blob.size))

  def set_pred_size(self, pred_zyx):
    """A model that predicts a smaller mask than the seed it reads (ModelInfo
    pred_mask_size < input_seed_size, reference model.py:168-183): the centred
    box of the FoV that canvas steps score and paste, and that `predict`
    returns (ffn_engine_set_pred_size)."""
    pred_zyx = tuple(int(v) for v in pred_zyx)
    check(self._lib.ffn_engine_set_pred_size(self._h, i3(pred_zyx)))
    self.pred_zyx = pred_zyx
    if pred_zyx == self.fov_zyx:
      self._pred_sel = None
    else:
      lo = [(f - p) // 2 for f, p in zip(self.fov_zyx, pred_zyx)]
      self._pred_sel = (slice(None),) + tuple(
          slice(l, l + p) for l, p in zip(lo, pred_zyx))

  def get_option(self, name: str) -> int:
    value = ctypes.c_int(0)
    check(self._lib.ffn_engine_get_option(self._h, name.encode(),
                                          ctypes.byref(value)))
    return value.value

  def set_option(self, name: str, value: int, explicit: bool = True):
    """ffn_engine_set_option.  conv_variant: 0 / 2 (exact f32), 6 .. 9, or -1 =
    this FoV's exact-f32 kernel; `explicit=False` is for the library's own
    choices (the batched drivers' pin, the fp16-range fallback)."""
    check(self._lib.ffn_engine_set_option(self._h, name.encode(), int(value)))
    if name == 'conv_variant' and explicit and int(value) != -1:
      self.variant_is_explicit = True

  def restore_default_variant(self):
    """Back to the kernel choice the engine was created with, as if nobody had
    chosen one."""
    self.set_option('conv_variant', self._default_variant, explicit=False)
    self.variant_is_explicit = False

  def set_profiling(self, mode: int):
    check(self._lib.ffn_engine_set_profiling(self._h, int(mode)))

  def get_profile(self, reset: bool = False):
    ms = ctypes.c_double()
    n = ctypes.c_int64()
    check(self._lib.ffn_engine_get_profile(self._h, ctypes.byref(ms),
                                           ctypes.byref(n), int(reset)))
    return ms.value, n.value

  def get_profile_samples(self, max_n: int = 65536) -> np.ndarray:
    """ms of every sampled event pair since the last get_profile(reset=True)."""
    out = np.zeros(max_n, np.float32)
    n = ctypes.c_int()
    check(self._lib.ffn_engine_get_profile_samples(self._h, out.ctypes.data, max_n,
                                                   ctypes.byref(n)))
    return out[:min(n.value, max_n)].copy()

  def debug_clocks(self):
    out = np.zeros(24, np.int64)
    check(self._lib.ffn_engine_debug_clocks(self._h, out.ctypes.data))
    return out.reshape(4, 6)

  def debug_workgroups(self, n: int = 4096):
    """[n, 4]: wall clock (100 MHz) at entry / exit, HW_ID, XCC_ID per block of
    the launch stamped under debug_clock 2 (ffn_engine_debug_workgroups)."""
    out = np.zeros((n, 4), np.int64)
    check(self._lib.ffn_engine_debug_workgroups(self._h, out.ctypes.data, n))
    return out

  def debug_flow_trace(self, slots: int):
    """[slots, 64, 8] wall-clock stamps (100 MHz) of the FLOW convs run under
    debug_clock 4 (ffn_engine_debug_flow_trace): per workgroup slot and conv of
    the stack {entry, tiles seen, first segment landed, taps over, stores
    drained, published}."""
    out = np.zeros((slots, 64, 8), np.int64)
    check(self._lib.ffn_engine_debug_flow_trace(self._h, out.ctypes.data, slots))
    return out

  def synchronize(self):
    check(self._lib.ffn_engine_synchronize(self._h))

  # -- stateless predict (ExecutorClient.predict) ------------------------------
  def predict(self, seed: np.ndarray, image: np.ndarray) -> np.ndarray:
    """seed, image: [n, z, y, x] f32 (input_seed_size) -> logits [n, z', y', x']
    f32 (pred_mask_size: the same unless set_pred_size made it smaller)."""
    seed = _f32(seed)
    image = _f32(image)
    if seed.shape != image.shape or seed.shape[1:] != self.fov_zyx:
      raise ValueError('predict expects [n,%d,%d,%d] arrays, got %r / %r' %
                       (self.fov_zyx + (seed.shape, image.shape)))
    out = np.empty_like(seed)
    check(self._lib.ffn_predict(self._h, seed.shape[0], seed.ctypes.data,
                                image.ctypes.data, out.ctypes.data))
    if self._pred_sel is not None:  # logits = (seed + update) of the pred box
      out = np.ascontiguousarray(out[self._pred_sel])
    return out

  def forward_resident(self, n: int = 1, repeats: int = 1):
    check(self._lib.ffn_forward_resident(self._h, n, repeats))

  # -- device canvases -----------------------------------------------------------
  def create_canvas(self, image) -> 'DeviceCanvasHandle':
    """image: normalised f32 [z, y, x], or an object with `raw_u8`, `mean`,
    `stddev` (inference.NormalizedU8Image): the raw uint8 volume, normalised
    on the device while the FoV is gathered (ffn_canvas_create_u8)."""
    return DeviceCanvasHandle(self, image)

  def step(self, canvases: Sequence['DeviceCanvasHandle'],
           requests: Sequence[StepRequest], params: StepParams):
    """One FoV step for each canvas; returns a list of StepResult copies."""
    n = len(canvases)
    with self._step_lock:
      for k in range(n):
        self._canvas_arr[k] = canvases[k]._h
        ctypes.pointer(self._req_arr[k])[0] = requests[k]
      self._blocking_step(n, self._req_arr, params)
      return [StepResult.from_buffer_copy(self._res_arr[k]) for k in range(n)]

  def _blocking_step(self, n, req, params):
    """ffn_canvas_step; a step voided by the fp16 range check (conv_variant >= 6)
    changed nothing on the device: it is repeated with the exact-f32 kernel
    (conv_variant -1 = the engine's `exact_variant`), which has the full f32
    exponent range, and the engine stays on it."""
    rc = self._lib.ffn_canvas_step(self._h, n, self._canvas_arr, req,
                                   ctypes.byref(params), self._res_arr)
    for _ in range(MAX_VOID_REPEATS):
      if rc not in _lib.ERR_VOIDED:
        break
      self.voided(rc)
      rc = self._lib.ffn_canvas_step(self._h, n, self._canvas_arr, req,
                                     ctypes.byref(params), self._res_arr)
    check(rc)

  def voided(self, rc):
    """Bookkeeping of a step that came back void (it changed nothing and is
    repeated by the caller).  FFN_ERR_RANGE: the engine moves to its exact-f32
    kernel and stays there.  FFN_ERR_FLOW: the resident launch timed out; the
    arithmetic stays, the library repeats with per-layer launches and disables
    the resident launch itself if that keeps happening."""
    if rc == _lib.ERR_FLOW:
      self.flow_fallbacks += 1
    else:
      self.range_fallbacks += 1
      self.set_option('conv_variant', -1, explicit=False)

  #: `segment_many(..., carry=True)`: ffn_canvas_segment_many_carry
  can_carry = True

  def segment_many(self, canvases: Sequence['DeviceCanvasHandle'], starts,
                   params: Sequence['_lib.SegmentParams'], resumes, carry=False):
    """ffn_canvas_segment_many: the segment loops of several canvases advanced
    together inside the library, one batched step per round; returns once at
    least one of them has ended.  -> (results, finished), one entry per canvas;
    results count from the start of each canvas' segment.  A round voided by
    the fp16 range check is repeated with the exact-f32 kernel and the call
    resumed for the canvases still running.  carry: when a loop ends, the step
    the others have prepared is left in flight across the return (see
    ffn_canvas_segment_many_carry; one driving thread, no step budgets)."""
    n = len(canvases)
    carr = (ctypes.c_void_p * n)(*[c._h for c in canvases])
    sarr = (ctypes.c_int32 * 3 * n)()
    parr = (_lib.SegmentParams * n)()
    rarr = (ctypes.c_int32 * n)(*[int(bool(r)) for r in resumes])
    for k in range(n):
      for a in range(3):
        sarr[k][a] = int(starts[k][a])
      ctypes.pointer(parr[k])[0] = params[k]
    def once(keys, sa, pa, ra, res, fin):
      ca = (ctypes.c_void_p * len(keys))(*[canvases[k]._h for k in keys])
      return self._lib.ffn_canvas_segment_many_carry(
          self._h, len(keys), ca, sa, pa, ra, res, fin, 1 if carry else 0)

    fallback = self.voided

    before = [c._many_steps if rarr[k] else 0 for k, c in enumerate(canvases)]
    rc, res, fin = segment_many_with_retry(once, n, sarr, parr, rarr, before,
                                           fallback)
    check(rc)
    for k, c in enumerate(canvases):
      c._many_steps = int(res[k].num_steps)
    return ([_lib.SegmentResult.from_buffer_copy(res[k]) for k in range(n)],
            [bool(fin[k]) for k in range(n)])

  def step_submit(self, canvases: Sequence['DeviceCanvasHandle'],
                  requests: Sequence[StepRequest], params: StepParams) -> int:
    """Enqueues one FoV step for each canvas and returns its ticket at once
    (at most two steps in flight; see ffn_canvas_step_submit)."""
    n = len(canvases)
    slot = self._submit_slot
    carr, rarr = self._slot_canvas_arr[slot], self._slot_req_arr[slot]
    for k in range(n):
      carr[k] = canvases[k]._h
      ctypes.pointer(rarr[k])[0] = requests[k]
    ticket = ctypes.c_uint32(0)
    check(self._lib.ffn_canvas_step_submit(self._h, n, carr, rarr,
                                           ctypes.byref(params),
                                           ctypes.byref(ticket)))
    self._submit_slot ^= 1  # only a successful submit occupies the slot
    self._ticket_slot[ticket.value] = (slot, n, params)
    return ticket.value

  def step_wait(self, ticket: int):
    """Blocks until the step is done; returns its StepResult array (valid until
    the second-next submit)."""
    slot, n, params = self._ticket_slot.pop(ticket)
    res = self._slot_res_arr[slot]
    rc = self._lib.ffn_canvas_step_wait(self._h, ticket, res)
    for _ in range(MAX_VOID_REPEATS):
      if rc not in _lib.ERR_VOIDED:
        break
      # voided (fp16 range check, or the resident launch timed out): nothing was
      # pasted.  Repeat this batch (its descriptor arrays are still intact) with
      # the exact-f32 kernel / without the resident launch; a step of the other
      # group that is already queued finishes first.
      self.voided(rc)
      again = ctypes.c_uint32(0)
      check(self._lib.ffn_canvas_step_submit(
          self._h, n, self._slot_canvas_arr[slot], self._slot_req_arr[slot],
          ctypes.byref(params), ctypes.byref(again)))
      rc = self._lib.ffn_canvas_step_wait(self._h, again.value, res)
    check(rc)
    return res

  def step1(self, canvas: 'DeviceCanvasHandle', request: StepRequest,
            params: StepParams) -> StepResult:
    """Single-canvas path: no per-call copy of the request."""
    with self._step_lock:
      self._canvas_arr[0] = canvas._h
      self._blocking_step(1, ctypes.byref(request), params)
      return StepResult.from_buffer_copy(self._res_arr[0])


class DeviceCanvasHandle:
  """image / seed / segmentation of one subvolume, resident in HBM."""

  def __init__(self, engine: HipEngine, image_f32: np.ndarray):
    self.engine = engine
    self._lib = engine._lib
    self._h = ctypes.c_void_p()
    raw = getattr(image_f32, 'raw_u8', None)
    if raw is not None:  # uint8 canvas: 1 B / voxel in HBM, no f32 host copy
      raw = np.ascontiguousarray(raw, dtype=np.uint8)
      if raw.ndim != 3:
        raise ValueError('image must be 3d (z, y, x)')
      self.shape = tuple(int(s) for s in raw.shape)
      self.is_u8 = True
      check(self._lib.ffn_canvas_create_u8(
          engine._h, raw.ctypes.data, i3(self.shape), float(image_f32.mean),
          float(image_f32.stddev), ctypes.byref(self._h)))
    else:
      image_f32 = _f32(image_f32)
      if image_f32.ndim != 3:
        raise ValueError('image must be 3d (z, y, x)')
      self.shape = tuple(int(s) for s in image_f32.shape)
      self.is_u8 = False
      check(self._lib.ffn_canvas_create(engine._h, image_f32.ctypes.data,
                                        i3(self.shape), ctypes.byref(self._h)))
    self._pt = (ctypes.c_int32 * 3)()
    self._pt_seed = ctypes.c_float()
    self._pt_seg = ctypes.c_int32()
    self._many_steps = 0  # steps of the current segment reported by segment_many
    engine._canvases.add(self)

  def close(self):
    if self._h:
      self._lib.ffn_canvas_destroy(self._h)
      self._h = ctypes.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint:disable=broad-except
      pass

  def init_seed(self, pos, value: float):
    check(self._lib.ffn_canvas_init_seed(self._h, i3(pos), float(value)))

  def read_point(self, pos):
    """(seed[pos], segmentation[pos]); out-of-canvas -> (nan, 0)."""
    self._pt[0], self._pt[1], self._pt[2] = int(pos[0]), int(pos[1]), int(pos[2])
    check(self._lib.ffn_canvas_read_points(
        self._h, 1, ctypes.addressof(self._pt), ctypes.addressof(self._pt_seed),
        ctypes.addressof(self._pt_seg)))
    return self._pt_seed.value, self._pt_seg.value

  def segment_at(self, start_pos, params: '_lib.SegmentParams',
                 resume: bool = False) -> '_lib.SegmentResult':
    """ffn_canvas_segment_at: the whole FoV loop of a segment inside the
    library.  A voided step (`HipEngine.voided`: fp16 range check, resident
    launch timed out) is repeated and the loop resumed."""
    res = _lib.SegmentResult()
    rc = self._lib.ffn_canvas_segment_at(self._h, i3(start_pos),
                                         ctypes.byref(params), int(resume),
                                         ctypes.byref(res))
    budget = params.max_steps
    summed = ('num_steps', 'skip_threshold', 'skip_invalid_pos', 'gate_rejects')
    before = dict.fromkeys(summed, 0)
    for _ in range(MAX_VOID_REPEATS):
      if rc not in _lib.ERR_VOIDED:
        break
      # the voided step changed nothing on the device; the loop keeps the
      # position pending, so resuming repeats exactly that step
      self.engine.voided(rc)
      for name in summed:
        before[name] += getattr(res, name)
      if budget > 0:
        params.max_steps = max(budget - before['num_steps'], 1)
      rc = self._lib.ffn_canvas_segment_at(self._h, i3(start_pos),
                                           ctypes.byref(params), 1,
                                           ctypes.byref(res))
    params.max_steps = budget
    for name in summed:
      setattr(res, name, getattr(res, name) + before[name])
    check(rc)
    return res

  def segment_history(self):
    """(positions [n, 3] int32, deleted counts [n] uint32) of the segment the
    last `segment_at` calls ran with keep_history."""
    total = ctypes.c_size_t(0)
    check(self._lib.ffn_canvas_segment_history(self._h, 0, 0, None, None,
                                               ctypes.byref(total)))
    n = total.value
    pos = np.empty((n, 3), np.int32)
    deleted = np.empty(n, np.uint32)
    if n:
      check(self._lib.ffn_canvas_segment_history(
          self._h, 0, n, pos.ctypes.data, deleted.ctypes.data, None))
    return pos, deleted

  def read_points(self, pos: np.ndarray):
    pos = np.ascontiguousarray(pos, dtype=np.int32).reshape(-1, 3)
    seed = np.empty(len(pos), np.float32)
    seg = np.empty(len(pos), np.int32)
    if len(pos):
      check(self._lib.ffn_canvas_read_points(self._h, len(pos), pos.ctypes.data,
                                             seed.ctypes.data, seg.ctypes.data))
    return seed, seg

  def write_seg_points(self, pos: np.ndarray, values: np.ndarray):
    pos = np.ascontiguousarray(pos, dtype=np.int32).reshape(-1, 3)
    values = np.ascontiguousarray(values, dtype=np.int32).ravel()
    assert len(pos) == len(values)
    if len(pos):
      check(self._lib.ffn_canvas_write_seg_points(
          self._h, len(pos), pos.ctypes.data, values.ctypes.data))

  def any_segmented(self, lo, hi) -> bool:
    out = ctypes.c_int32()
    check(self._lib.ffn_canvas_any_segmented(self._h, i3(lo), i3(hi),
                                             ctypes.byref(out)))
    return bool(out.value)

  def commit_count(self, lo, hi, segment_threshold: float, max_existing_id: int):
    counts = CommitCounts()
    cap = max(int(max_existing_id), 1)
    ids = np.zeros(cap, np.int32)
    cnts = np.zeros(cap, np.int64)
    check(self._lib.ffn_canvas_commit_count(
        self._h, i3(lo), i3(hi), float(segment_threshold),
        int(max_existing_id), ctypes.byref(counts), cap, ids.ctypes.data,
        cnts.ctypes.data))
    n = counts.num_overlapped_ids
    return (int(counts.raw_segmented_voxels),
            int(counts.actual_segmented_voxels), ids[:n].copy(),
            cnts[:n].copy())

  def commit_assign(self, lo, hi, segment_threshold: float, segment_id: int):
    check(self._lib.ffn_canvas_commit_assign(self._h, i3(lo), i3(hi),
                                             float(segment_threshold),
                                             int(segment_id)))

  def segment_turn(self, commit=None, mark=None, candidates=(), mbd=(0, 0, 0),
                   init_value=None):
    """The between-segment turn as one device-side sequence
    (`ffn_canvas_segment_turn`, include/ffn_hip.h; reference
    inference.py:573-660).

    commit: None or (lo, hi, segment_threshold, min_segment_size, segment_id,
    max_existing_id); mark: None or (pos, mode) with mode 1 (always) / 2 (when
    nothing was committed); candidates: the policy's next seeds, zyx, inside the
    canvas; init_value: `init_seed` at the first candidate that passes.
    Returns (raw, actual, overlapped ids, counts, committed, chosen, flags,
    candidate seed values, candidate segmentation values)."""
    rq = _lib.TurnRequest()
    cap = 1
    if commit is not None:
      lo, hi, thr, min_size, sid, max_id = commit
      rq.do_commit = 1
      rq.lo[:] = [int(v) for v in lo]
      rq.hi[:] = [int(v) for v in hi]
      rq.segment_threshold = float(thr)
      rq.min_segment_size = int(min_size)
      rq.segment_id = int(sid)
      rq.max_existing_id = int(max_id)
      cap = max(int(max_id), 1)
    if mark is not None:
      rq.mark_pos[:] = [int(v) for v in mark[0]]
      rq.mark_mode = int(mark[1])
    cand = np.ascontiguousarray(np.asarray(candidates, np.int32).reshape(-1, 3))
    n = len(cand)
    rq.num_candidates = n
    rq.min_boundary_dist[:] = [int(v) for v in mbd]
    rq.do_init = 0 if init_value is None else 1
    rq.init_value = 0.0 if init_value is None else float(init_value)
    ids = np.zeros(cap, np.int32)
    cnts = np.zeros(cap, np.int64)
    flags = np.zeros(max(n, 1), np.int32)
    cseed = np.zeros(max(n, 1), np.float32)
    cseg = np.zeros(max(n, 1), np.int32)
    res = _lib.TurnResult()
    check(self._lib.ffn_canvas_segment_turn(
        self._h, ctypes.byref(rq), cand.ctypes.data if n else None,
        ctypes.byref(res), cap, ids.ctypes.data, cnts.ctypes.data,
        flags.ctypes.data, cseed.ctypes.data, cseg.ctypes.data))
    k = res.counts.num_overlapped_ids
    return (int(res.counts.raw_segmented_voxels),
            int(res.counts.actual_segmented_voxels), ids[:k].copy(),
            cnts[:k].copy(), bool(res.committed), int(res.chosen), flags[:n],
            cseed[:n], cseg[:n])

  def _box(self, lo, hi):
    lo = [int(v) for v in lo]
    hi = [int(v) for v in hi]
    return lo, hi, tuple(h - l for l, h in zip(lo, hi))

  def read_seed(self, lo=None, hi=None) -> np.ndarray:
    lo, hi, shp = self._box(lo or (0, 0, 0), hi or self.shape)
    out = np.empty(shp, np.float32)
    check(self._lib.ffn_canvas_read_seed(self._h, i3(lo), i3(hi),
                                         out.ctypes.data))
    return out

  def read_segmentation(self, lo=None, hi=None) -> np.ndarray:
    lo, hi, shp = self._box(lo or (0, 0, 0), hi or self.shape)
    out = np.empty(shp, np.int32)
    check(self._lib.ffn_canvas_read_segmentation(self._h, i3(lo), i3(hi),
                                                 out.ctypes.data))
    return out

  def write_seed(self, lo, hi, src: np.ndarray):
    lo, hi, shp = self._box(lo, hi)
    src = np.ascontiguousarray(np.broadcast_to(src, shp), dtype=np.float32)
    check(self._lib.ffn_canvas_write_seed(self._h, i3(lo), i3(hi),
                                          src.ctypes.data))

  def write_segmentation(self, lo, hi, src: np.ndarray):
    lo, hi, shp = self._box(lo, hi)
    src = np.ascontiguousarray(np.broadcast_to(src, shp), dtype=np.int32)
    check(self._lib.ffn_canvas_write_segmentation(self._h, i3(lo), i3(hi),
                                                  src.ctypes.data))
