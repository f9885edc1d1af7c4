"""TEST INFRASTRUCTURE: the library's segment loop (ffn_amd/csrc/ffn_host_loop.h)
over the emulated device -- tests/host_loop_shim.cpp built with g++ and driven
through callbacks, so `DeviceCanvas._segment_at_native` runs without a GPU."""
import ctypes
import os
import subprocess

import numpy as np

from ffn_amd import _lib
from ffn_amd import engine as hip_engine
from tests.emulated_device import EmulatedDeviceClient, EmulatedHandle

HERE = os.path.dirname(os.path.abspath(__file__))

_STEP_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(_lib.StepRequest),
                            ctypes.POINTER(_lib.StepParams),
                            ctypes.POINTER(_lib.StepResult))
_READ_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_int32),
                            ctypes.POINTER(ctypes.c_float),
                            ctypes.POINTER(ctypes.c_int32))


_HINT_CB = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.POINTER(ctypes.c_int32))


_BATCH_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int,
                             ctypes.POINTER(ctypes.c_int),
                             ctypes.POINTER(_lib.StepRequest),
                             ctypes.POINTER(_lib.StepParams),
                             ctypes.POINTER(_lib.StepResult))
_READ_K_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int,
                              ctypes.POINTER(ctypes.c_int32),
                              ctypes.POINTER(ctypes.c_float),
                              ctypes.POINTER(ctypes.c_int32))


def build_shim(out_dir):
  """Compiles the shim into `out_dir` and returns the loaded library."""
  out = os.path.join(str(out_dir), 'host_loop_shim.so')
  subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-o',
                         out, os.path.join(HERE, 'host_loop_shim.cpp')])
  lib = ctypes.CDLL(out)
  lib.shim_state_create.restype = ctypes.c_void_p
  lib.shim_state_destroy.argtypes = [ctypes.c_void_p]
  lib.shim_segment_at.restype = ctypes.c_int
  lib.shim_segment_at.argtypes = [
      ctypes.c_void_p, _STEP_CB, _READ_CB, ctypes.POINTER(ctypes.c_int32),
      ctypes.POINTER(_lib.SegmentParams), ctypes.c_int,
      ctypes.POINTER(_lib.SegmentResult)]
  lib.shim_segment_many.restype = ctypes.c_int
  lib.shim_segment_many.argtypes = [
      ctypes.c_int, ctypes.c_void_p, _BATCH_CB, _READ_K_CB, ctypes.c_void_p,
      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
  lib.shim_segment_many_carry.restype = ctypes.c_int
  lib.shim_segment_many_carry.argtypes = lib.shim_segment_many.argtypes + [
      ctypes.c_int]
  lib.shim_carry_active.restype = ctypes.c_int
  lib.shim_history.restype = ctypes.c_size_t
  lib.shim_history.argtypes = [ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_void_p, ctypes.c_size_t]
  lib.shim_sizeof_params.restype = ctypes.c_size_t
  lib.shim_sizeof_result.restype = ctypes.c_size_t
  lib.shim_set_hint_cb.argtypes = [_HINT_CB]
  return lib


class ShimHandle(EmulatedHandle):
  """Emulated canvas + `segment_at` through the C++ loop (what
  DeviceCanvasHandle.segment_at is on the GPU)."""

  shim = None
  client = None
  fail_step = None  # step number at which the device reports a voided step once
  fail_code = _lib.ERR_RANGE  # ... as FFN_ERR_RANGE, or FFN_ERR_FLOW
  total_native_calls = 0  # over all handles
  # the speculative conv0_a launch of the HIP device (ffn_hip.hip SpecArgs),
  # emulated: [hints taken, steps made at a hinted position]; every such step
  # asserts that the device's choice (first hinted position that is valid on the
  # canvas as the previous step left it) is the position the loop then popped
  spec_stats = [0, 0]

  def __init__(self, image):
    super().__init__(image)
    self._state = self.shim.shim_state_create()
    self.native_calls = 0
    self.steps_seen = []

  def segment_at(self, start_pos, params, resume=False):
    self.native_calls += 1
    ShimHandle.total_native_calls += 1

    hint = []   # the loop's hint for the step being made
    spec = {}   # 'list', 'choice': what the device queued behind the last step

    def hint_cb(n, pos):
      hint[:] = [(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]) for k in range(n)]

    def step_cb(req, par, res):
      if self.fail_step is not None and len(self.steps_seen) == self.fail_step:
        self.fail_step = None
        return self.fail_code
      pos = tuple(req.contents.pos)
      if spec and pos in spec['list']:
        assert spec['list'].index(pos) == spec['choice'], (pos, spec)
        ShimHandle.spec_stats[1] += 1
      spec.clear()
      self.steps_seen.append(pos)
      out = self.client.step(self, req.contents, par.contents)
      ctypes.memmove(res, ctypes.addressof(out), ctypes.sizeof(out))
      if hint:  # conv0_a's choice, on the canvas as this step left it
        thr = par.contents.move_threshold
        choice = -1
        for k, q in enumerate(hint):
          sv, gv = self.read_point(q)
          if not (sv < thr) and gv <= 0:
            choice = k
            break
        spec.update(list=list(hint), choice=choice)
        ShimHandle.spec_stats[0] += 1
        del hint[:]
      return 0

    def read_cb(pos, seed, seg):
      s, g = self.read_point((pos[0], pos[1], pos[2]))
      seed[0], seg[0] = s, g
      return 0

    res = _lib.SegmentResult()
    start = (ctypes.c_int32 * 3)(*start_pos)
    hint_fn = _HINT_CB(hint_cb)
    self.shim.shim_set_hint_cb(hint_fn)
    rc = self.shim.shim_segment_at(self._state, _STEP_CB(step_cb),
                                   _READ_CB(read_cb), start,
                                   ctypes.byref(params), int(resume),
                                   ctypes.byref(res))
    summed = ('num_steps', 'skip_threshold', 'skip_invalid_pos', 'gate_rejects')
    before = dict.fromkeys(summed, 0)
    for _ in range(hip_engine.MAX_VOID_REPEATS):  # what DeviceCanvasHandle.segment_at does
      if rc not in _lib.ERR_VOIDED:
        break
      for name in summed:
        before[name] += getattr(res, name)
      rc = self.shim.shim_segment_at(self._state, _STEP_CB(step_cb),
                                     _READ_CB(read_cb), start,
                                     ctypes.byref(params), 1, ctypes.byref(res))
    for name in summed:
      setattr(res, name, getattr(res, name) + before[name])
    self.shim.shim_set_hint_cb(_HINT_CB())
    assert rc == 0, rc
    return res

  def segment_history(self):
    n = self.shim.shim_history(self._state, None, None, 0)
    pos = np.empty((n, 3), np.int32)
    deleted = np.empty(n, np.uint32)
    self.shim.shim_history(self._state, pos.ctypes.data, deleted.ctypes.data, n)
    return pos, deleted


class ShimClient(EmulatedDeviceClient):
  in_thread = True

  def create_canvas(self, image):
    ShimHandle.client = self
    return ShimHandle(image)


class ShimEngine:
  """What `MultiCanvasDriver` needs of a HipEngine, over emulated canvases:
  `segment_many` through the C++ loop (ffn_host::segment_many), blocking `step`
  for canvases on the Python loop."""

  def __init__(self, client, max_batch):
    self.client = client
    self.max_batch = max_batch
    self.many_calls = 0
    self.batch_sizes = []
    self.fail_round = None  # batched round at which the device reports a voided step once
    self.fail_code = _lib.ERR_RANGE
    self.flow_fallbacks = 0
    # 'short': void (once) the first round in which a loop of the call has just
    # ended -- the round carries fewer steps than the call has canvases
    self.rounds = 0
    self.range_fallbacks = 0
    #: ffn_canvas_segment_many_carry: calls that asked for it, calls that left a
    #: step in flight; defer_error: a voided carried step shows at the next call
    self.can_carry = True
    self.carry_calls = 0
    self.carried = 0
    self.defer_error = False

  def step(self, handles, requests, params):
    return [self.client.step(h, r, params) for h, r in zip(handles, requests)]

  def _segment_many_once(self, handles, sarr, parr, rarr, res, fin, carry=False):
    n = len(handles)

    def batch_cb(nb, idx, reqs, par, out):
      if self.fail_round is not None and (
          nb < n if self.fail_round == 'short' else self.rounds == self.fail_round):
        self.fail_round = None
        return self.fail_code
      self.rounds += 1
      self.batch_sizes.append(nb)
      for b in range(nb):
        h = handles[idx[b]]
        h.steps_seen.append(tuple(reqs[b].pos))
        r = self.client.step(h, reqs[b], par.contents)
        ctypes.memmove(ctypes.addressof(out[b]), ctypes.addressof(r),
                       ctypes.sizeof(r))
      return 0

    def read_cb(k, pos, seed, seg):
      s, g = handles[k].read_point((pos[0], pos[1], pos[2]))
      seed[0], seg[0] = s, g
      return 0

    states = (ctypes.c_void_p * n)(*[h._state for h in handles])
    if carry:
      self.carry_calls += 1
      rc = ShimHandle.shim.shim_segment_many_carry(
          n, states, _BATCH_CB(batch_cb), _READ_K_CB(read_cb), sarr, parr, rarr,
          res, fin, int(self.defer_error))
      self.carried += ShimHandle.shim.shim_carry_active()
      return rc
    assert not ShimHandle.shim.shim_carry_active()
    return ShimHandle.shim.shim_segment_many(
        n, states, _BATCH_CB(batch_cb), _READ_K_CB(read_cb), sarr, parr, rarr, res,
        fin)

  def segment_many(self, handles, starts, params, resumes, carry=False):
    """Mirrors HipEngine.segment_many, ERR_RANGE handling included."""
    n = len(handles)
    assert n <= self.max_batch
    self.many_calls += 1
    sarr = (ctypes.c_int32 * 3 * n)()
    parr = (_lib.SegmentParams * n)()
    rarr = (ctypes.c_int32 * n)(*[int(bool(r)) for r in resumes])
    for k in range(n):
      for a in range(3):
        sarr[k][a] = int(starts[k][a])
      ctypes.pointer(parr[k])[0] = params[k]
    def once(keys, sa, pa, ra, res, fin):
      return self._segment_many_once([handles[k] for k in keys], sa, pa, ra, res,
                                     fin, carry)

    def fallback(rc):
      if rc == _lib.ERR_FLOW:
        self.flow_fallbacks += 1
      else:
        self.range_fallbacks += 1

    before = [getattr(h, '_many_steps', 0) if rarr[k] else 0
              for k, h in enumerate(handles)]
    rc, res, fin = hip_engine.segment_many_with_retry(once, n, sarr, parr, rarr,
                                                      before, fallback)
    assert rc == 0, rc
    for k, h in enumerate(handles):
      h._many_steps = int(res[k].num_steps)
    return ([_lib.SegmentResult.from_buffer_copy(res[k]) for k in range(n)],
            [bool(fin[k]) for k in range(n)])
