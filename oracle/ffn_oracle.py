"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the ffn_amd product).

CPU restatement (numpy + the plain-C conv stack in convstack_oracle.c) of the one
hot path this repository accelerates: the FFN field-of-view inference loop.

Each function cites the reference lines it restates (paths relative to the
google/ffn checkout).  Pinning status:

* Host logic (`update_at`, disco bias, paste-back, face-max move scoring, BFS
  move queue, segment commit): pinned by tests/golden/ref_*.npz, which were
  produced by running the reference's own unmodified Python modules
  (ffn/inference/{inference,movement,seed,storage,segmentation}.py) through
  import shims in this container -- see tools/make_golden.py.
* Conv-stack arithmetic (`forward`): the reference delegates it to TensorFlow /
  tf-slim, which are absent here and for which the reference ships no golden
  vectors: PARITY UNPINNED vs TensorFlow; cross-checked against torch conv3d
  (f32 / f64) in tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""

from __future__ import annotations

import collections
import ctypes
import os
import subprocess
import time

import numpy as np
from scipy.special import logit

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
  global _LIB
  if _LIB is None:
    path = os.path.join(_HERE, 'libffn_oracle.so')
    if not os.path.exists(path):
      subprocess.check_call(['make', '-C', _HERE, 'libffn_oracle.so'])
    lib = ctypes.CDLL(path)
    lib.ffn_oracle_forward.restype = ctypes.c_int
    lib.ffn_oracle_forward.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    ]
    lib.ffn_oracle_weight_count.restype = ctypes.c_size_t
    lib.ffn_oracle_weight_count.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.ffn_oracle_set_threads.restype = ctypes.c_int
    lib.ffn_oracle_set_threads.argtypes = [ctypes.c_int]
    _LIB = lib
  return _LIB


def set_threads(n=0):
  """Sets / queries the OpenMP thread count of the C conv stack."""
  return _lib().ffn_oracle_set_threads(int(n))


# ---------------------------------------------------------------------------
# Weights
# ---------------------------------------------------------------------------


def conv_names(depth):
  """Variable scopes in graph order (convstack_3d.py:38-54)."""
  names = ['conv0_a', 'conv0_b']
  for i in range(1, depth):
    names += ['conv%d_a' % i, 'conv%d_b' % i]
  names.append('conv_lom')
  return names


def weights_blob(variables, depth):
  """Flattens {'seed_update/<scope>/{weights,biases}': array} into the blob
  layout documented in convstack_oracle.c."""
  parts = []
  for name in conv_names(depth):
    parts.append(
        np.ascontiguousarray(variables['seed_update/%s/weights' % name],
                             dtype=np.float32).ravel())
    parts.append(
        np.ascontiguousarray(variables['seed_update/%s/biases' % name],
                             dtype=np.float32).ravel())
  return np.concatenate(parts)


def random_weights(depth, features=32, seed=0, stddev=0.05):
  """Random-init variables dict with the reference's names and shapes."""
  rng = np.random.RandomState(seed)
  out = {}
  for name in conv_names(depth):
    if name == 'conv0_a':
      shape = (3, 3, 3, 2, features)
    elif name == 'conv_lom':
      shape = (1, 1, 1, features, 1)
    else:
      shape = (3, 3, 3, features, features)
    out['seed_update/%s/weights' % name] = rng.normal(
        0, stddev, shape).astype(np.float32)
    out['seed_update/%s/biases' % name] = rng.normal(
        0, stddev, shape[-1:]).astype(np.float32)
  return out


# ---------------------------------------------------------------------------
# Forward pass  (convstack_3d.py:26-56, 83-95; model.py:168-183)
# ---------------------------------------------------------------------------


def forward(image, seed, blob, depth, features=32, stop_after=-1):
  """logits = seed + conv_stack(concat(image, seed)).

  Args:
    image, seed: [n, z, y, x] or [z, y, x] float32
    blob: flat float32 weights (see weights_blob)
    stop_after: >= 0 returns the [z, y, x, features] activation after that conv
  """
  image = np.ascontiguousarray(image, dtype=np.float32)
  seed = np.ascontiguousarray(seed, dtype=np.float32)
  squeeze = image.ndim == 3
  if squeeze:
    image = image[None]
    seed = seed[None]
  n, z, y, x = image.shape
  assert seed.shape == image.shape
  blob = np.ascontiguousarray(blob, dtype=np.float32)
  lib = _lib()
  assert blob.size == lib.ffn_oracle_weight_count(depth, features), (
      blob.size, lib.ffn_oracle_weight_count(depth, features))
  out = np.empty_like(seed)
  act = None
  act_ptr = None
  if stop_after >= 0:
    act = np.empty((z, y, x, features), dtype=np.float32)
    act_ptr = act.ctypes.data
  rc = lib.ffn_oracle_forward(image.ctypes.data, seed.ctypes.data, n, z, y, x,
                              depth, features, blob.ctypes.data,
                              out.ctypes.data, stop_after, act_ptr)
  if rc != 0:
    raise RuntimeError('ffn_oracle_forward failed: %d' % rc)
  if stop_after >= 0:
    return act
  return out[0] if squeeze else out


def forward_torch(image, seed, variables, depth, threads=None, f64=False):
  """The same forward pass restated on torch-CPU (oneDNN conv3d, f32): an
  independent implementation used to cross-check `forward` and as the stronger
  CPU baseline (BASELINE.md section 3 names it as the stand-in for the
  reference's TF CPU path, TensorFlow being unavailable).  f64: the whole
  stack in double precision (f32 inputs and weights, logits rounded to f32 at
  the end) -- the arithmetic every f32 implementation approximates."""
  import torch
  import torch.nn.functional as F
  if threads:
    torch.set_num_threads(int(threads))
  image = np.asarray(image, np.float32)
  seed = np.asarray(seed, np.float32)
  squeeze = image.ndim == 3
  if squeeze:
    image, seed = image[None], seed[None]
  cache = variables.setdefault('__torch64__' if f64 else '__torch__', {})

  def wb(name):
    if name not in cache:
      w = torch.from_numpy(np.ascontiguousarray(
          variables['seed_update/%s/weights' % name])).permute(4, 3, 0, 1, 2)
      b = torch.from_numpy(np.ascontiguousarray(
          variables['seed_update/%s/biases' % name]))
      if f64:
        w, b = w.double(), b.double()
      cache[name] = (w.contiguous(), b)
    return cache[name]

  with torch.no_grad():
    s = torch.from_numpy(seed)
    im = torch.from_numpy(image)
    if f64:
      s, im = s.double(), im.double()
    x = torch.stack([im, s], dim=1)
    net = torch.relu(F.conv3d(x, *wb('conv0_a'), padding=1))
    net = F.conv3d(net, *wb('conv0_b'), padding=1)
    for i in range(1, depth):
      skip = net
      net = torch.relu(net)
      net = torch.relu(F.conv3d(net, *wb('conv%d_a' % i), padding=1))
      net = F.conv3d(net, *wb('conv%d_b' % i), padding=1) + skip
    net = torch.relu(net)
    out = s + F.conv3d(net, *wb('conv_lom'))[:, 0]
  out = out.float().numpy()
  return out[0] if squeeze else out


_LIB64 = None


def forward_f64c(image, seed, blob, depth, threads=0):
  """The forward pass in double precision throughout (convstack_f64.c: built on
  request for this host, `make -C oracle libffn_oracle_f64.so`): what
  forward_torch(f64=True) computes, ~5x faster.  One FoV or a batch of them."""
  global _LIB64
  if _LIB64 is None:
    path = os.path.join(_HERE, 'libffn_oracle_f64.so')
    if not os.path.exists(path):
      subprocess.check_call(['make', '-C', _HERE, 'libffn_oracle_f64.so'])
    lib = ctypes.CDLL(path)
    lib.ffn_oracle_forward_f64.restype = ctypes.c_int
    lib.ffn_oracle_forward_f64.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    _LIB64 = lib
  image = np.ascontiguousarray(image, dtype=np.float32)
  seed = np.ascontiguousarray(seed, dtype=np.float32)
  squeeze = image.ndim == 3
  if squeeze:
    image, seed = image[None], seed[None]
  blob = np.ascontiguousarray(blob, dtype=np.float32)
  out = np.empty_like(seed)
  for k in range(image.shape[0]):
    z, y, x = image.shape[1:]
    rc = _LIB64.ffn_oracle_forward_f64(image[k].ctypes.data, seed[k].ctypes.data, z, y, x,
                                       depth, blob.ctypes.data, out[k].ctypes.data,
                                       int(threads or 0))
    if rc != 0:
      raise RuntimeError('ffn_oracle_forward_f64 failed: %d' % rc)
  return out[0] if squeeze else out


# ---------------------------------------------------------------------------
# Move scoring  (movement.py:42-100)
# ---------------------------------------------------------------------------


def scored_move_offsets(deltas, prob_map, threshold):
  """List of (score, (dz, dy, dx)) -- max of each of the 6 faces at +-delta.

  argmax is first-occurrence in C order (movement.py:86); duplicates of the
  same (score, offset) are dropped (movement.py:98-100).
  """
  center = np.array(prob_map.shape) // 2
  lo = [int(c - d) for c, d in zip(center, deltas)]
  hi = [int(c + d + 1) for c, d in zip(center, deltas)]
  out = []
  seen = set()
  for axis in range(3):
    d = int(deltas[axis])
    if d == 0:
      continue
    for off in (-d, d):
      sel = [slice(lo[0], hi[0]), slice(lo[1], hi[1]), slice(lo[2], hi[2])]
      sel[axis] = int(center[axis]) + off
      face = prob_map[tuple(sel)]
      flat = int(face.argmax())
      i, j = divmod(flat, face.shape[1])
      score = face[i, j]
      if score < threshold:
        continue
      rel = [i - face.shape[0] // 2, j - face.shape[1] // 2]
      rel.insert(axis, off)
      item = (score, tuple(rel))
      if item not in seen:
        seen.add(item)
        out.append(item)
  return out


def face_maxima(deltas, prob_map):
  """Raw per-face (score, flat_index) for the 6 faces in the reference's
  iteration order (axis z,y,x; sign -,+) -- what the HIP step kernel returns."""
  center = np.array(prob_map.shape) // 2
  lo = [int(c - d) for c, d in zip(center, deltas)]
  hi = [int(c + d + 1) for c, d in zip(center, deltas)]
  scores = np.zeros(6, np.float32)
  idx = np.zeros(6, np.int32)
  k = 0
  for axis in range(3):
    d = int(deltas[axis])
    for off in (-d, d):
      sel = [slice(lo[0], hi[0]), slice(lo[1], hi[1]), slice(lo[2], hi[2])]
      sel[axis] = int(center[axis]) + off
      face = prob_map[tuple(sel)]
      idx[k] = int(face.argmax())
      scores[k] = face.ravel()[idx[k]]
      k += 1
  return scores, idx


# ---------------------------------------------------------------------------
# Canvas  (inference.py:137-683) + FaceMaxMovementPolicy (movement.py:166-222)
# ---------------------------------------------------------------------------


class Options:
  """InferenceOptions (inference.proto:131-168) in *probability* space."""

  def __init__(self, init_activation=0.95, pad_value=0.05, move_threshold=0.9,
               segment_threshold=0.6, min_segment_size=1000,
               min_boundary_dist=(1, 1, 1), disco_seed_threshold=0.0):
    self.init_activation = init_activation
    self.pad_value = pad_value
    self.move_threshold = move_threshold
    self.segment_threshold = segment_threshold
    self.min_segment_size = min_segment_size
    self.min_boundary_dist = tuple(min_boundary_dist)  # zyx
    self.disco_seed_threshold = disco_seed_threshold


def f32_logit(p):
  """Canvas stores logit(p) back into a float32 proto field
  (inference.py:189-195): value = float(float32(logit(float(float32(p)))))."""
  return float(np.float32(logit(float(np.float32(p)))))


class OracleCanvas:
  """Single-subvolume FoV loop on numpy arrays (the reference's Canvas)."""

  def __init__(self, image, blob, depth, fov_zyx, deltas_zyx, options,
               features=32, policy_threshold=None):
    self.image = np.asarray(image, dtype=np.float32)
    self.shape = self.image.shape
    self.blob = blob
    self.depth = depth
    self.features = features
    self.fov = np.array(fov_zyx)
    self.deltas = np.array(deltas_zyx)
    self.margin = self.fov // 2
    o = options
    self.init_activation = f32_logit(o.init_activation)
    self.pad_value = f32_logit(o.pad_value)
    self.move_threshold = f32_logit(o.move_threshold)
    self.segment_threshold = f32_logit(o.segment_threshold)
    self.disco_seed_threshold = float(np.float32(o.disco_seed_threshold))
    self.min_segment_size = o.min_segment_size
    self.mbd = np.array(o.min_boundary_dist)
    # Runner path: policy threshold is the f64 logit of the f32 proto value
    # (movement.py:241-242); bare Canvas: the f32-rounded one (inference.py:252).
    if policy_threshold is None:
      policy_threshold = float(logit(float(np.float32(o.move_threshold))))
    self.policy_threshold = policy_threshold

    self.seed = np.full(self.shape, np.nan, dtype=np.float32)
    self.segmentation = np.zeros(self.shape, dtype=np.int32)
    self.origins = {}
    self.overlaps = {}
    self.max_id = 0
    self.counters = collections.Counter()
    self.trace = []  # (pos, scored moves) per FoV step, for golden comparison
    self.forward_fn = None  # optional override: f(image_fov, seed_fov)->logits

  # inference.py:312-346
  def is_valid_pos(self, pos, ignore_move_threshold=False):
    if not ignore_move_threshold:
      if self.seed[pos] < self.move_threshold:
        self.counters['skip_threshold'] += 1
        return False
    p = np.array(pos)
    if np.any(p - self.margin < 0) or np.any(p + self.margin >= self.shape):
      self.counters['skip_invalid_pos'] += 1
      return False
    if self.segmentation[pos] > 0:
      self.counters['skip_invalid_pos'] += 1
      return False
    return True

  tie_tol = 1e-4  # the per-step float tolerance of the parity tests

  def _forward(self, img, seed):
    if self.forward_fn is not None:
      return self.forward_fn(img, seed)
    return forward(img, seed, self.blob, self.depth, self.features)

  # inference.py:386-441 (+ 348-384)
  def update_at(self, pos):
    start = np.array(pos) - self.margin
    end = start + self.fov
    sel = tuple(slice(s, e) for s, e in zip(start, end))
    logit_seed = np.array(self.seed[sel])
    logit_seed[np.isnan(logit_seed)] = np.float32(self.pad_value)
    logits = self._forward(self.image[sel], logit_seed)
    if self.disco_seed_threshold >= 0:
      th_max = logit(0.5)
      old_seed = self.seed[sel]
      # keep_history record (inference.py:420-423)
      with np.errstate(invalid='ignore'):
        self.last_deleted = int(np.sum(
            (old_seed >= np.float32(logit(0.8))) & (logits < th_max)))
        # voxels of that count a forward within `tie_tol` of this one may put
        # on the other side of th_max (the count is exact up to these)
        self.last_deleted_ties = int(np.sum(
            (old_seed >= np.float32(logit(0.8))) &
            (np.abs(logits - th_max) <= self.tie_tol)))
      if np.mean(logits >= self.move_threshold) > self.disco_seed_threshold:
        with np.errstate(invalid='ignore'):
          mask = (old_seed < th_max) & (logits > old_seed)
        logits[mask] = old_seed[mask]
    self.seed[sel] = logits
    self.counters['update_at-calls'] += 1
    return logits

  # movement.py:200-208
  def _quantize(self, pos, start_pos):
    rel = np.array(pos) - np.array(start_pos)
    return tuple(int(v) for v in (rel + self.deltas // 2) //
                 np.maximum(self.deltas, 1))

  # inference.py:460-533 with FaceMaxMovementPolicy inlined
  def segment_at(self, start_pos):
    self.seed[...] = np.nan  # init_seed -> NumpyArray.clear (storage.py:69-71)
    self.seed[start_pos] = self.init_activation
    queue = collections.deque([(self.policy_threshold * 2, tuple(start_pos))])
    done = set()
    self.min_pos = np.array(start_pos)
    self.max_pos = np.array(start_pos)
    num_iters = 0
    while True:
      # FaceMaxMovementPolicy.__next__ (movement.py:186-198)
      pos = None
      while queue:
        _, coord = queue.popleft()
        coord = tuple(int(c) for c in coord)
        if self._quantize(coord, start_pos) in done:
          continue
        if self.is_valid_pos(coord):
          pos = coord
          break
      if pos is None:
        break
      if self.seed[start_pos] < self.move_threshold:
        self.counters['seed_got_too_weak'] += 1
        break
      pred = self.update_at(pos)
      self.min_pos = np.minimum(self.min_pos, pos)
      self.max_pos = np.maximum(self.max_pos, pos)
      num_iters += 1
      # FaceMaxMovementPolicy.update (movement.py:210-222)
      done.add(self._quantize(pos, start_pos))
      moves = sorted(
          scored_move_offsets(self.deltas, pred, self.policy_threshold),
          reverse=True)
      self.trace.append((pos, [(float(s), o) for s, o in moves]))
      for score, rel in moves:
        queue.append((score, tuple(rel[i] + pos[i] for i in range(3))))
    return num_iters

  # inference.py:538-683
  def segment_all(self, seeds):
    for pos in seeds:
      pos = tuple(int(p) for p in pos)
      if not self.is_valid_pos(pos, ignore_move_threshold=True):
        continue
      low = np.array(pos) - self.mbd
      high = np.array(pos) + self.mbd + 1
      sel = tuple(slice(s, e) for s, e in zip(low, high))
      if np.any(self.segmentation[sel] > 0):
        self.segmentation[pos] = -1
        continue
      t0 = time.time()
      num_iters = self.segment_at(pos)
      t_seg = time.time() - t0
      if num_iters <= 0:
        continue
      if self.seed[pos] < self.move_threshold:
        if self.segmentation[pos] == 0:
          self.segmentation[pos] = -1
        continue
      sel = tuple(
          slice(max(int(s), 0), int(e) + 1)
          for s, e in zip(self.min_pos - self.fov // 2,
                          self.max_pos + self.fov // 2))
      mask = self.seed[sel] >= self.segment_threshold
      raw = int(np.sum(mask))
      ids, counts = np.unique(self.segmentation[sel][mask], return_counts=True)
      valid = ids > 0
      ids, counts = ids[valid], counts[valid]
      mask &= self.segmentation[sel] <= 0
      actual = int(np.sum(mask))
      if actual < self.min_segment_size:
        if self.segmentation[pos] == 0:
          self.segmentation[pos] = -1
        continue
      self.counters['voxels-segmented'] += actual
      self.counters['voxels-overlapping'] += raw - actual
      self.max_id += 1
      while self.max_id in self.origins:
        self.max_id += 1
      self.segmentation[sel][mask] = self.max_id
      self.overlaps[self.max_id] = np.array([ids, counts])
      self.origins[self.max_id] = (pos, num_iters, t_seg)


# ---------------------------------------------------------------------------
# Seeds, output helpers
# ---------------------------------------------------------------------------


def grid_seeds(shape, margin, step=16, offsets=(0, 8, 4, 12, 2, 10, 14)):
  """PolicyGrid3d (seed.py:411-430) + the base-class margin filter
  (seed.py:80-89)."""
  coords = []
  for off in offsets:
    for z in range(off, shape[0], step):
      for y in range(off, shape[1], step):
        for x in range(off, shape[2], step):
          coords.append((z, y, x))
  coords = np.array(coords)
  m = np.array(margin)[None]
  keep = np.all((coords - m >= 0) & (coords + m < np.array(shape)), axis=1)
  return coords[keep]


def quantize_probability(prob):
  """storage.py:137-143."""
  ret = np.digitize(prob, np.linspace(0.0, 1.0, 255))
  ret[np.isnan(prob)] = 0
  return ret.astype(np.uint8)


def reduce_id_bits(seg):
  """segmentation.py:66-86."""
  max_uint32 = 2**32 - 1
  max_uint16 = 2**16 - 1
  max_uint8 = 2**8 - 1
  max_id = seg.max()
  if max_id <= max_uint8:
    return seg.astype(np.uint8)
  elif max_id <= max_uint16:
    return seg.astype(np.uint16)
  elif max_id <= max_uint32:
    return seg.astype(np.uint32)
  return seg
