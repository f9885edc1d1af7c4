"""Movement policies of the FoV loop (mirror of reference ffn/inference/movement.py).

`get_scored_move_offsets` (:42-100), `BaseMovementPolicy` (:103-163),
`FaceMaxMovementPolicy` (:166-222), `get_policy_fn` (:225-244),
`MovementRestrictor` (:247-336).  All triples are (z, y, x).

The BFS move queue, the quantised visited set and the sort order stay in
Python exactly as in the reference.  What moves to the GPU is the scoring: a
device-resident canvas hands `FaceMaxMovementPolicy.update` a `FacePrediction`
(six face maxima + first-occurrence argmax computed by wavefront reductions in
the paste kernel) instead of a 33^3 array.
"""

from __future__ import annotations

from collections import deque
import json
from typing import Optional
import weakref

import numpy as np
from scipy.special import logit

from ..training import model as ffn_model
from ..training.import_util import import_symbol


def get_scored_move_offsets(deltas, prob_map: np.ndarray, threshold: float = 0.9):
  """Yields (score, (dz, dy, dx)) for the maximum of each face of the +-delta
  cuboid around the centre of `prob_map` (reference movement.py:42-100)."""
  center = np.array(prob_map.shape) // 2
  assert center.size == 3
  subvol_sel = [slice(c - dx, c + dx + 1) for c, dx in zip(center, deltas)]
  done = set()
  for axis, axis_delta in enumerate(deltas):
    if axis_delta == 0:
      continue
    for axis_offset in (-axis_delta, axis_delta):
      face_sel = subvol_sel[:]
      face_sel[axis] = axis_offset + center[axis]
      face_prob = prob_map[tuple(face_sel)]
      shape = face_prob.shape
      # argmax: first occurrence in C order.
      face_pos = np.unravel_index(face_prob.argmax(), shape)
      score = face_prob[face_pos]
      if score < threshold:
        continue
      relative_pos = [face_pos[0] - shape[0] // 2, face_pos[1] - shape[1] // 2]
      relative_pos.insert(axis, axis_offset)
      ret = (score, tuple(int(v) for v in relative_pos))
      if ret not in done:
        done.add(ret)
        yield ret


class FacePrediction:
  """What a device-resident FoV step returns instead of the logit array.

  face_score / face_index follow the reference's iteration order: axis z, y, x
  and sign -, + (movement.py:67-71); face_index is the flat C-order argmax
  inside the face; face_seg the segmentation value at that voxel.
  """

  __slots__ = ('face_score', 'face_index', 'face_seg', 'shape', 'read_fn')

  def __init__(self, face_score, face_index, face_seg, shape, read_fn=None):
    self.face_score = face_score
    self.face_index = face_index
    self.face_seg = face_seg
    self.shape = tuple(shape)
    self.read_fn = read_fn

  def __array__(self, dtype=None, copy=None):
    """Full logit array (device -> host read) for policies that need it."""
    if self.read_fn is None:
      raise ValueError('this prediction carries face maxima only')
    arr = self.read_fn()
    return arr if dtype is None else arr.astype(dtype)

  def scored_move_offsets(self, deltas, threshold):
    """Same output as get_scored_move_offsets(deltas, logits, threshold), plus
    the segmentation value at each move target: (score, offset, seg)."""
    out = []
    seen = set()
    k = 0
    for axis in range(3):
      d = int(deltas[axis])
      for sign in (-1, 1):
        kk = k
        k += 1
        if d == 0:
          continue
        score = float(self.face_score[kk])
        if score < threshold:
          continue
        # face = the two non-fixed axes, in zyx order
        others = [a for a in range(3) if a != axis]
        ncols = 2 * int(deltas[others[1]]) + 1
        fi, fj = divmod(int(self.face_index[kk]), ncols)
        rel = [fi - int(deltas[others[0]]), fj - int(deltas[others[1]])]
        rel.insert(axis, sign * d)
        item = (score, tuple(rel))
        if item not in seen:
          seen.add(item)
          out.append((score, tuple(rel), int(self.face_seg[kk])))
    return out


class BaseMovementPolicy:
  """Base class for movement policy queues (reference movement.py:103-163)."""

  def __init__(self, canvas, scored_coords, deltas):
    self.canvas = weakref.proxy(canvas)
    self.scored_coords = scored_coords
    self.deltas = np.array(deltas)

  def __len__(self):
    return len(self.scored_coords)

  def __iter__(self):
    return self

  def __next__(self):
    raise StopIteration()

  def next(self):
    return self.__next__()

  def append(self, item):
    self.scored_coords.append(item)

  def update(self, prob_map, position):
    raise NotImplementedError()

  def get_state(self):
    raise NotImplementedError()

  def restore_state(self, state):
    raise NotImplementedError()

  def reset_state(self, start_pos):
    raise NotImplementedError()


class FaceMaxMovementPolicy(BaseMovementPolicy):
  """Selects candidates from maxima on prediction cuboid faces.

  Same queue semantics as the reference (FIFO deque, quantised visited set,
  descending (score, coord) insertion order).  Internally every queue entry also
  carries its quantised position -- computed once at append time instead of at
  every scan -- and `get_state` / `restore_state` convert to / from the
  reference's (score, coord) layout so checkpoints stay interchangeable.
  """

  def __init__(self, canvas, deltas=(4, 8, 8), score_threshold=0.9):
    self.done_rounded_coords = set()
    self.score_threshold = score_threshold
    self._start_pos = None
    super().__init__(canvas, deque([]), deltas)
    # plain-int copies: the per-step path must not go through numpy
    self._d = tuple(int(v) for v in self.deltas)
    self._dh = tuple(v // 2 for v in self._d)
    self._dm = tuple(max(v, 1) for v in self._d)
    # per face (axis z,y,x; sign -,+): axis, signed offset, deltas of the two
    # in-face axes, number of face columns
    self._faces = []
    for axis in range(3):
      others = [a for a in range(3) if a != axis]
      for sign in (-1, 1):
        self._faces.append((axis, sign * self._d[axis], self._d[others[0]],
                            self._d[others[1]], 2 * self._d[others[1]] + 1))

  def reset_state(self, start_pos):
    self.scored_coords = deque([])
    self.done_rounded_coords = set()
    self._start_pos = tuple(int(v) for v in start_pos)

  def get_state(self):
    queue = deque((s, list(c)) for s, c, _ in self.scored_coords)
    return [(queue, self.done_rounded_coords, self._start_pos)]

  def restore_state(self, state):
    queue, done, start = state[0]
    self._start_pos = tuple(int(v) for v in start)
    self.done_rounded_coords = set(tuple(int(v) for v in q) for q in done)
    self.scored_coords = deque([])
    for score, coord in queue:
      self.append((score, coord))

  def append(self, item):
    coord = tuple(int(v) for v in item[1])
    self.scored_coords.append((item[0], coord, self.quantize_pos(coord)))

  def __next__(self):
    """Pops positions from the queue until a valid one is found."""
    sc = self.scored_coords
    done = self.done_rounded_coords
    is_valid = self.canvas.is_valid_pos
    while sc:
      _, coord, q = sc.popleft()
      if q in done:
        continue
      if is_valid(coord):
        return coord
    raise StopIteration()

  def quantize_pos(self, pos):
    """Quantises symmetrically to a grid downsampled by deltas
    ((rel + delta//2) // max(delta, 1), floor division; movement.py:200-208)."""
    s = self._start_pos
    return ((int(pos[0]) - s[0] + self._dh[0]) // self._dm[0],
            (int(pos[1]) - s[1] + self._dh[1]) // self._dm[1],
            (int(pos[2]) - s[2] + self._dh[2]) // self._dm[2])

  def peek_candidates(self, limit):
    """First `limit` queued coordinates not yet visited, in queue order.

    Already-visited entries at the head are dropped for good: `__next__` would
    skip them anyway, without side effects."""
    sc = self.scored_coords
    done = self.done_rounded_coords
    while sc and sc[0][2] in done:
      sc.popleft()
    out = []
    for entry in sc:
      if entry[2] in done:
        continue
      out.append(entry[1])
      if len(out) >= limit:
        break
    return out

  def update(self, prob_map, position):
    """Adds movements to the queue for the cuboid face maxima of `prob_map`."""
    self.done_rounded_coords.add(self.quantize_pos(position))
    if isinstance(prob_map, FacePrediction):
      thr = self.score_threshold
      fs, fi_, fg = prob_map.face_score, prob_map.face_index, prob_map.face_seg
      moves = []
      for k, (axis, off, d_row, d_col, ncols) in enumerate(self._faces):
        if off == 0:
          continue
        score = fs[k]
        if score < thr:
          continue
        fi, fj = divmod(fi_[k], ncols)
        if axis == 0:
          rel = (off, fi - d_row, fj - d_col)
        elif axis == 1:
          rel = (fi - d_row, off, fj - d_col)
        else:
          rel = (fi - d_row, fj - d_col, off)
        moves.append((score, rel, fg[k]))
      if len(moves) > 1:
        # descending (score, offset), as sorted(..., reverse=True) of the
        # reference (movement.py:220); duplicates of the same (score, offset)
        # -- one voxel shared by two faces -- are dropped (movement.py:98-100)
        moves.sort(reverse=True)
        k = 1
        while k < len(moves):
          if moves[k][0] == moves[k - 1][0] and moves[k][1] == moves[k - 1][1]:
            del moves[k]
          else:
            k += 1
      new = []
      pz, py, px = position
      sz, sy, sx = self._start_pos
      hz, hy, hx = self._dh
      mz, my, mx = self._dm
      sc = self.scored_coords
      for score, rel, seg in moves:
        cz, cy, cx = rel[0] + pz, rel[1] + py, rel[2] + px
        coord = (cz, cy, cx)
        # quantize_pos, inlined (this loop runs 6 x per FoV step)
        sc.append((score, coord, ((cz - sz + hz) // mz, (cy - sy + hy) // my,
                                  (cx - sx + hx) // mx)))
        new.append((coord, score, seg))
      return new
    scored = sorted(
        get_scored_move_offsets(self.deltas, prob_map,
                                threshold=self.score_threshold), reverse=True)
    for score, rel in scored:
      self.append((score, [rel[i] + position[i] for i in range(3)]))
    return None


def get_policy_fn(request, model_info: ffn_model.ModelInfo):
  """Returns a policy factory for an InferenceRequest (movement.py:225-244)."""
  if request.movement_policy_name:
    movement_policy_class = globals().get(request.movement_policy_name, None)
    if movement_policy_class is None:
      movement_policy_class = import_symbol(request.movement_policy_name)
  else:
    movement_policy_class = FaceMaxMovementPolicy
  if request.movement_policy_args:
    kwargs = json.loads(request.movement_policy_args)
  else:
    kwargs = {}
  if 'deltas' not in kwargs:
    kwargs['deltas'] = model_info.deltas[::-1]
  if 'score_threshold' not in kwargs:
    kwargs['score_threshold'] = float(
        logit(request.inference_options.move_threshold))
  return lambda canvas: movement_policy_class(canvas, **kwargs)


class MovementRestrictor:
  """Restricts the movement of the FFN FoV (reference movement.py:247-336)."""

  def __init__(self, mask: Optional[np.ndarray] = None,
               shift_mask: Optional[np.ndarray] = None, shift_mask_fov=None,
               shift_mask_threshold: int = 4, shift_mask_scale: int = 1,
               seed_mask: Optional[np.ndarray] = None):
    self.mask = mask
    self.seed_mask = seed_mask
    self._shift_mask_scale = shift_mask_scale
    self.shift_mask = None
    if shift_mask is not None:
      self.shift_mask = (np.max(np.abs(shift_mask), axis=0) >=
                         shift_mask_threshold)
      assert shift_mask_fov is not None
      self._shift_mask_fov_pre_offset = np.array(shift_mask_fov.start[::-1])
      self._shift_mask_fov_post_offset = np.array(shift_mask_fov.end[::-1]) - 1

  @property
  def is_trivial(self):
    return self.mask is None and self.shift_mask is None

  def is_valid_seed(self, pos):
    if self.seed_mask is not None and self.seed_mask[pos]:
      return False
    return True

  def is_valid_pos(self, pos):
    if self.mask is not None and self.mask[pos]:
      return False
    if self.shift_mask is not None:
      np_pos = np.array(pos)
      fov_low = np.maximum(np_pos + self._shift_mask_fov_pre_offset, 0)
      fov_high = np_pos + self._shift_mask_fov_post_offset
      start = fov_low // self._shift_mask_scale
      end = fov_high // self._shift_mask_scale
      if np.any(self.shift_mask[fov_low[0]:(fov_high[0] + 1),
                                start[1]:(end[1] + 1),
                                start[2]:(end[2] + 1)]):
        return False
    return True
