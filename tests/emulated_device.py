"""TEST DOUBLE for the device side of the C-ABI (numpy + the oracle forward).

Lets the CPU-only test-suite exercise the *host* logic of `DeviceCanvas`
(candidate prefetch, value caches, commit protocol) without a GPU.  It
restates, in numpy, what `ffn_canvas_step` / `ffn_canvas_commit_*` etc. are
specified to return (include/ffn_hip.h).  Never imported by the product.
"""

import numpy as np

from ffn_amd import _lib
from ffn_amd.inference import executor
from oracle import ffn_oracle


class EmulatedHandle:

  def __init__(self, image):
    self.image = np.array(image, np.float32)
    self.shape = self.image.shape
    self.seed = np.full(self.shape, np.nan, np.float32)
    self.seg = np.zeros(self.shape, np.int32)
    self.point_reads = 0

  def close(self):
    pass

  def init_seed(self, pos, value):
    self.seed[...] = np.nan
    self.seed[tuple(pos)] = value

  def read_point(self, pos):
    self.point_reads += 1
    pos = tuple(int(p) for p in pos)
    if any(p < 0 or p >= s for p, s in zip(pos, self.shape)):
      return float('nan'), 0
    return float(self.seed[pos]), int(self.seg[pos])

  def read_points(self, pos):
    """ffn_canvas_read_points: (seed [n], segmentation [n]) of n positions in one
    call (out-of-canvas: nan / 0)."""
    pos = np.asarray(pos).reshape(-1, 3)
    self.point_reads += 1
    seeds = np.full(len(pos), np.nan, np.float32)
    segs = np.zeros(len(pos), np.int32)
    for k, p in enumerate(pos):
      p = tuple(int(v) for v in p)
      if all(0 <= v < s for v, s in zip(p, self.shape)):
        seeds[k], segs[k] = self.seed[p], self.seg[p]
    return seeds, segs

  def write_seg_points(self, pos, values):
    for p, v in zip(pos, values):
      self.seg[tuple(int(x) for x in p)] = v

  def any_segmented(self, lo, hi):
    sel = tuple(slice(max(l, 0), h) for l, h in zip(lo, hi))
    return bool(np.any(self.seg[sel] > 0))

  def commit_count(self, lo, hi, thr, max_existing_id):
    sel = tuple(slice(l, h) for l, h in zip(lo, hi))
    mask = self.seed[sel] >= np.float32(thr)
    raw = int(mask.sum())
    ids, counts = np.unique(self.seg[sel][mask], return_counts=True)
    keep = ids > 0
    actual = int((mask & (self.seg[sel] <= 0)).sum())
    return raw, actual, ids[keep].astype(np.int32), counts[keep].astype(np.int64)

  def commit_assign(self, lo, hi, thr, sid):
    sel = tuple(slice(l, h) for l, h in zip(lo, hi))
    mask = (self.seed[sel] >= np.float32(thr)) & (self.seg[sel] <= 0)
    self.seg[sel][mask] = sid

  def segment_turn(self, commit=None, mark=None, candidates=(), mbd=(0, 0, 0),
                   init_value=None):
    """ffn_canvas_segment_turn as include/ffn_hip.h specifies it, with the
    single-question calls above."""
    self.turns = getattr(self, 'turns', 0) + 1
    raw = actual = 0
    ids = np.zeros(0, np.int32)
    counts = np.zeros(0, np.int64)
    committed = False
    if commit is not None:
      lo, hi, thr, min_size, sid, max_id = commit
      raw, actual, ids, counts = self.commit_count(lo, hi, thr, max_id)
      keep = ids <= max_id
      ids, counts = ids[keep], counts[keep]
      if actual >= min_size:
        self.commit_assign(lo, hi, thr, sid)
        committed = True
    if mark is not None:
      pos, mode = mark
      pos = tuple(int(v) for v in pos)
      if (mode == 1 or (mode == 2 and not committed)) and self.seg[pos] == 0:
        self.seg[pos] = -1
    cand = np.asarray(candidates, np.int32).reshape(-1, 3)
    n = len(cand)
    flags = np.full(n, 3, np.int32)
    cseed = np.zeros(n, np.float32)
    cseg = np.zeros(n, np.int32)
    for k, p in enumerate(cand):  # (values: the canvas before any marker)
      cseed[k], cseg[k] = self.seed[tuple(p)], self.seg[tuple(p)]
    chosen = -1
    for k, p in enumerate(cand):
      p = tuple(int(v) for v in p)
      if self.seg[p] > 0:
        flags[k] = 1
        continue
      lo = [v - m for v, m in zip(p, mbd)]
      hi = [v + m + 1 for v, m in zip(p, mbd)]
      if self.any_segmented(lo, hi):
        flags[k] = 2
        self.seg[p] = -1
        continue
      flags[k] = 0
      chosen = k
      break
    if chosen >= 0 and init_value is not None:
      self.init_seed(tuple(int(v) for v in cand[chosen]), init_value)
    return raw, actual, ids, counts, committed, chosen, flags, cseed, cseg

  def read_seed(self, lo=None, hi=None):
    lo = lo or (0, 0, 0)
    hi = hi or self.shape
    return np.array(self.seed[tuple(slice(l, h) for l, h in zip(lo, hi))])

  def read_segmentation(self, lo=None, hi=None):
    lo = lo or (0, 0, 0)
    hi = hi or self.shape
    return np.array(self.seg[tuple(slice(l, h) for l, h in zip(lo, hi))])

  def write_seed(self, lo, hi, src):
    self.seed[tuple(slice(l, h) for l, h in zip(lo, hi))] = src

  def write_segmentation(self, lo, hi, src):
    self.seg[tuple(slice(l, h) for l, h in zip(lo, hi))] = src


class EmulatedDeviceClient(executor.ExecutorClient):
  """ExecutorClient with the device-canvas surface, computed on the CPU."""

  def __init__(self, counters, blob, depth, fov_zyx, deltas_zyx, pred_zyx=None):
    super().__init__(counters, None)
    self.blob = blob
    self.depth = depth
    self.fov = np.array(fov_zyx)
    self.deltas = np.array(deltas_zyx)
    # ffn_engine_set_pred_size: the centred box of the FoV a step scores / pastes
    self.pred = np.array(pred_zyx if pred_zyx is not None else fov_zyx)
    self.pred_lo = (self.fov - self.pred) // 2
    self.steps = 0

  def start(self):
    self._client_id = 0
    return 0

  def finish(self):
    self._client_id = None

  def predict(self, seed, image, fetches):
    out = ffn_oracle.forward(image, seed, self.blob, self.depth)
    box = tuple(slice(l, l + p) for l, p in zip(self.pred_lo, self.pred))
    return {'logits': np.ascontiguousarray(out[box])[..., None]}

  def create_canvas(self, image):
    return EmulatedHandle(image)

  def canvas_call(self, fn, *args, **kwargs):
    return fn(*args, **kwargs)

  def step(self, h, req, params):
    self.steps += 1
    pos = np.array(list(req.pos))
    start = pos - self.fov // 2
    sel = tuple(slice(s, s + f) for s, f in zip(start, self.fov))
    old = np.array(h.seed[sel])
    seed_in = old.copy()
    seed_in[np.isnan(seed_in)] = np.float32(params.pad_value)
    logits = ffn_oracle.forward(h.image[sel], seed_in, self.blob, self.depth)
    # the prediction: the centred box of the FoV (the whole FoV by default)
    box = tuple(slice(l, l + p) for l, p in zip(self.pred_lo, self.pred))
    sel = tuple(slice(s + l, s + l + p)
                for s, l, p in zip(start, self.pred_lo, self.pred))
    old = np.array(old[box])
    logits = np.ascontiguousarray(logits[box])
    cnt = int(np.sum(logits >= np.float32(params.move_threshold)))
    disco = (params.disco_seed_threshold >= 0 and
             cnt / logits.size > params.disco_seed_threshold)
    res = _lib.StepResult()
    if params.deleted_threshold == params.deleted_threshold:
      with np.errstate(invalid='ignore'):
        res.num_deleted = int(np.sum(
            (old >= np.float32(params.deleted_threshold)) & (logits < 0)))
    if disco:
      with np.errstate(invalid='ignore'):
        mask = (old < 0) & (logits > old)
      logits[mask] = old[mask]
    h.seed[sel] = logits
    scores, idx = ffn_oracle.face_maxima(self.deltas, logits)
    k = 0
    for axis in range(3):
      others = [a for a in range(3) if a != axis]
      for sign in (-1, 1):
        res.face_score[k] = scores[k]
        res.face_index[k] = idx[k]
        ncols = 2 * self.deltas[others[1]] + 1
        fi, fj = divmod(int(idx[k]), int(ncols))
        rel = [0, 0, 0]
        rel[axis] = sign * self.deltas[axis]
        rel[others[0]] = fi - self.deltas[others[0]]
        rel[others[1]] = fj - self.deltas[others[1]]
        coord = tuple(int(p + r) for p, r in zip(pos, rel))
        res.face_seg[k] = int(h.seg[coord])
        k += 1
    res.start_logit = h.seed[tuple(req.start_pos)]
    res.num_above_move = cnt
    res.disco_applied = int(disco)
    for k in range(req.num_candidates):
      cpos = tuple(req.candidates[k])
      res.cand_seed[k] = h.seed[cpos]
      res.cand_seg[k] = h.seg[cpos]
    return res


class EmulatedLabelOps:
  """TEST DOUBLE for ffn_amd.labels.LabelOps (numpy via oracle/labels_oracle):
  lets CPU-only tests run the host logic layered on the label kernels
  (distributed reconciliation, segmentation tables).  Pair order is shuffled,
  as the GPU hash table returns pairs in unspecified order."""

  def __init__(self, seed=0):
    self._rng = np.random.RandomState(seed)
    self._resident = None

  def pair_counts(self, a, b=None):
    from oracle import labels_oracle
    a = np.asarray(a)
    pa, pb, cnt = labels_oracle.pair_counts(
        a.astype(np.uint64) & np.uint64(0xffffffff) if a.dtype.itemsize == 4
        else a, None if b is None else (
            np.asarray(b).astype(np.uint64) & np.uint64(0xffffffff)
            if np.asarray(b).dtype.itemsize == 4 else b))
    perm = self._rng.permutation(pa.size)
    self._resident = (a, None if b is None else np.asarray(b), pa, pb)
    return pa[perm], pb[perm], cnt[perm], perm.astype(np.uint32)

  def apply_pair_labels(self, slots, new_labels):
    a, b, pa, pb = self._resident
    table = np.zeros(pa.size, np.uint64)
    table[np.asarray(slots, np.int64)] = new_labels
    key = a.ravel().astype(np.uint64)
    if b is not None:
      key = key | (b.ravel().astype(np.uint64) << np.uint64(32))
    ukeys = pa | (pb << np.uint64(32))
    order = np.argsort(ukeys)
    pos = order[np.searchsorted(ukeys[order], key)]
    return table[pos].astype(a.dtype).reshape(a.shape)

  def remap(self, arr, keys, values, keep_missing=True):
    from oracle import labels_oracle
    return labels_oracle.remap(arr, keys, values, keep_missing)

  def connected_components(self, arr, connectivity=1, stats=False):
    from oracle import labels_oracle
    out, first, sizes, fz = labels_oracle.connected_components(arr,
                                                               connectivity)
    return (out, first, sizes, fz) if stats else out


class EmulatedSeeder:
  """TEST DOUBLE for ffn_amd.seeding.Seeder (scipy via oracle/seeds_oracle)."""

  def peaks(self, image, exclusion_mask=None, force_edge=None,
            voxel_size_zyx=(1, 1, 1)):
    from oracle import seeds_oracle
    return seeds_oracle.policy_peaks(image, exclusion_mask, force_edge,
                                     voxel_size_zyx)

  def peaks_canvas(self, handle, voxel_size_zyx=(1, 1, 1)):
    from oracle import seeds_oracle
    return seeds_oracle.policy_peaks(handle.image, handle.seg > 0, None,
                                     voxel_size_zyx)

  def edt(self, mask, voxel_size_zyx=(1, 1, 1)):
    from scipy import ndimage
    return ndimage.distance_transform_edt(np.asarray(mask) != 0,
                                          sampling=voxel_size_zyx)
