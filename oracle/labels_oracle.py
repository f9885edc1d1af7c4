"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the ffn_amd product).

CPU restatement (numpy / scipy) of the label routines behind the GPU label
operations (include/ffn_labels.h, ffn_amd/inference/segmentation.py,
ffn_amd/distributed.py).  Each function cites the reference lines it restates
(paths relative to the google/ffn checkout).  Pinning status:

* split_segmentation_by_intersection, clean_up_and_count, clear_dust: PINNED by
  tests/golden/ref_labels.npz, produced by the reference's own unmodified
  ffn/inference/segmentation.py (tools/make_golden_labels.py).
* connected components: the reference delegates to the un-vendored
  connectomics.segmentation.labels.split_disconnected_components (setup.py:42,
  unpinned); the fixtures used its published body with the real scikit-image
  0.18.3 `measure.label`.  Pinned against that.
* reconcile (union-find assembly of overlapping sub-boxes): NOT in the reference
  ("currently *not implemented*", doc/manual.md:119-127) -- there is nothing to
  pin against; this restatement is the specification the GPU path is held to.
"""

from __future__ import annotations

import numpy as np
import scipy.sparse
import scipy.sparse.csgraph


# -- building blocks (what the kernels compute) ---------------------------------

def pair_counts(a, b=None):
  """Unique (a[i], b[i]) pairs with voxel counts, ascending (b, a) -- the
  np.unique of the packed key at segmentation.py:256-260."""
  a = np.asarray(a).ravel().astype(np.uint64)
  b = (np.zeros_like(a) if b is None else
       np.asarray(b).ravel().astype(np.uint64))
  if a.size == 0:
    z = np.zeros(0, np.uint64)
    return z, z.copy(), z.copy()
  if a.max() > 0xffffffff or b.max() > 0xffffffff:
    raise ValueError('ids must fit 32 bits (remap first)')
  keys, counts = np.unique(a | (b << np.uint64(32)), return_counts=True)
  return (keys & np.uint64(0xffffffff), keys >> np.uint64(32),
          counts.astype(np.uint64))


def remap(arr, keys, values, keep_missing=True):
  arr = np.asarray(arr)
  keys = np.asarray(keys, np.uint64)
  values = np.asarray(values, np.uint64)
  flat = arr.ravel().astype(np.uint64)
  out = flat.copy() if keep_missing else np.zeros_like(flat)
  if keys.size:
    order = np.argsort(keys)
    ks, vs = keys[order], values[order]
    pos = np.clip(np.searchsorted(ks, flat), 0, ks.size - 1)
    hit = ks[pos] == flat
    out[hit] = vs[pos[hit]]
  return out.astype(arr.dtype).reshape(arr.shape)


def _forward_offsets(connectivity):
  offs = []
  for dz in (-1, 0, 1):
    for dy in (-1, 0, 1):
      for dx in (-1, 0, 1):
        order = (dz != 0) + (dy != 0) + (dx != 0)
        if order == 0 or order > connectivity:
          continue
        if (dz, dy, dx) > (0, 0, 0):
          offs.append((dz, dy, dx))
  return offs


def connected_components(labels, connectivity=1):
  """Components of equal non-zero label, numbered 1.. in raster order of their
  first voxel (skimage.measure.label(background=0) order, which
  split_disconnected_components returns; called at segmentation.py:161-162).

  Returns (out, first_index[k], sizes[k], first_zero_index)."""
  labels = np.asarray(labels)
  shape = labels.shape
  n = labels.size
  idx = np.arange(n, dtype=np.int64).reshape(shape)
  rows, cols = [], []
  for dz, dy, dx in _forward_offsets(connectivity):
    src = tuple(slice(max(0, -d), s - max(0, d)) for d, s in
                zip((dz, dy, dx), shape))
    dst = tuple(slice(max(0, d), s - max(0, -d)) for d, s in
                zip((dz, dy, dx), shape))
    same = (labels[src] == labels[dst]) & (labels[src] != 0)
    rows.append(idx[src][same])
    cols.append(idx[dst][same])
  rows = np.concatenate(rows) if rows else np.zeros(0, np.int64)
  cols = np.concatenate(cols) if cols else np.zeros(0, np.int64)
  graph = scipy.sparse.coo_matrix((np.ones(rows.size, np.int8), (rows, cols)),
                                  shape=(n, n))
  _, comp = scipy.sparse.csgraph.connected_components(graph, directed=False)
  flat = labels.ravel()
  fg = np.nonzero(flat != 0)[0]
  out = np.zeros(n, np.int64)
  first_index = np.zeros(0, np.uint64)
  sizes = np.zeros(0, np.uint64)
  if fg.size:
    uniq, first, inverse, counts = np.unique(comp[fg], return_index=True,
                                             return_inverse=True,
                                             return_counts=True)
    order = np.argsort(first)  # raster order of each component's first voxel
    rank = np.empty(uniq.size, np.int64)
    rank[order] = np.arange(1, uniq.size + 1)
    out[fg] = rank[inverse]
    first_index = fg[first[order]].astype(np.uint64)
    sizes = counts[order].astype(np.uint64)
  zeros = np.nonzero(flat == 0)[0]
  first_zero = int(zeros[0]) if zeros.size else -1
  return out.astype(labels.dtype).reshape(shape), first_index, sizes, first_zero


# -- reference routines ----------------------------------------------------------

def clear_dust(data, min_size=10):
  """segmentation.py:21-63."""
  if data.size == 0 or min_size <= 0 or not np.any(data):
    return data
  ids, sizes = np.unique(data, return_counts=True)
  small = ids[(sizes < min_size) & (ids != 0)]
  if small.size > 0:
    data[np.isin(data, small)] = 0
  return data


def clean_up_and_count(seg, split_cc=True, connectivity=1, min_size=0):
  """segmentation.py:125-178 with both maps computed; `seg` modified in place."""
  if not np.any(seg):
    if seg.size == 0:
      return {}, {}
    zero = seg.dtype.type(0)
    return {zero: zero}, {zero: np.int64(seg.size)}
  seg_orig = seg.copy()
  if split_cc:
    seg[...] = connected_components(seg, connectivity)[0]
  if min_size > 0:
    clear_dust(seg, min_size)
  cc_ids, cc_idx, cc_counts = np.unique(seg.ravel(), return_index=True,
                                        return_counts=True)
  orig_ids = seg_orig.ravel()[cc_idx]
  return dict(zip(cc_ids, orig_ids)), dict(zip(cc_ids, cc_counts))


def split_segmentation_by_intersection(a, b, min_size):
  """segmentation.py:181-290 (same control flow, numpy only)."""
  if a.shape != b.shape:
    raise ValueError
  out_shape = a.shape
  af, bf = a.ravel(), b.ravel()

  def remap_input(x):  # :208-243
    if x.dtype != np.uint64:
      raise TypeError
    max_uint32 = 2**32 - 1
    max_id = x.max()
    orig = None
    if max_id > max_uint32:
      orig, x = np.unique(x, return_inverse=True)
      if len(orig) > max_uint32:
        raise ValueError('More than 2**32-1 unique labels not supported')
      x = np.asarray(x, dtype=np.uint64).ravel()
      if orig[0] != 0:
        orig = np.concatenate([np.array([0], dtype=np.uint64), orig])
        x[...] += 1
    return x, max_id, orig

  ra, max_id, a_reverse = remap_input(af)
  rb, _, _ = remap_input(bf)
  joint = np.bitwise_or(ra, rb << np.uint64(32))
  uniq, inverse, counts = np.unique(joint, return_inverse=True,
                                    return_counts=True)
  ua = np.bitwise_and(uniq, np.uint64(0xFFFFFFFF))
  ub = uniq >> np.uint64(32)
  best = {}
  for la, lb, c in zip(ua, ub, counts):  # :266-273
    cur = best.setdefault(la, (lb, c))
    if cur[1] < c:
      best[la] = (lb, c)
  new_labels = np.zeros(len(uniq), np.uint64)
  max_id = int(max_id)
  for i, (la, lb, c) in enumerate(zip(ua, ub, counts)):  # :276-288
    if c < min_size or la == 0:
      new = 0
    elif lb == best[la][0]:
      new = a_reverse[la] if a_reverse is not None else la
    else:
      max_id += 1
      new = max_id
    new_labels[i] = new
  a[...] = new_labels[inverse.ravel()].reshape(out_shape)


# -- union-find assembly of overlapping sub-boxes (doc/manual.md:119-127) --------

class UnionFind:
  """Deterministic union-find: the root of a set is its smallest id."""

  def __init__(self):
    self.parent = {}

  def find(self, x):
    p = self.parent.setdefault(x, x)
    while p != self.parent[p]:
      p = self.parent[p]
    while self.parent[x] != p:  # path compression
      self.parent[x], x = p, self.parent[x]
    return p

  def union(self, x, y):
    rx, ry = self.find(x), self.find(y)
    if rx == ry:
      return
    if rx < ry:
      self.parent[ry] = rx
    else:
      self.parent[rx] = ry


def margin_edges(seg_global_ids, assembled_box, core_lo, core_hi,
                 min_overlap_voxels=1, min_overlap_fraction=0.0):
  """Merge candidates of one sub-box: its own labelling (already in the global
  id space) against the assembled volume, over the sub-box's margin (the part
  outside its core, where the assembled ids come from neighbouring sub-boxes).

  An edge (a, g) is kept when the two labels share `count` margin voxels with
  count >= min_overlap_voxels and count >= min_overlap_fraction * min(|a in
  margin|, |g in margin|).  Returns an int64 array [k, 3] of (a, g, count),
  sorted.
  """
  a = np.array(seg_global_ids, np.uint64)
  g = np.array(assembled_box, np.uint64)
  core = tuple(slice(l, h) for l, h in zip(core_lo, core_hi))
  a[core] = 0
  g[core] = 0
  pa, pb, cnt = pair_counts(a, g)
  size_a = {}
  size_g = {}
  for x, y, c in zip(pa, pb, cnt):
    size_a[x] = size_a.get(x, 0) + int(c)
    size_g[y] = size_g.get(y, 0) + int(c)
  edges = []
  for x, y, c in zip(pa, pb, cnt):
    if x == 0 or y == 0 or x == y:
      continue
    c = int(c)
    if c < min_overlap_voxels:
      continue
    if c < min_overlap_fraction * min(size_a[x], size_g[y]):
      continue
    edges.append((int(x), int(y), c))
  edges.sort()
  return np.array(edges, np.int64).reshape(-1, 3)


def reconcile(sub_results, shape_zyx, min_overlap_voxels=1,
              min_overlap_fraction=0.0):
  """Single-process specification of ffn_amd.distributed.reconcile_segmentations.

  sub_results: list of (box, seg) in sub-box index order, box having .corner,
  .size, .core_lo, .core_hi; ids local per sub-box.  Returns (global int32
  volume, sorted edge array, {id: root}).
  """
  offsets, base = [], 0
  for _, seg in sub_results:
    offsets.append(base)
    base += int(seg.max()) if seg.size else 0
  out = np.zeros(tuple(shape_zyx), np.int32)
  shifted = []
  for (box, seg), off in zip(sub_results, offsets):
    s = np.where(seg > 0, seg.astype(np.int64) + off, 0)
    shifted.append(s)
    lo = [c - b for c, b in zip(box.core_lo, box.corner)]
    hi = [c - b for c, b in zip(box.core_hi, box.corner)]
    out[box.core_lo[0]:box.core_hi[0], box.core_lo[1]:box.core_hi[1],
        box.core_lo[2]:box.core_hi[2]] = s[lo[0]:hi[0], lo[1]:hi[1],
                                          lo[2]:hi[2]]
  all_edges = []
  for (box, _), s in zip(sub_results, shifted):
    sel = tuple(slice(c, c + n) for c, n in zip(box.corner, box.size))
    lo = [c - b for c, b in zip(box.core_lo, box.corner)]
    hi = [c - b for c, b in zip(box.core_hi, box.corner)]
    all_edges.append(margin_edges(s, out[sel], lo, hi, min_overlap_voxels,
                                  min_overlap_fraction))
  edges = (np.concatenate(all_edges) if all_edges else
           np.zeros((0, 3), np.int64))
  uf = UnionFind()
  for x, y, _ in sorted(map(tuple, edges)):
    uf.union(int(x), int(y))
  roots = {x: uf.find(x) for x in list(uf.parent)}
  keys = np.array(sorted(k for k, v in roots.items() if k != v), np.uint64)
  vals = np.array([roots[int(k)] for k in keys], np.uint64)
  out = remap(out, keys, vals, keep_missing=True)
  order = np.lexsort((edges[:, 2], edges[:, 1], edges[:, 0])) if len(
      edges) else np.zeros(0, np.int64)
  return out, edges[order], roots
