"""Routines for manipulating arrays of segmentation ids
(reference ffn/inference/segmentation.py), with the per-voxel work on the GPU.

Same names, arguments and error behaviour as the reference; every function
that touches whole volumes runs HBM-bound HIP kernels through
`ffn_amd.labels.LabelOps` (include/ffn_labels.h) and finishes the small
per-label tables in numpy.  There is no CPU fallback for those.
"""

from __future__ import annotations

import numpy as np

from .. import labels as label_ops

_MAX_UINT32 = 2**32 - 1


def _ops(device_id=None):
  return label_ops.default_ops(0 if device_id is None else device_id)


def clear_dust(data: np.ndarray, min_size: int = 10, device_id=None):
  """Replaces objects smaller than `min_size` with 0, in place
  (segmentation.py:21-63).  Returns `data`."""
  if data.size == 0 or min_size <= 0:
    return data
  ops = _ops(device_id)
  ids, _, sizes, _ = ops.pair_counts(data)
  small = ids[(sizes < np.uint64(min_size)) & (ids != 0)]
  if small.size > 0:
    data[...] = ops.remap(data, small, np.zeros_like(small),
                          keep_missing=True).astype(data.dtype, copy=False)
  return data


def reduce_id_bits(segmentation: np.ndarray):
  """Converts to the smallest unsigned type that holds every id
  (segmentation.py:66-86)."""
  max_id = segmentation.max()
  for dt in (np.uint8, np.uint16, np.uint32):
    if max_id <= np.iinfo(dt).max:
      return segmentation.astype(dt)
  return segmentation


def split_disconnected_components(labels: np.ndarray, connectivity: int = 1,
                                  device_id=None):
  """Relabels the connected components of equal non-zero label 1.. in raster
  order of their first voxel; 0 stays 0.

  This is connectomics.segmentation.labels.split_disconnected_components (an
  un-vendored dependency, called at segmentation.py:161-162), i.e.
  skimage.measure.label(labels, background=0, connectivity=connectivity).
  """
  if labels.size == 0:
    return labels.copy()
  out = _ops(device_id).connected_components(labels, connectivity)
  return out.astype(labels.dtype, copy=False)


def clean_up(seg: np.ndarray, split_cc=True, connectivity=1, min_size=0,
             return_id_map=False, device_id=None):
  """Runs connected components and removes small objects, in place
  (segmentation.py:89-122)."""
  cc_to_orig, _ = clean_up_and_count(seg, split_cc, connectivity, min_size,
                                     compute_id_map=return_id_map,
                                     compute_counts=False,
                                     device_id=device_id)
  if return_id_map:
    return cc_to_orig


def clean_up_and_count(seg: np.ndarray, split_cc=True, connectivity=1,
                       min_size=0, compute_id_map=True, compute_counts=True,
                       device_id=None):
  """clean_up that also returns {new id: original id} and {new id: voxels}
  (segmentation.py:125-178).  `seg` is modified in place."""
  if not np.any(seg):
    if seg.size == 0:
      return ({} if compute_id_map else None, {} if compute_counts else None)
    zero = seg.dtype.type(0)
    return ({zero: zero} if compute_id_map else None,
            {zero: np.int64(seg.size)} if compute_counts else None)

  ops = _ops(device_id)
  dt = seg.dtype.type
  if split_cc:
    orig_flat = seg.ravel().copy() if compute_id_map else None
    out, first, sizes, first_zero = ops.connected_components(
        seg, connectivity, stats=True)
    ids = np.arange(1, first.size + 1, dtype=np.uint64)
    keep = np.ones(first.size, bool)
    if min_size > 0:
      keep = sizes >= np.uint64(min_size)
      if not keep.all():
        dusted = ids[~keep]
        out = ops.remap(out, dusted, np.zeros_like(dusted), keep_missing=True)
    seg[...] = out.astype(seg.dtype, copy=False)
    cc_to_orig = cc_to_count = None
    n_zero = int(seg.size) - int(sizes[keep].sum())
    # np.unique(seg.ravel(), return_index, return_counts) of the reference,
    # rebuilt from the per-component tables: id 0 first (if present).
    if compute_id_map:
      cc_to_orig = {}
      if n_zero:
        zero_first = [first_zero] if first_zero >= 0 else []
        zero_first += [int(f) for f in first[~keep]]
        cc_to_orig[dt(0)] = orig_flat[min(zero_first)]
      for i, f in zip(ids[keep], first[keep]):
        cc_to_orig[dt(i)] = orig_flat[int(f)]
    if compute_counts:
      cc_to_count = {}
      if n_zero:
        cc_to_count[dt(0)] = np.int64(n_zero)
      for i, c in zip(ids[keep], sizes[keep]):
        cc_to_count[dt(i)] = np.int64(c)
    return cc_to_orig, cc_to_count

  # No CC pass: ids keep their values; only dust removal + the unique tables.
  seg_orig = seg.copy() if compute_id_map else None
  if min_size > 0:
    clear_dust(seg, min_size, device_id=device_id)
  cc_to_orig = cc_to_count = None
  if compute_id_map:
    cc_ids, cc_idx = np.unique(seg.ravel(), return_index=True)
    cc_to_orig = dict(zip(cc_ids, seg_orig.ravel()[cc_idx]))
  if compute_counts:
    ids, _, counts, _ = ops.pair_counts(seg)
    order = np.argsort(ids)
    cc_to_count = dict(zip(ids[order].astype(seg.dtype),
                           counts[order].astype(np.int64)))
  return cc_to_orig, cc_to_count


def _remap_input(x: np.ndarray):
  """Fits ids into 32 bits if needed (segmentation.py:208-243): returns
  (remapped, max_id, orig_values_map or None)."""
  if x.dtype != np.uint64:
    raise TypeError
  max_id = x.max() if x.size else np.uint64(0)
  orig_values_map = None
  # (>= rather than the reference's >: id 2^32 - 1 itself is reserved by the
  # device hash table; the remap is transparent to the result.)
  if max_id >= _MAX_UINT32:
    orig_values_map, x = np.unique(x, return_inverse=True)
    if len(orig_values_map) > _MAX_UINT32:
      raise ValueError('More than 2**32-1 unique labels not supported')
    x = np.asarray(x, dtype=np.uint64).ravel()
    if orig_values_map[0] != 0:
      orig_values_map = np.concatenate(
          [np.array([0], dtype=np.uint64), orig_values_map])
      x[...] += 1
  return x, max_id, orig_values_map


def split_segmentation_by_intersection(a: np.ndarray, b: np.ndarray,
                                       min_size: int, device_id=None):
  """Intersection of two segmentations (segmentation.py:181-290).

  Every unique (id_a, id_b) pair of overlapping voxels becomes one output
  segment: the pair with the largest overlap for `id_a` keeps the label `id_a`,
  the others get fresh ids counting up from a.max() + 1 in ascending (id_b,
  id_a) order; pairs smaller than `min_size`, and everything where a == 0,
  become 0.  `a` is modified in place, `b` is not changed.

  Raises:
    TypeError: if a or b are not uint64
    ValueError: if shapes differ or there are more than 2**32-1 unique labels
  """
  if a.shape != b.shape:
    raise ValueError
  if a.dtype != np.uint64 or b.dtype != np.uint64:
    raise TypeError
  if a.size == 0:
    return
  ra, max_id, a_reverse_map = _remap_input(a.ravel())
  rb, _, _ = _remap_input(b.ravel())

  ops = _ops(device_id)
  pa, pb, cnt, slots = ops.pair_counts(ra, rb)
  # np.unique order of (a | b << 32): ascending id_b, then id_a
  order = np.lexsort((pa, pb))
  pa, pb, cnt, slots = pa[order], pb[order], cnt[order], slots[order]
  m = pa.size

  # partner of each id_a = the id_b with the largest overlap; the first such
  # pair in iteration order wins ties (strict `<` at segmentation.py:272)
  by_a = np.lexsort((np.arange(m), -cnt.astype(np.int64), pa))
  firsts = np.ones(m, bool)
  firsts[1:] = pa[by_a][1:] != pa[by_a][:-1]
  best_a = pa[by_a][firsts]
  best_b = pb[by_a][firsts]
  partner = best_b[np.searchsorted(best_a, pa)]

  new_labels = np.zeros(m, np.uint64)
  alive = (cnt >= np.uint64(max(int(min_size), 0))) & (pa != 0)
  keeps = alive & (pb == partner)
  new_labels[keeps] = (a_reverse_map[pa[keeps]] if a_reverse_map is not None
                       else pa[keeps])
  fresh = alive & ~keeps
  new_labels[fresh] = np.uint64(max_id) + np.arange(
      1, int(fresh.sum()) + 1, dtype=np.uint64)

  out = ops.apply_pair_labels(slots, new_labels)
  a[...] = out.reshape(a.shape)
