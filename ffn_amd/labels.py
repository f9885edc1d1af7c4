"""Python handle over the label operations of libffn_hip.so (include/ffn_labels.h).

Integer label volumes go to the GPU, HBM-bound kernels do the per-voxel work
(joint histograms, table relabelling, connected components), the small
per-label tables come back and are finished in numpy.  No CPU fallback: without
the library / a GPU every call raises.
"""

from __future__ import annotations

import ctypes
import threading
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import check


def _as_labels(arr: np.ndarray) -> np.ndarray:
  """C-contiguous 4- or 8-byte integer view/copy of `arr`."""
  arr = np.asarray(arr)
  if arr.dtype.kind not in 'iu':
    raise TypeError('label arrays must be integer, got %s' % arr.dtype)
  if arr.dtype.itemsize < 4:
    arr = arr.astype(np.uint32)
  return np.ascontiguousarray(arr)


class LabelOps:
  """One stream + grow-only device scratch for label kernels on one GPU."""

  def __init__(self, device_id: int = 0):
    self._lib = _lib.load()
    self._h = ctypes.c_void_p()
    self.device_id = int(device_id)
    check(self._lib.ffn_labels_create(self.device_id, ctypes.byref(self._h)))
    self._lock = threading.Lock()
    self._resident = None  # (shape, dtype) of the volumes of the pair table

  def close(self):
    if self._h:
      self._lib.ffn_labels_destroy(self._h)
      self._h = ctypes.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint:disable=broad-except
      pass

  # -- joint histogram + relabel by pair ---------------------------------------
  def pair_counts(self, a: np.ndarray, b: Optional[np.ndarray] = None):
    """Unique (a[i], b[i]) pairs and their voxel counts, unsorted.

    Returns (pair_a, pair_b, counts, slots) as uint64/uint64/uint64/uint32
    arrays; `slots` feeds apply_pair_labels.  Ids must be < 2**32 - 1.
    """
    a = _as_labels(a)
    if b is not None:
      b = _as_labels(b)
      if b.shape != a.shape:
        raise ValueError('shape mismatch')
      if b.dtype.itemsize != a.dtype.itemsize:
        wide = np.uint64 if 8 in (a.dtype.itemsize, b.dtype.itemsize) else None
        a, b = a.astype(wide), b.astype(wide)
    n = a.size
    cap = max(min(n, 1 << 22), 1)
    while True:
      pa = np.empty(cap, np.uint64)
      pb = np.empty(cap, np.uint64)
      pc = np.empty(cap, np.uint64)
      ps = np.empty(cap, np.uint32)
      found = ctypes.c_size_t(0)
      rc = self._lib.ffn_labels_pair_counts(
          self._h, a.ctypes.data, b.ctypes.data if b is not None else None,
          a.dtype.itemsize, n, cap, pa.ctypes.data, pb.ctypes.data,
          pc.ctypes.data, ps.ctypes.data, ctypes.byref(found))
      if rc != 0 and found.value > cap:
        cap = found.value
        continue
      check(rc)
      break
    m = found.value
    self._resident = (a.shape, a.dtype)
    return pa[:m], pb[:m], pc[:m], ps[:m]

  def apply_pair_labels(self, slots: np.ndarray, new_labels: np.ndarray):
    """Volume whose voxel i carries new_labels[k] of its pair k (see pair_counts)."""
    if self._resident is None:
      raise _lib.FFNHipError('apply_pair_labels without a resident pair table')
    shape, dtype = self._resident
    slots = np.ascontiguousarray(slots, np.uint32)
    new_labels = np.ascontiguousarray(new_labels, np.uint64)
    if slots.shape != new_labels.shape:
      raise ValueError('slots / new_labels length mismatch')
    out = np.empty(shape, dtype)
    check(self._lib.ffn_labels_apply_pair_labels(
        self._h, slots.size, slots.ctypes.data, new_labels.ctypes.data,
        out.ctypes.data))
    return out

  # -- table relabel -------------------------------------------------------------
  def remap(self, arr: np.ndarray, keys, values, keep_missing: bool = True):
    """arr with every id in `keys` replaced by the matching `values` entry;
    other ids are kept (keep_missing) or zeroed."""
    src = _as_labels(arr)
    keys = np.ascontiguousarray(keys, np.uint64)
    values = np.ascontiguousarray(values, np.uint64)
    if keys.shape != values.shape:
      raise ValueError('keys / values length mismatch')
    out = np.empty(src.shape, src.dtype)
    check(self._lib.ffn_labels_remap(
        self._h, src.ctypes.data, src.dtype.itemsize, src.size, keys.size,
        keys.ctypes.data, values.ctypes.data, 1 if keep_missing else 0,
        out.ctypes.data))
    self._resident = None
    return out

  # -- connected components --------------------------------------------------------
  def connected_components(self, arr: np.ndarray, connectivity: int = 1,
                           stats: bool = False):
    """Components of equal non-zero label, ids 1.. in raster order of their
    first voxel.  With stats: (out, first_index, sizes, first_zero_index)."""
    src = _as_labels(arr)
    if src.ndim != 3:
      raise ValueError('connected_components expects a 3d array')
    out = np.empty(src.shape, src.dtype)
    shape = (ctypes.c_int64 * 3)(*src.shape)
    ncomp = ctypes.c_uint64(0)
    fz = ctypes.c_int64(-1)
    cap = 1 << 20 if stats else 0
    while True:
      first = np.empty(max(cap, 1), np.uint64) if stats else None
      sizes = np.empty(max(cap, 1), np.uint64) if stats else None
      rc = self._lib.ffn_labels_connected_components(
          self._h, src.ctypes.data, src.dtype.itemsize, shape,
          int(connectivity), out.ctypes.data, ctypes.byref(ncomp), cap,
          first.ctypes.data if stats else None,
          sizes.ctypes.data if stats else None, ctypes.byref(fz))
      if rc != 0 and stats and ncomp.value > cap:
        cap = int(ncomp.value)
        continue
      check(rc)
      break
    self._resident = None
    if not stats:
      return out
    k = int(ncomp.value)
    return out, first[:k], sizes[:k], int(fz.value)

  # -- device-resident assembly (raw device pointers, int32 labels) ----------------
  @staticmethod
  def _i64x3(v):
    return (ctypes.c_int64 * 3)(*[int(x) for x in v])

  def copy_device(self, src_ptr: int, n: int, dst_ptr: int):
    """dst[i] = max(src[i], 0): a canvas segmentation without its -1 markers."""
    check(self._lib.ffn_labels_copy_device(self._h, ctypes.c_void_p(src_ptr),
                                           int(n), ctypes.c_void_p(dst_ptr)))

  def copy_canvas(self, canvas_handle, dst_ptr: int):
    """The segmentation of a live device canvas (ffn_canvas*), negatives
    dropped, into a device buffer -- no trip through the host."""
    check(self._lib.ffn_labels_copy_canvas(self._h, canvas_handle,
                                           ctypes.c_void_p(dst_ptr)))

  def place_core_device(self, src_ptr, src_shape, core_lo, core_hi, id_offset,
                        dst_ptr, dst_shape, corner):
    """Core of a sub-box (ids + id_offset) into the assembled device volume."""
    check(self._lib.ffn_labels_place_core_device(
        self._h, ctypes.c_void_p(src_ptr), self._i64x3(src_shape),
        self._i64x3(core_lo), self._i64x3(core_hi), int(id_offset),
        ctypes.c_void_p(dst_ptr), self._i64x3(dst_shape), self._i64x3(corner)))

  def margin_pairs_device(self, own_ptr, own_shape, id_offset, core_lo, core_hi,
                          assembled_ptr, assembled_shape, corner):
    """(own id + offset, assembled id, voxels) over a sub-box's margin."""
    cap = 1 << 16
    while True:
      pa = np.empty(cap, np.uint64)
      pb = np.empty(cap, np.uint64)
      pc = np.empty(cap, np.uint64)
      found = ctypes.c_size_t(0)
      rc = self._lib.ffn_labels_margin_pairs_device(
          self._h, ctypes.c_void_p(own_ptr), self._i64x3(own_shape),
          int(id_offset), self._i64x3(core_lo), self._i64x3(core_hi),
          ctypes.c_void_p(assembled_ptr), self._i64x3(assembled_shape),
          self._i64x3(corner), cap, pa.ctypes.data, pb.ctypes.data,
          pc.ctypes.data, ctypes.byref(found))
      if rc != 0 and found.value > cap:
        cap = found.value
        continue
      check(rc)
      break
    self._resident = None
    m = found.value
    return pa[:m], pb[:m], pc[:m]

  def remap_device(self, vol_ptr: int, n: int, keys, values):
    """In place on a device int32 volume: ids in `keys` -> `values`."""
    keys = np.ascontiguousarray(keys, np.uint64)
    values = np.ascontiguousarray(values, np.uint64)
    check(self._lib.ffn_labels_remap_device(
        self._h, ctypes.c_void_p(vol_ptr), int(n), keys.size, keys.ctypes.data,
        values.ctypes.data))
    self._resident = None

  def last_timing(self) -> Tuple[float, float]:
    """(kernel milliseconds, algorithmic HBM bytes) of the last call."""
    ms = ctypes.c_double(0)
    nbytes = ctypes.c_double(0)
    check(self._lib.ffn_labels_last_timing(self._h, ctypes.byref(ms),
                                           ctypes.byref(nbytes)))
    return ms.value, nbytes.value


_default = {}
_default_lock = threading.Lock()


def default_ops(device_id: int = 0) -> LabelOps:
  """Process-wide LabelOps of a device (created on first use)."""
  with _default_lock:
    ops = _default.get(device_id)
    if ops is None:
      ops = LabelOps(device_id)
      _default[device_id] = ops
    return ops


import atexit  # pylint:disable=wrong-import-position


@atexit.register
def _close_default_ops():
  # release device objects while the HIP runtime is still alive
  for ops in list(_default.values()):
    try:
      ops.close()
    except Exception:  # pylint:disable=broad-except
      pass
  _default.clear()
