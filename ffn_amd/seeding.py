"""Python handle over the GPU seed generator (include/ffn_seeds.h).

`Seeder.peaks*` = PolicyPeaks.init_coords of the reference
(ffn/inference/seed.py:153-199) as a chain of HBM-bound HIP kernels that
reproduce scipy's arithmetic bit for bit.  The host contributes only what must
be the reference's own numbers: the Mersenne-Twister noise (seed 42) and the
gaussian taps, both computed with numpy exactly as scipy does.
"""

from __future__ import annotations

import atexit
import ctypes
import threading

import numpy as np

from . import _lib
from ._lib import check

SIGMA = 49.0 / 6.0  # seed.py:163


def gaussian_weights(sigma: float = SIGMA, truncate: float = 4.0):
  """The taps ndimage.gaussian_filter1d hands to correlate1d (scipy
  _filters._gaussian_kernel1d, order 0), and their radius."""
  sd = float(sigma)
  lw = int(truncate * sd + 0.5)
  x = np.arange(-lw, lw + 1)
  phi = np.exp(-0.5 / (sd * sd) * x**2)
  phi = phi / phi.sum()
  return np.ascontiguousarray(phi[::-1]), lw


class Seeder:
  """One stream + grow-only device scratch for PolicyPeaks on one GPU."""

  def __init__(self, device_id: int = 0):
    self._lib = _lib.load()
    self._h = ctypes.c_void_p()
    self.device_id = int(device_id)
    check(self._lib.ffn_seeder_create(self.device_id, ctypes.byref(self._h)))
    w, r = gaussian_weights()
    check(self._lib.ffn_seeder_set_gaussian(self._h, w.ctypes.data, r))
    self._noise_n = 0
    self._lock = threading.Lock()

  def close(self):
    if self._h:
      self._lib.ffn_seeder_destroy(self._h)
      self._h = ctypes.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint:disable=broad-except
      pass

  def _ensure_noise(self, n: int):
    if n <= self._noise_n:
      return
    # RandomState(42).rand(*shape) fills in C order, so the first n values of
    # a longer draw are the draw for n voxels (seed.py:136-138).
    noise = np.random.RandomState(seed=42).rand(n)
    check(self._lib.ffn_seeder_set_noise(self._h, noise.ctypes.data, n))
    self._noise_n = n

  def _collect(self, call, n: int):
    cap = 1 << 16
    while True:
      coords = np.empty((cap, 3), np.int32)
      found = ctypes.c_size_t(0)
      all_edges = ctypes.c_int32(0)
      rc = call(cap, coords.ctypes.data, ctypes.byref(found),
                ctypes.byref(all_edges))
      if rc != 0 and found.value > cap:
        cap = int(found.value)
        continue
      check(rc)
      break
    if all_edges.value:
      return None
    coords = coords[:found.value].astype(np.int64)
    order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
    return coords[order]  # ascending (z, y, x), seed.py:193

  def peaks(self, image, exclusion_mask=None, force_edge=None,
            voxel_size_zyx=(1, 1, 1)):
    """Sorted [N, 3] zyx peaks for host arrays, or None if every voxel is an
    edge (the reference then yields no seeds)."""
    image = np.ascontiguousarray(image, np.float32)
    if image.ndim != 3:
      raise ValueError('peaks expects a 3d image')
    shape = (ctypes.c_int64 * 3)(*image.shape)
    voxel = (ctypes.c_double * 3)(*[float(v) for v in voxel_size_zyx])
    ex = (None if exclusion_mask is None else
          np.ascontiguousarray(exclusion_mask, np.uint8))
    fe = (None if force_edge is None else
          np.ascontiguousarray(force_edge, np.uint8))
    with self._lock:
      self._ensure_noise(image.size)
      return self._collect(
          lambda cap, coords, found, all_edges: self._lib.ffn_seeder_peaks(
              self._h, image.ctypes.data,
              ex.ctypes.data if ex is not None else None,
              fe.ctypes.data if fe is not None else None, shape, voxel, cap,
              coords, found, all_edges), image.size)

  def peaks_canvas(self, canvas_handle, voxel_size_zyx=(1, 1, 1)):
    """The same on a device-resident canvas (engine.DeviceCanvasHandle): image
    and segmentation are read in HBM, only the peak list comes back."""
    voxel = (ctypes.c_double * 3)(*[float(v) for v in voxel_size_zyx])
    n = int(np.prod(canvas_handle.shape))
    with self._lock:
      self._ensure_noise(n)
      return self._collect(
          lambda cap, coords, found, all_edges:
          self._lib.ffn_seeder_peaks_canvas(
              self._h, canvas_handle._h, voxel, cap, coords, found, all_edges),
          n)

  def edt(self, mask, voxel_size_zyx=(1, 1, 1)):
    """scipy.ndimage.distance_transform_edt(mask, sampling=voxel_size_zyx) on
    the GPU (exact): f64 distances to the nearest voxel where `mask` is 0."""
    m = np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
    if m.ndim != 3:
      raise ValueError('edt expects a 3d mask')
    out = np.empty(m.shape, np.float64)
    shape = (ctypes.c_int64 * 3)(*m.shape)
    voxel = (ctypes.c_double * 3)(*[float(v) for v in voxel_size_zyx])
    with self._lock:
      check(self._lib.ffn_seeder_edt(self._h, m.ctypes.data, shape, voxel,
                                     out.ctypes.data))
    return out

  def read_stage(self, which: int, shape):
    out = np.empty(shape, np.float32)
    check(self._lib.ffn_seeder_read_stage(self._h, which, out.ctypes.data))
    return out

  def last_timing(self):
    ms = ctypes.c_double(0)
    vox = ctypes.c_double(0)
    check(self._lib.ffn_seeder_last_timing(self._h, ctypes.byref(ms),
                                           ctypes.byref(vox)))
    return ms.value, vox.value


_default = {}
_default_lock = threading.Lock()


def default_seeder(device_id: int = 0) -> Seeder:
  with _default_lock:
    s = _default.get(device_id)
    if s is None:
      s = Seeder(device_id)
      _default[device_id] = s
    return s


@atexit.register
def _close_default_seeders():
  for s in list(_default.values()):
    try:
      s.close()
    except Exception:  # pylint:disable=broad-except
      pass
  _default.clear()
