"""Seed policies: iterators over candidate start points (z, y, x).

Mirror of reference ffn/inference/seed.py: `BaseSeedPolicy` (:37-130),
`PolicyPeaks` (:142-199), `PolicyMax` (:330-362), `PolicyGrid3d` (:411-430),
`PolicyGrid2d` (:433-450), `PolicyInvertOrigins` (:453-475).  Seed generation
stays on the CPU (moving it to the GPU is a "next" row).

`PolicyPeaks` in the reference depends on the un-vendored `edt` and
`skimage.feature.peak_local_max`; neither is installable here.  It is restated
with scipy (`distance_transform_edt`, `maximum_filter`): same pipeline (3-D
Sobel magnitude -> gaussian adaptive threshold sigma=49/6 -> EDT -> local maxima
with min_distance=3 and the fixed-seed 1e-4 noise -> ascending sort), and the
result is pinned against the reference's own PolicyPeaks run with the real
scikit-image 0.18.3 (tests/golden/ref_policy_peaks.npz, minted by
tools/make_golden_peaks.py under the image's conda python; `edt` there is
scipy's exact EDT).  Benchmarks use `PolicyGrid3d` so CPU and GPU runs consume
identical seeds by construction.
"""

from __future__ import annotations

import logging
import threading
import weakref

import numpy as np
from scipy import ndimage


class BaseSeedPolicy:
  """Base class for seed policies."""

  def __init__(self, canvas, **kwargs):
    del kwargs
    self.canvas = weakref.proxy(canvas)
    self.coords = None  # [N, 3] zyx
    self.idx = 0

  def init_coords(self):
    raise NotImplementedError()

  def __iter__(self):
    return self

  def __next__(self):
    """Next seed as (z, y, x); seeds too close to the border are dropped early."""
    if self.coords is None:
      self.init_coords()
      if self.coords is None:
        raise StopIteration()
      if self.coords.size:
        margin = np.array(self.canvas.margin)[np.newaxis, ...]
        self.coords = self.coords[np.all(
            (self.coords - margin >= 0) &
            (self.coords + margin < self.canvas.shape), axis=1), :]
    while self.idx < self.coords.shape[0]:
      curr = self.coords[self.idx, :]
      self.idx += 1
      return tuple(int(v) for v in curr)
    raise StopIteration()

  def next(self):
    return self.__next__()

  def get_state(self, previous=False):
    if previous:
      return self.coords, max(0, self.idx - 1)
    return self.coords, self.idx

  def set_state(self, state):
    self.coords, self.idx = state

  def get_exclusion_mask(self):
    """Voxels that are invalid for seeds (segmented or masked)."""
    mask = np.asarray(self.canvas.segmentation) > 0
    if self.canvas.restrictor is not None:
      if self.canvas.restrictor.mask is not None:
        mask |= self.canvas.restrictor.mask
      if self.canvas.restrictor.seed_mask is not None:
        mask |= self.canvas.restrictor.seed_mask
    return mask


class PolicyFixed(BaseSeedPolicy):
  """Explicit seed list (used by benchmarks / tests for identical seeds)."""

  def __init__(self, canvas, coords=None, **kwargs):
    super().__init__(canvas, **kwargs)
    self._fixed = np.array(coords, dtype=np.int64).reshape(-1, 3)

  def init_coords(self):
    self.coords = self._fixed


def _peak_local_max(dist, min_distance=3):
  """Local maxima of `dist` (> 0) at least `min_distance` apart: what
  skimage.feature.peak_local_max(min_distance, threshold_abs=0,
  threshold_rel=0) returns for tie-free input -- maximum filter over the
  (2*min_distance+1) cube, peaks within `min_distance` of the border excluded
  (skimage's default exclude_border=True)."""
  size = 2 * min_distance + 1
  mx = ndimage.maximum_filter(dist, size=size, mode='constant', cval=0.0)
  peaks = (dist == mx) & (dist > 0)
  border = np.zeros_like(peaks)
  inner = tuple(slice(min_distance, max(n - min_distance, min_distance))
                for n in dist.shape)
  border[inner] = True
  return np.argwhere(peaks & border)


class PolicyPeaks(BaseSeedPolicy):
  """Points away from edges: Sobel -> adaptive threshold -> EDT -> peaks."""

  _sem = threading.Semaphore(4)

  def init_coords(self):
    logging.info('peaks: starting')
    image = np.asarray(self.canvas.image).astype(np.float32)
    edges = ndimage.generic_gradient_magnitude(image, ndimage.sobel)
    sigma = 49.0 / 6.0
    thresh_image = np.zeros(edges.shape, dtype=np.float32)
    ndimage.gaussian_filter(edges, sigma, output=thresh_image, mode='reflect')
    filt_edges = edges > thresh_image
    del edges, thresh_image
    mask = self.get_exclusion_mask()
    if self.canvas.restrictor is not None:
      if self.canvas.restrictor.mask is not None:
        filt_edges[self.canvas.restrictor.mask] = 1
      if self.canvas.restrictor.seed_mask is not None:
        filt_edges[self.canvas.restrictor.seed_mask] = 1
    if np.all(filt_edges == 1):
      return
    with PolicyPeaks._sem:
      dt = ndimage.distance_transform_edt(
          1 - filt_edges,
          sampling=self.canvas.voxel_size_zyx).astype(np.float32)
      dt[mask] = -1
      dt[~np.isfinite(dt)] = -1
      rng = np.random.RandomState(seed=42)
      # f32 dt + f64 noise -> f64, as in the reference (seed.py:136-138); in f32
      # the 1e-4 noise would collapse into ties
      idxs = _peak_local_max(dt + rng.rand(*dt.shape) * 1e-4, min_distance=3)
      idxs = np.array(sorted((z, y, x) for z, y, x in idxs)).reshape(-1, 3)
      logging.info('peaks: found %d local maxima', idxs.shape[0])
      self.coords = idxs


class PolicyMax(BaseSeedPolicy):
  """All points in the image, sorted by decreasing intensity."""

  def init_coords(self):
    image = np.asarray(self.canvas.image)
    order = np.argsort(image.ravel())[::-1]
    self.coords = np.stack(np.unravel_index(order, image.shape), axis=1)


class PolicyGrid3d(BaseSeedPolicy):
  """Points distributed on a uniform 3d grid."""

  def __init__(self, canvas, step=16, offsets=(0, 8, 4, 12, 2, 10, 14),
               **kwargs):
    super().__init__(canvas, **kwargs)
    self.step = step
    self.offsets = offsets

  def init_coords(self):
    shape = self.canvas.image.shape
    coords = []
    for offset in self.offsets:
      for z in range(offset, shape[0], self.step):
        for y in range(offset, shape[1], self.step):
          for x in range(offset, shape[2], self.step):
            coords.append((z, y, x))
    self.coords = np.array(coords).reshape(-1, 3)


class PolicyGrid2d(BaseSeedPolicy):
  """Points distributed on a uniform 2d grid, every z slice."""

  def __init__(self, canvas, step=16, offsets=(0, 8, 4, 12, 2, 10, 14),
               **kwargs):
    super().__init__(canvas, **kwargs)
    self.step = step
    self.offsets = offsets

  def init_coords(self):
    shape = self.canvas.image.shape
    coords = []
    for offset in self.offsets:
      for z in range(shape[0]):
        for y in range(offset, shape[1], self.step):
          for x in range(offset, shape[2], self.step):
            coords.append((z, y, x))
    self.coords = np.array(coords).reshape(-1, 3)


class PolicyInvertOrigins(BaseSeedPolicy):
  """Re-seeds from the origins of an existing segmentation, in reverse."""

  def __init__(self, canvas, corner=None, segmentation_dir=None, **kwargs):
    super().__init__(canvas, **kwargs)
    self.corner = corner
    self.segmentation_dir = segmentation_dir

  def init_coords(self):
    from . import storage  # pylint:disable=g-import-not-at-top
    origins_to_invert = storage.load_origins(self.segmentation_dir,
                                             self.corner)
    points = origins_to_invert.items()
    points = sorted(points, reverse=True)
    self.coords = np.array([origin_info.start_zyx
                            for _, origin_info in points]).reshape(-1, 3)
