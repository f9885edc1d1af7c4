"""Seed policies: iterators over candidate start points (z, y, x).

Mirror of reference ffn/inference/seed.py: `BaseSeedPolicy` (:37-130),
`PolicyPeaks` (:142-199), `PolicyMax` (:330-362), `PolicyGrid3d` (:411-430),
`PolicyGrid2d` (:433-450), `PolicyInvertOrigins` (:453-475).  `PolicyPeaks`
 runs on the GPU: 3-D Sobel magnitude ->
gaussian adaptive threshold sigma=49/6 -> exact EDT -> local maxima with
min_distance=3 and the fixed-seed 1e-4 noise -> ascending sort, bit-identical
to the reference's scipy / edt / skimage pipeline (pinned through
tests/golden/ref_policy_peaks.npz, minted by the reference's own PolicyPeaks
with scikit-image 0.18.3, tools/make_golden_peaks.py).  The throughput
benchmark uses `PolicyGrid3d` so the CPU baseline and the GPU run consume
identical seeds by construction.
"""

from __future__ import annotations

import logging
import threading
import weakref

import numpy as np


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


class PolicyPeaks(BaseSeedPolicy):
  """Points away from edges: 3-D Sobel -> adaptive threshold -> EDT -> peaks
  (reference seed.py:142-199), computed on the GPU.

  The whole pipeline runs as HIP kernels (`ffn_amd.seeding.Seeder`,
  include/ffn_seeds.h) that reproduce the arithmetic of the scipy / edt /
  skimage calls of the reference bit for bit; for a `DeviceCanvas` the image
  and the segmentation are read where they already live, in HBM.  The
  reference spends 7.9 s of CPU per 250^3 subvolume here.
  """

  _sem = threading.Semaphore(4)

  def __init__(self, canvas, seeder=None, **kwargs):
    super().__init__(canvas, **kwargs)
    self._seeder = seeder

  def _get_seeder(self, device_id=0):
    if self._seeder is None:
      from .. import seeding  # pylint:disable=g-import-not-at-top
      self._seeder = seeding.default_seeder(device_id)
    return self._seeder

  def init_coords(self):
    logging.info('peaks: starting')
    canvas = self.canvas
    restrictor = getattr(canvas, 'restrictor', None)
    rmask = getattr(restrictor, 'mask', None)
    smask = getattr(restrictor, 'seed_mask', None)
    voxel = getattr(canvas, 'voxel_size_zyx', (1, 1, 1))
    handle = getattr(canvas, '_handle', None)
    with PolicyPeaks._sem:
      if handle is not None and rmask is None and smask is None:
        engine = getattr(handle, 'engine', None)
        seeder = self._get_seeder(getattr(engine, 'device_id', 0))
        idxs = seeder.peaks_canvas(handle, voxel)
      else:
        force = None  # masked areas count as edges (seed.py:172-176)
        if rmask is not None or smask is not None:
          force = np.zeros(canvas.shape, bool)
          if rmask is not None:
            force |= rmask
          if smask is not None:
            force |= smask
        idxs = self._get_seeder().peaks(
            np.asarray(canvas.image), self.get_exclusion_mask(), force, voxel)
    if idxs is None:  # every voxel is an edge (seed.py:178-179)
      return
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
