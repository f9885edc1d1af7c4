"""Seed policies: iterators over candidate start points (z, y, x).

Mirror of reference ffn/inference/seed.py -- every policy a request can name:
`BaseSeedPolicy` (:37-130), `PolicyPeaks` (:142-199), `PolicyPeaks2d`
(:202-279), `PolicyFillEmptySpace` (:282-302), `PolicyMax` (:305-311),
`PolicyMaxPeaks` (:314-353), `PolicyImagePeaks3D2D` (:356-380),
`PolicyImagePeaks2DDisk` (:383-408), `PolicyGrid3d` (:411-430), `PolicyGrid2d`
(:433-450), `PolicyInvertOrigins` (:453-469), `PolicyDenseSeeds` (:472-493),
`ReverseCoords` (:496-505), `SequentialPolicies` (:508-549).  Those outside
SURVEY.md 8(a) row a17 are host code (scipy) whose un-vendored skimage / edt
calls are restated below (`peak_local_max`, `_edt`); all are pinned by
tests/golden/ref_seed_policies.npz, minted by the reference's own classes with
scikit-image 0.18.3 (tools/make_golden_seed_policies.py).  `PolicyPeaks'
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

import itertools
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


def _ensure_spacing(coords, spacing, p_norm):
  """skimage._shared.coord.ensure_spacing: walking the peaks from the highest
  down, a peak drops every other one closer than `spacing` (strictly; Minkowski
  p-norm) to it."""
  if len(coords) < 2 or spacing <= 0:
    return coords
  from scipy.spatial import cKDTree  # pylint:disable=g-import-not-at-top
  near = cKDTree(coords).query_ball_point(coords, r=spacing, p=p_norm)
  rejected = set()
  for i, cand in enumerate(near):
    if i in rejected:
      continue
    for j in cand:
      if j == i:
        continue
      d = np.abs(coords[j] - coords[i]).astype(np.float64)
      dist = d.max() if np.isinf(p_norm) else (d**p_norm).sum()**(1.0 / p_norm)
      if dist < spacing:
        rejected.add(j)
  return np.delete(coords, sorted(rejected), axis=0)


def peak_local_max(image, min_distance=1, threshold_abs=None, threshold_rel=None,
                   footprint=None, p_norm=np.inf):
  """skimage.feature.peak_local_max (0.18: exclude_border=True, no labels, all
  peaks) -> [N, ndim] coordinates, highest peak first.

  A peak = a voxel equal to the maximum over the (2 min_distance + 1)^ndim box
  (or `footprint`; outside the image counts as 0) and above
  max(threshold_abs or image.min(), threshold_rel * image.max()); peaks within
  min_distance of the border are dropped; then `_ensure_spacing`.  Equal
  intensities are ordered by numpy's unstable sort in skimage, i.e. not defined:
  the order is exact for tie-free images (every caller below but the two
  ImagePeaks policies adds the reference's tie-breaking noise)."""
  image = np.asarray(image)
  if image.size == 0:
    return np.zeros((0, image.ndim), np.intp)
  threshold = threshold_abs if threshold_abs is not None else image.min()
  if threshold_rel is not None:
    threshold = max(threshold, threshold_rel * image.max())
  if footprint is None:
    footprint = np.ones((2 * min_distance + 1,) * image.ndim, bool)
  if footprint.size == 1 or image.size == 1:
    mask = image > threshold
  else:
    mx = ndimage.maximum_filter(image, footprint=footprint, mode='constant')
    mask = image == mx
    if mask.all():  # a constant image has no peak
      mask[...] = False
    mask &= image > threshold
  for axis in range(mask.ndim if min_distance > 0 else 0):
    # exclude_border=True: the border width is min_distance
    edge = [slice(None)] * mask.ndim
    edge[axis] = slice(None, min_distance)
    mask[tuple(edge)] = False
    edge[axis] = slice(-min_distance, None)
    mask[tuple(edge)] = False
  coords = np.transpose(np.nonzero(mask))
  coords = coords[np.argsort(-image[mask], kind='stable')]
  return _ensure_spacing(coords, min_distance, p_norm)


def _find_peaks(distances, **kwargs):
  """seed.py:133-139: peaks of `distances` + the fixed-seed 1e-4 noise."""
  rng = np.random.RandomState(seed=42)
  return peak_local_max(distances + rng.rand(*distances.shape) * 1e-4, **kwargs)


def _edt(mask, sampling=None):
  """edt.edt(mask): exact Euclidean distance of every non-zero voxel to the
  nearest zero voxel (the image border is not background), f32."""
  mask = np.asarray(mask) != 0
  if mask.all():  # no background at all: edt reports "infinitely far"
    return np.full(mask.shape, np.inf, np.float32)
  return ndimage.distance_transform_edt(mask, sampling=sampling).astype(np.float32)


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


class PolicyPeaks2d(BaseSeedPolicy):
  """Points away from edges in every 2-D slice: 2-D Sobel -> adaptive threshold
  -> 2-D EDT -> peaks (seed.py:202-279)."""

  def __init__(self, canvas, min_distance=7, threshold_abs=2.5,
               sort_cmp='ascending', **kwargs):
    super().__init__(canvas, **kwargs)
    self.min_distance = min_distance
    self.threshold_abs = threshold_abs
    self.sort_reverse = sort_cmp.strip().lower().startswith('de')

  def init_coords(self):
    logging.info('2d peaks: starting')
    image = np.asarray(self.canvas.image)
    restrictor = getattr(self.canvas, 'restrictor', None)
    rmask = getattr(restrictor, 'mask', None)
    found = []
    for z in range(image.shape[0]):
      image_2d = image[z, :, :].astype(np.float32)
      edges = ndimage.generic_gradient_magnitude(image_2d, ndimage.sobel)
      thresh_image = np.zeros(edges.shape, dtype=np.float32)
      ndimage.gaussian_filter(edges, 49.0 / 6.0, output=thresh_image,
                              mode='reflect')
      filt_edges = edges > thresh_image
      if rmask is not None:  # masked areas count as edges
        filt_edges[rmask[z, :, :]] = 1
      dt = _edt(1 - filt_edges)
      idxs = _find_peaks(dt, min_distance=self.min_distance,
                         threshold_abs=self.threshold_abs, threshold_rel=0)
      zs = np.full((idxs.shape[0], 1), z, dtype=np.int64)
      found.append(np.concatenate((zs, idxs), axis=1))
    coords = np.concatenate(found) if found else np.zeros((0, 3), np.int64)
    self.coords = np.array(
        sorted([(z, y, x) for z, y, x in coords],
               reverse=self.sort_reverse)).reshape(-1, 3)
    logging.info('2d peaks: found %d total local maxima', self.coords.shape[0])


class PolicyFillEmptySpace(BaseSeedPolicy):
  """Local maxima of the distance transform of the unsegmented space
  (seed.py:282-302): fills what an otherwise complete segmentation left."""

  def init_coords(self):
    logging.info('fill_empty: starting')
    dt = _edt(np.asarray(self.canvas.segmentation) == 0)
    # threshold < 1: no seeds inside segmented areas (dt = 0 there)
    idxs = _find_peaks(dt, min_distance=2, threshold_abs=0.5, threshold_rel=0)
    logging.info('fill_empty: found %d local maxima', idxs.shape[0])
    self.coords = np.array(sorted((z, y, x) for z, y, x in idxs)).reshape(-1, 3)


class PolicyMax(BaseSeedPolicy):
  """All points in the image, sorted by decreasing intensity."""

  def init_coords(self):
    image = np.asarray(self.canvas.image)
    order = np.argsort(image.ravel())[::-1]
    self.coords = np.stack(np.unravel_index(order, image.shape), axis=1)


class PolicyMaxPeaks(BaseSeedPolicy):
  """Local peaks of intensity outside the exclusion mask (seed.py:314-353)."""

  def __init__(self, canvas, min_distance=3, threshold_abs=0, threshold_rel=0,
               **kwargs):
    super().__init__(canvas, **kwargs)
    self.min_distance = min_distance
    self.threshold_abs = threshold_abs
    self.threshold_rel = threshold_rel

  def init_coords(self):
    img = np.asarray(self.canvas.image).astype(np.float32).copy()
    img[self.get_exclusion_mask()] = 0
    idxs = _find_peaks(img, min_distance=self.min_distance,
                       threshold_abs=self.threshold_abs,
                       threshold_rel=self.threshold_rel)
    self.coords = np.array(sorted((z, y, x) for z, y, x in idxs)).reshape(-1, 3)


class PolicyImagePeaks3D2D(BaseSeedPolicy):
  """3-D image peaks followed by per-slice 2-D image peaks (seed.py:356-380);
  highest first within each group, as skimage returns them."""

  def __init__(self, canvas, min_distance_2d=2, min_distance_3d=4, **kwargs):
    super().__init__(canvas, **kwargs)
    self._min_distance_2d = min_distance_2d
    self._min_distance_3d = min_distance_3d

  def init_coords(self):
    img = np.asarray(self.canvas.image)
    coords3d = []
    if self._min_distance_3d >= 0:
      coords3d = peak_local_max(img, min_distance=self._min_distance_3d).tolist()
    coords2d = []
    if self._min_distance_2d >= 0:
      for z in range(img.shape[0]):
        for y, x in peak_local_max(img[z, ...],
                                   min_distance=self._min_distance_2d):
          coords2d.append((z, y, x))
    self.coords = np.array(coords3d + coords2d).reshape(-1, 3)


def _disk(radius):
  """skimage.morphology.disk."""
  r = np.arange(-radius, radius + 1)
  x, y = np.meshgrid(r, r)
  return (x**2 + y**2 <= radius**2).astype(np.uint8)


class PolicyImagePeaks2DDisk(BaseSeedPolicy):
  """Per-slice 2-D image peaks over a disk footprint (seed.py:383-408)."""

  def __init__(self, canvas, min_distance_2d=3, threshold_rel=0.5,
               disk_radius=1, **kwargs):
    super().__init__(canvas, **kwargs)
    self._min_distance_2d = min_distance_2d
    self._threshold_rel = threshold_rel
    self._disk_radius = disk_radius

  def init_coords(self):
    img = np.asarray(self.canvas.image)
    footprint = _disk(self._disk_radius).astype(bool)
    coords = []
    for z in range(img.shape[0]):
      for y, x in peak_local_max(img[z, ...],
                                 min_distance=self._min_distance_2d,
                                 threshold_rel=self._threshold_rel,
                                 footprint=footprint, p_norm=2):
        coords.append((z, y, x))
    self.coords = np.array(coords).reshape(-1, 3)


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

  def __init__(self, canvas, step=16, offsets=(0, 8, 4, 12, 2, 6, 10, 14),
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


class PolicyDenseSeeds(BaseSeedPolicy):
  """Every voxel of the thresholded (optionally inverted, eroded) image, in
  raster order (seed.py:472-493)."""

  def __init__(self, canvas, threshold: float = 0.5, num_erosions: int = 0,
               invert: bool = False, **kwargs):
    super().__init__(canvas, **kwargs)
    self._threshold = threshold
    self._num_erosions = num_erosions
    self._invert = invert

  def init_coords(self):
    x = np.asarray(self.canvas.image) > self._threshold
    if self._invert:
      x = ~x
    if self._num_erosions:
      # skimage.morphology.binary_erosion: the 3^ndim cross, border voxels keep
      # their neighbours outside the image as foreground
      cross = ndimage.generate_binary_structure(x.ndim, 1)
      for _ in range(self._num_erosions):
        x = ndimage.binary_erosion(x, structure=cross, border_value=1)
    self.coords = np.array(np.where(x)).T.reshape(-1, 3)


class ReverseCoords(BaseSeedPolicy):
  """Wraps another policy and reverses its (margin-filtered) seed order
  (seed.py:496-505)."""

  def __init__(self, canvas, policy_to_reverse: str, **policy_kwargs):
    super().__init__(canvas)
    self._policy = globals()[policy_to_reverse](canvas, **policy_kwargs)

  def init_coords(self):
    self.coords = np.array(list(self._policy)[::-1]).reshape(-1, 3)


class SequentialPolicies(BaseSeedPolicy):
  """Chains policies: `policies` = sequence of (policy name, keyword dict)
  (seed.py:508-549)."""

  def __init__(self, canvas, policies, **kwargs):
    del kwargs
    super().__init__(canvas)
    self._policies = [globals()[name](canvas, **kw) for name, kw in policies]

  def init_coords(self):
    self.coords = np.array(list(itertools.chain(*self._policies))).reshape(-1, 3)

  def set_state(self, state):
    """A state of this class is the base class's (coords of the whole chain,
    index): a resumed chain goes on where it stopped.  The reference's own
    get_state (seed.py:537-549) returns a LIST of the chained policies' states
    instead -- which are exhausted as soon as the chain has been built, so a
    run resumed from it seeds nothing more; such a list (a checkpoint written
    by the reference) is accepted and restored the reference's way."""
    if isinstance(state, list):
      for s, p in zip(state, self._policies):
        p.set_state(s)
      self.coords, self.idx = None, 0
      return
    super().set_state(state)
