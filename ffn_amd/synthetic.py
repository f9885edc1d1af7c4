"""Deterministic synthetic uint8 EM-like volumes for benchmarks and tests.

The reference's sample input (training_sample2 grayscale_maps.h5) is not
shipped with the repository, and there is no network, so throughput and parity
are measured on seeded phantoms:

* ``noise_volume``  -- S-noise: uniform uint8 noise, RandomState(0).
* ``cells_volume``  -- S-cells: a Voronoi "cell" phantom: dark membranes between
  bright cell interiors, RandomState(1234).
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage


def noise_volume(shape=(250, 250, 250), seed=0) -> np.ndarray:
  return np.random.RandomState(seed).randint(0, 256, shape).astype(np.uint8)


def cells_volume(shape=(250, 250, 250), seed=1234, cells_per_96cube=12.0,
                 membrane=60.0, interior=160.0, noise_sigma=10.0,
                 membrane_dilate=1, blur_sigma=1.0) -> np.ndarray:
  """Voronoi-membrane phantom, uint8 zyx."""
  rng = np.random.RandomState(seed)
  shape = tuple(int(s) for s in shape)
  n_cells = max(2, int(round(cells_per_96cube * np.prod(shape) / 96.0**3)))
  centers = rng.uniform(0, 1, (n_cells, 3)) * np.array(shape)[None]
  # Nearest-centre labelling in z-slabs to bound memory.
  from scipy.spatial import cKDTree
  tree = cKDTree(centers)
  labels = np.empty(shape, dtype=np.int32)
  yy, xx = np.meshgrid(np.arange(shape[1]), np.arange(shape[2]), indexing='ij')
  plane = np.stack([yy.ravel(), xx.ravel()], axis=1).astype(np.float64)
  for z in range(shape[0]):
    pts = np.concatenate(
        [np.full((plane.shape[0], 1), float(z)), plane], axis=1)
    # (exact nearest neighbours: the worker count does not change the result)
    labels[z] = tree.query(pts, workers=-1)[1].reshape(shape[1], shape[2])
  edge = np.zeros(shape, dtype=bool)
  for axis in range(3):
    d = np.diff(labels, axis=axis) != 0
    sl_lo = [slice(None)] * 3
    sl_hi = [slice(None)] * 3
    sl_lo[axis] = slice(0, -1)
    sl_hi[axis] = slice(1, None)
    edge[tuple(sl_lo)] |= d
    edge[tuple(sl_hi)] |= d
  if membrane_dilate > 0:
    edge = ndimage.binary_dilation(edge, iterations=membrane_dilate)
  vol = np.where(edge, membrane, interior).astype(np.float32)
  vol += rng.normal(0, noise_sigma, shape).astype(np.float32)
  if blur_sigma > 0:
    vol = ndimage.gaussian_filter(vol, blur_sigma)
  return np.clip(np.rint(vol), 0, 255).astype(np.uint8)


def shared_volume(build, path, rank=0, barrier=None):
  """One copy of a synthetic volume for all ranks of a node: rank 0 builds it
  (`build()` -> ndarray) and saves it to `path` (a file in /dev/shm or TMPDIR),
  `barrier()` orders the ranks, the others map it read-only."""
  if barrier is None:  # a single rank: nothing to share
    return build()
  if rank == 0:
    vol = build()
    np.save(path, vol)
    barrier()
    return vol
  barrier()
  return np.load(path, mmap_mode='r')


def normalize(volume_u8: np.ndarray, mean: float = 128.0,
              stddev: float = 33.0) -> np.ndarray:
  """(u8 -> f32 - mean) / stddev, exactly as reference runner.py:383-385."""
  return (volume_u8.astype(np.float32) - mean) / stddev
