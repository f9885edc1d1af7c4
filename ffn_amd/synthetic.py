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


def flood_fill_weights(depth: int, features: int = 32, gain: float = 8.0,
                       decay: float = -1.0, texture: float = 0.2) -> dict:
  """TF-named variables of a ConvStack3DFFNModel of any depth / FoV that floods
  BRIGHT, 26-CONNECTED regions: a constructed stand-in for a trained network
  where none exists (BASELINE configs[4]: depth 18, FoV 41 x 41 x 21 -- the
  reference ships no checkpoint of that shape), with the same architecture and
  arithmetic cost and a behaviour a segmentation can be judged by: on the
  `cells_volume` phantom every flood stays inside its cell.

  Two channels of the residual stream carry everything (the others stay 0):
    X0 = clamp(16 image - 6)   "bright"  (normalised image > 0.44), constant
    X1 = clamp(4 seed)         "object"  (seed logit > 0.25)
  with clamp(z) = relu(z) - relu(z - 1).  Every residual module dilates the
  object by one voxel inside the bright region:
    X1 <- clamp(boxsum3x3x3(X1) + 27 X0 - 27)
  (conv_a: u1 = relu(z), u2 = relu(z - 1), u3 = relu(X1); conv_b: u1 - u2 - u3
  added to the skip), so one FoV step grows the object by depth - 1 voxels, and
  the head adds `gain + decay` to the logits of object voxels and `decay` to the
  others -- plus `texture` x clamp(0.5 + a seeded random 3x3x3 filter of the
  image), a third channel carried through the stack untouched: without it whole faces of
  the FoV would hold EQUAL logits and the move policy's argmax would be decided
  by the last bit of each kernel's rounding."""
  assert features >= 6 and depth >= 2
  f = features
  v = {}

  def conv(name, cin, cout, k=3):
    v['seed_update/%s/weights' % name] = np.zeros((k, k, k, cin, cout), np.float32)
    v['seed_update/%s/biases' % name] = np.zeros((cout,), np.float32)
    return v['seed_update/%s/weights' % name], v['seed_update/%s/biases' % name]

  w, b = conv('conv0_a', 2, f)       # input channel 0 = image, 1 = seed
  w[1, 1, 1, 0, 0] = 16.0; b[0] = -6.0  # relu(16 img - 6)
  w[1, 1, 1, 0, 1] = 16.0; b[1] = -7.0  # relu(16 img - 7)
  w[1, 1, 1, 1, 2] = 4.0; b[2] = 0.0    # relu(4 seed)
  w[1, 1, 1, 1, 3] = 4.0; b[3] = -1.0   # relu(4 seed - 1)
  # texture: clamp(0.5 + a seeded random 3x3x3 filter of the image), in [0, 1]
  tex = np.random.RandomState(5).normal(0, 0.03, (3, 3, 3))
  w[:, :, :, 0, 4] = tex; b[4] = 0.5
  w[:, :, :, 0, 5] = tex; b[5] = -0.5
  w, b = conv('conv0_b', f, f)       # linear: the two clamps
  w[1, 1, 1, 0, 0] = 1.0
  w[1, 1, 1, 1, 0] = -1.0
  w[1, 1, 1, 2, 1] = 1.0
  w[1, 1, 1, 3, 1] = -1.0
  w[1, 1, 1, 4, 4] = 1.0
  w[1, 1, 1, 5, 4] = -1.0
  for i in range(1, depth):
    w, b = conv('conv%d_a' % i, f, f)
    for out, bias in ((1, -27.0), (2, -28.0)):
      w[:, :, :, 1, out] = 1.0       # box sum of the object channel
      w[1, 1, 1, 0, out] = 27.0      # + 27 bright
      b[out] = bias
    w[1, 1, 1, 1, 3] = 1.0           # relu(X1) = X1
    w, b = conv('conv%d_b' % i, f, f)
    w[1, 1, 1, 1, 1] = 1.0
    w[1, 1, 1, 2, 1] = -1.0
    w[1, 1, 1, 3, 1] = -1.0
  w, b = conv('conv_lom', f, 1, k=1)
  w[0, 0, 0, 1, 0] = gain
  w[0, 0, 0, 4, 0] = texture
  b[0] = decay
  return v
