"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the ffn_amd product).

CPU restatement of PolicyPeaks.init_coords (reference ffn/inference/seed.py:
133-139, 153-199) for checking the GPU seeder (include/ffn_seeds.h).

Two levels:
* `policy_peaks` calls the very scipy routines the reference calls
  (ndimage.generic_gradient_magnitude + sobel, ndimage.gaussian_filter) and
  stands in for the two un-vendored dependencies with exact equivalents
  (`edt.edt` -> scipy's exact distance_transform_edt; skimage.feature.
  peak_local_max -> maximum_filter + border exclusion).  PINNED by
  tests/golden/ref_policy_peaks.npz, minted by the reference's own PolicyPeaks
  with the real scikit-image 0.18.3 (tools/make_golden_peaks.py).
* `gradient_magnitude_exact` / `gaussian_exact` spell out, in numpy f64, the
  arithmetic ORDER of scipy's NI_Correlate1D (symmetric / anti-symmetric
  branches, f64 accumulation, one f32 rounding per separable pass) that the HIP
  kernels reproduce; tests pin them bit-for-bit against scipy itself.
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage

SIGMA = 49.0 / 6.0


def _reflect(i, n):
  p = 2 * n
  i = np.mod(i, p)
  return np.where(i >= n, p - 1 - i, i)


def _corr3(x, axis, w_center, w_side, anti):
  n = x.shape[axis]
  xd = np.moveaxis(x.astype(np.float64), axis, -1)
  idx = np.arange(n)
  left = xd[..., _reflect(idx - 1, n)]
  right = xd[..., _reflect(idx + 1, n)]
  tmp = xd * w_center
  tmp = tmp + ((left - right) if anti else (left + right)) * w_side
  return np.moveaxis(tmp.astype(np.float32), -1, axis)


def gradient_magnitude_exact(image):
  """ndimage.generic_gradient_magnitude(image, ndimage.sobel) for f32 input
  (scipy/ndimage/_filters.py sobel + generic_gradient_magnitude,
  ni_filters.c NI_Correlate1D)."""
  image = np.asarray(image, np.float32)
  out = None
  for axis in range(3):
    d = _corr3(image, axis, 0.0, -1.0, True)
    for other in range(3):
      if other != axis:
        d = _corr3(d, other, 2.0, 1.0, False)
    d = d * d
    out = d if out is None else out + d
  return np.sqrt(out)


def gaussian_weights(sigma=SIGMA, truncate=4.0):
  """(weights in correlate1d order, radius) as ndimage.gaussian_filter1d
  builds them (_gaussian_kernel1d)."""
  sd = float(sigma)
  lw = int(truncate * sd + 0.5)
  x = np.arange(-lw, lw + 1)
  phi = np.exp(-0.5 / (sd * sd) * x**2)
  phi = phi / phi.sum()
  return np.ascontiguousarray(phi[::-1]), lw


def gaussian_exact(x, sigma=SIGMA):
  """ndimage.gaussian_filter(x, sigma, output=f32, mode='reflect')."""
  w, r = gaussian_weights(sigma)
  x = np.asarray(x, np.float32)
  for axis in range(3):
    n = x.shape[axis]
    xd = np.moveaxis(x.astype(np.float64), axis, -1)
    idx = np.arange(n)
    tmp = xd * w[r]
    for ii in range(-r, 0):
      tmp = tmp + (xd[..., _reflect(idx + ii, n)] +
                   xd[..., _reflect(idx - ii, n)]) * w[ii + r]
    x = np.moveaxis(tmp.astype(np.float32), -1, axis)
  return x


def peak_local_max(dist, min_distance=3):
  """skimage.feature.peak_local_max(min_distance, threshold_abs=0,
  threshold_rel=0) for tie-free input: maximum over the (2*min_distance+1)
  cube, peaks within min_distance of the border excluded."""
  size = 2 * min_distance + 1
  mx = ndimage.maximum_filter(dist, size=size, mode='constant', cval=0.0)
  peaks = (dist == mx) & (dist > 0)
  inner = np.zeros_like(peaks)
  inner[tuple(slice(min_distance, max(n - min_distance, min_distance))
              for n in dist.shape)] = True
  return np.argwhere(peaks & inner)


def policy_peaks(image, exclusion_mask=None, force_edge=None,
                 voxel_size_zyx=(1, 1, 1), stages=None):
  """Sorted [N, 3] zyx seed list of PolicyPeaks.init_coords (seed.py:153-199),
  before the margin filter of BaseSeedPolicy.__next__.  `stages`, if a dict,
  receives the intermediate volumes."""
  image = np.asarray(image).astype(np.float32)
  edges = ndimage.generic_gradient_magnitude(image, ndimage.sobel)
  thresh = np.zeros(edges.shape, dtype=np.float32)
  ndimage.gaussian_filter(edges, SIGMA, output=thresh, mode='reflect')
  filt = edges > thresh
  if force_edge is not None:
    filt[np.asarray(force_edge, bool)] = 1
  if stages is not None:
    stages.update(edges=edges, thresh=thresh, filt=filt.copy())
  if np.all(filt == 1):
    return None
  if not filt.any():
    dt = np.full(filt.shape, -1, np.float32)  # edt: inf without edges -> -1
  else:
    dt = ndimage.distance_transform_edt(
        1 - filt, sampling=voxel_size_zyx).astype(np.float32)
  if exclusion_mask is not None:
    dt[np.asarray(exclusion_mask, bool)] = -1
  dt[~np.isfinite(dt)] = -1
  if stages is not None:
    stages['dt'] = dt.copy()
  rng = np.random.RandomState(seed=42)
  idxs = peak_local_max(dt + rng.rand(*dt.shape) * 1e-4, min_distance=3)
  return np.array(sorted((z, y, x) for z, y, x in idxs)).reshape(-1, 3)
