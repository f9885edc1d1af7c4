#!/opt/conda/bin/python3.9
"""Mints tests/golden/ref_labels.npz with the reference's own label routines.

Must run under /opt/conda/bin/python3.9 (the interpreter with scikit-image
0.18.3).  What is the reference's unmodified code: everything in
ffn/inference/segmentation.py -- split_segmentation_by_intersection (:181-290),
clean_up_and_count (:125-178), clear_dust (:21-63).  What is NOT in the
checkout: connectomics.segmentation.labels.split_disconnected_components (an
un-vendored git dependency, setup.py:42, unpinned @main).  Its published body is
`skimage.measure.label(labels, connectivity=connectivity, background=0)` followed
by a fix-up that only matters when label() merges 0 with something else (it
never does with background=0); that is what the stand-in below calls, with the
real skimage.
"""
import os
import sys

import numpy as np
import skimage.measure  # the real one, before the shim path is added

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'tools', 'ref_shims'))

from connectomics.segmentation import labels as shim_labels  # noqa: E402


def _split_disconnected_components(labels, connectivity=1):
  has_zero = 0 in labels
  fixed = skimage.measure.label(labels, connectivity=connectivity,
                                background=0)
  if has_zero or (not has_zero and 0 in fixed):
    if np.any((fixed == 0) != (labels == 0)):
      fixed[...] += 1
      fixed[labels == 0] = 0
  return np.asarray(fixed, dtype=labels.dtype)


shim_labels.split_disconnected_components = _split_disconnected_components

from ffn.inference import segmentation as ref_seg  # noqa: E402


def voronoi(shape, k, seed, zero_frac=0.0, ids=None):
  rng = np.random.RandomState(seed)
  pts = rng.rand(k, 3) * np.array(shape)
  zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing='ij')
  grid = np.stack([zz, yy, xx], -1).reshape(-1, 1, 3).astype(np.float32)
  d = ((grid - pts[None].astype(np.float32))**2).sum(-1)
  lab = d.argmin(1).reshape(shape) + 1
  if ids is not None:
    lab = np.asarray(ids)[lab - 1]
  lab = lab.astype(np.uint64)
  if zero_frac > 0:
    # smooth-ish zero blobs: threshold a coarse random field
    coarse = rng.rand(*[(s + 7) // 8 for s in shape])
    field = np.kron(coarse, np.ones((8, 8, 8)))[:shape[0], :shape[1], :shape[2]]
    lab[field < zero_frac] = 0
  return lab


def main():
  out = {}
  shape = (32, 40, 48)
  # --- split_segmentation_by_intersection -----------------------------------
  cases = {
      'plain': (voronoi(shape, 25, 1), voronoi(shape, 18, 2), 0),
      'zeros_min50': (voronoi(shape, 25, 3, 0.25), voronoi(shape, 30, 4, 0.2),
                      50),
      'ties': (np.repeat(np.arange(1, 5, dtype=np.uint64), 16).reshape(4, 4, 4),
               np.tile(np.arange(1, 5, dtype=np.uint64), 16).reshape(4, 4, 4),
               0),
      'big_ids': (voronoi(shape, 12, 5, 0.1,
                          ids=2**40 + 7 * np.arange(12, dtype=np.uint64)),
                  voronoi(shape, 9, 6, 0.1,
                          ids=2**33 + np.arange(9, dtype=np.uint64)**2), 20),
      'max_uint32': (voronoi(shape, 6, 7, 0.0,
                             ids=np.array([2**32 - 1, 5, 2**32 - 2, 9, 1, 3],
                                          np.uint64)),
                     voronoi(shape, 5, 8, 0.3), 10),
  }
  for name, (a, b, min_size) in cases.items():
    got = a.copy()
    ref_seg.split_segmentation_by_intersection(got, b, min_size)
    out['split_%s_a' % name] = a
    out['split_%s_b' % name] = b
    out['split_%s_min_size' % name] = np.int64(min_size)
    out['split_%s_out' % name] = got
  # --- clean_up_and_count / split_disconnected_components -------------------
  cc_cases = {
      'conn1': (voronoi(shape, 40, 11, 0.3,
                        ids=(np.arange(40) % 6 + 1).astype(np.uint64)), 1, 0),
      'conn2_min30': (voronoi(shape, 60, 12, 0.35,
                              ids=(np.arange(60) % 4 + 1).astype(np.uint64)),
                      2, 30),
      'conn3_min5': (voronoi((20, 24, 28), 80, 13, 0.45,
                             ids=(np.arange(80) % 3 + 2).astype(np.uint64)),
                     3, 5),
      'no_zero': (voronoi((16, 16, 16), 10, 14,
                          ids=(np.arange(10) % 3 + 1).astype(np.uint64)), 1, 0),
  }
  for name, (seg, conn, min_size) in cc_cases.items():
    if min_size > 0:
      # the reference's clear_dust cannot take uint64 (np.bincount refuses the
      # cast, segmentation.py:46); its dust path is exercised with int64 ids
      seg = seg.astype(np.int64)
    work = seg.copy()
    cc_to_orig, cc_to_count = ref_seg.clean_up_and_count(
        work, True, conn, min_size, compute_id_map=True, compute_counts=True)
    out['cc_%s_in' % name] = seg
    out['cc_%s_connectivity' % name] = np.int64(conn)
    out['cc_%s_min_size' % name] = np.int64(min_size)
    out['cc_%s_out' % name] = work
    ks = sorted(cc_to_orig)
    out['cc_%s_ids' % name] = np.array(ks, np.uint64)
    out['cc_%s_orig' % name] = np.array([cc_to_orig[k] for k in ks], np.uint64)
    out['cc_%s_count' % name] = np.array([cc_to_count[k] for k in ks], np.int64)
    out['cc_%s_plain' % name] = _split_disconnected_components(seg, conn)
  # --- clear_dust -------------------------------------------------------------
  dust = voronoi(shape, 300, 21, 0.2).astype(np.int64)
  out['dust_in'] = dust
  out['dust_out'] = ref_seg.clear_dust(dust.copy(), 150)
  dst = os.path.join(ROOT, 'tests', 'golden', 'ref_labels.npz')
  np.savez_compressed(dst, **out)
  print('wrote', dst, os.path.getsize(dst), 'bytes;', len(out), 'arrays;',
        'skimage', skimage.__version__)


if __name__ == '__main__':
  main()
