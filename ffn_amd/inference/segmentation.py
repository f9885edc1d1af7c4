"""Segmentation helpers on the output path (reference ffn/inference/segmentation.py).

Only `reduce_id_bits` (:66-86) and `clear_dust` (:21-63) are on the hot path's
output side; CC clean-up and split-consensus are "next" rows.
"""

import numpy as np


def clear_dust(data: np.ndarray, min_size: int = 10):
  """Zeroes out segments smaller than `min_size` voxels (in place)."""
  ids, sizes = np.unique(data, return_counts=True)
  small = ids[sizes < min_size]
  if small.size > 0:
    data[np.isin(data, small)] = 0
  return data


def reduce_id_bits(segmentation: np.ndarray):
  """Converts to the smallest unsigned type that holds every id."""
  max_id = segmentation.max()
  for dt in (np.uint8, np.uint16, np.uint32):
    if max_id <= np.iinfo(dt).max:
      return segmentation.astype(dt)
  return segmentation
