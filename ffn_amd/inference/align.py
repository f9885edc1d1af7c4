"""Identity alignment (reference ffn/inference/align.py:27-172).

The reference ships only the no-op `Alignment` / `Aligner`; subvolumes are
cropped / padded but never warped.  Same interface, numpy >= 1.24 clean (the
reference still uses the removed `np.int`, align.py:112-115).
"""

import numpy as np


class Alignment:
  """Identity transform between source and destination subvolumes."""

  def __init__(self, corner, size):
    self._corner = corner
    self._size = size

  @property
  def corner(self):
    return self._corner

  @property
  def size(self):
    return self._size

  def expand_bounds(self, corner, size, forward=True):
    del forward
    return corner, size

  def transform_shift_mask(self, corner, scale, mask):
    del corner, scale
    return mask

  def align_and_crop(self, src_corner, source, dst_corner, dst_size, fill=0,
                     forward=True):
    """Copies the overlap of [src_corner, +source.shape) into a dst_size box."""
    del forward
    if source is None:
      return None
    if (np.all(np.array(src_corner) == np.array(dst_corner)) and
        np.all(np.array(source.shape) == np.array(dst_size))):
      return source
    destination = np.full(dst_size, fill, dtype=source.dtype)
    zyx_offset = np.array(src_corner) - np.array(dst_corner)
    src_size = np.array(source.shape)
    dst_beg = np.clip(zyx_offset, 0, dst_size).astype(int)
    dst_end = np.clip(zyx_offset + src_size, 0, dst_size).astype(int)
    src_beg = np.clip(-zyx_offset, 0, src_size).astype(int)
    src_end = src_beg + (dst_end - dst_beg)
    if np.any(dst_end - dst_beg <= 0):
      return destination
    destination[dst_beg[0]:dst_end[0], dst_beg[1]:dst_end[1],
                dst_beg[2]:dst_end[2]] = source[src_beg[0]:src_end[0],
                                               src_beg[1]:src_end[1],
                                               src_beg[2]:src_end[2]]
    return destination

  def transform(self, zyx, forward=True):
    del forward
    return zyx

  def rescaled(self, zyx_scale):
    zyx_scale = np.array(zyx_scale)
    return Alignment(zyx_scale * self.corner, zyx_scale * self.size)


class Aligner:
  """Generates (identity) alignments for subvolumes."""

  def generate_alignment(self, corner, size):
    return Alignment(corner, size)
