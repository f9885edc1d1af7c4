import numpy as np


class BoundingBox:
  """xyz bounding box: start + size (or end)."""

  def __init__(self, start=None, size=None, end=None):
    if start is None:
      start = np.asarray(end) - np.asarray(size)
    self.start = np.asarray(start, dtype=np.int64)
    if size is None:
      size = np.asarray(end) - self.start
    self.size = np.asarray(size, dtype=np.int64)

  @property
  def end(self):
    return self.start + self.size

  def to_slice3d(self):
    return np.index_exp[self.start[2]:self.end[2], self.start[1]:self.end[1],
                        self.start[0]:self.end[0]]

  def intersection(self, other):
    start = np.maximum(self.start, other.start)
    end = np.minimum(self.end, other.end)
    if np.any(end <= start):
      return None
    return BoundingBox(start=start, end=end)
