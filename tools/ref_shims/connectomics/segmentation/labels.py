import numpy as np


def make_contiguous(seg):
  ids, inv = np.unique(seg, return_inverse=True)
  new = np.arange(len(ids))
  if ids[0] != 0:
    new = new + 1
  return new[inv].reshape(seg.shape).astype(seg.dtype), list(zip(ids, new))


def split_disconnected_components(seg, connectivity=1):
  raise NotImplementedError
