"""Shim for the un-vendored `edt` package: exact Euclidean distance transform
via scipy (same mathematical definition; used only by tools/make_golden*.py)."""
import numpy as np
from scipy import ndimage


def edt(data, anisotropy=None, **kwargs):
  del kwargs
  return ndimage.distance_transform_edt(np.asarray(data) != 0,
                                        sampling=anisotropy)
