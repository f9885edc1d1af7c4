"""Geometry of an FFN model (host-side mirror of reference ffn/training/model.py).

Only what the inference path needs: `ModelInfo` (reference model.py:25-46) and
a minimal `FFNModel` base carrying it (reference model.py:64-113).  There is no
graph: the forward pass lives in csrc/ as HIP kernels.
"""

from __future__ import annotations

import dataclasses

import numpy as np


@dataclasses.dataclass
class ModelInfo:
  """Arrays are (x, y, z), as in the reference (model.py:29-31)."""
  deltas: np.ndarray
  pred_mask_size: np.ndarray
  input_seed_size: np.ndarray
  input_image_size: np.ndarray
  additive: bool = False


class FFNModel:
  """Base class: holds `info` and `batch_size` (reference model.py:64-103)."""

  dim = 3

  def __init__(self, info: ModelInfo, batch_size=None, **kwargs):
    del kwargs
    self.info = info
    self.batch_size = batch_size
    for name in ('deltas', 'pred_mask_size', 'input_seed_size',
                 'input_image_size'):
      setattr(self.info, name, np.array(getattr(self.info, name)))
