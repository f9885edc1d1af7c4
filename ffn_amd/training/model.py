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

  def update_seed(self, seed, update):
    """Updates the initial `seed` with `update` (reference model.py:168-183).

    Arrays are [batch, z, y, x, 1] (or [z, y, x]); when the model predicts a
    smaller mask than the seed it reads, the update is zero-padded around the
    centre exactly as the reference's tf.pad does (dz // 2 in front, the rest
    behind, per axis).  Returns the updated seed (in place for ndarrays).
    """
    dx = int(self.info.input_seed_size[0] - self.info.pred_mask_size[0])
    dy = int(self.info.input_seed_size[1] - self.info.pred_mask_size[1])
    dz = int(self.info.input_seed_size[2] - self.info.pred_mask_size[2])
    seed = np.asarray(seed)
    update = np.asarray(update)
    if dx == 0 and dy == 0 and dz == 0:
      seed += update
      return seed
    pad3 = [(dz // 2, dz - dz // 2), (dy // 2, dy - dy // 2),
            (dx // 2, dx - dx // 2)]
    if update.ndim == 5:
      pad = [(0, 0)] + pad3 + [(0, 0)]
    elif update.ndim == 4:
      pad = [(0, 0)] + pad3
    else:
      pad = pad3
    seed += np.pad(update, pad)
    return seed
