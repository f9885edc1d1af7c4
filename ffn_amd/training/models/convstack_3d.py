"""ConvStack3DFFNModel: host-side description of the FFN conv stack.

Mirror of reference ffn/training/models/convstack_3d.py:59-102 for inference:
same constructor arguments (xyz order), same `.info`.  The network itself
(depth-N residual stack of 3x3x3 SAME convs, 32 features, 1x1x1 logit head,
logits = seed + update; reference convstack_3d.py:26-56,83-95) executes as the
HIP kernels in ffn_amd/csrc; this class only carries geometry and weights.
"""

from __future__ import annotations

import numpy as np

from .. import model
from .. import tf_checkpoint


def conv_scopes(depth: int):
  """TF variable scopes in graph order (reference convstack_3d.py:38-54)."""
  names = ['conv0_a', 'conv0_b']
  for i in range(1, depth):
    names += ['conv%d_a' % i, 'conv%d_b' % i]
  names.append('conv_lom')
  return names


class ConvStack3DFFNModel(model.FFNModel):
  """Geometry + weights of the conv-stack FFN."""

  dim = 3

  def __init__(self, fov_size=None, deltas=None, batch_size=None,
               depth: int = 9, features: int = 32, **kwargs):
    info = model.ModelInfo(deltas, fov_size, fov_size, fov_size)
    super().__init__(info, batch_size, **kwargs)
    self.depth = int(depth)
    self.features = int(features)
    self.variables = None

  # -- weights ---------------------------------------------------------------
  def load_checkpoint(self, checkpoint_path: str):
    """Reads a TF checkpoint prefix (or an .npz keyed by TF variable names)."""
    if checkpoint_path.endswith('.npz'):
      with np.load(checkpoint_path) as data:
        variables = {k: data[k] for k in data.files}
    else:
      variables = tf_checkpoint.load_checkpoint(checkpoint_path)
    self.set_variables(variables)

  def set_variables(self, variables):
    f = self.features
    for name in conv_scopes(self.depth):
      w = variables['seed_update/%s/weights' % name]
      b = variables['seed_update/%s/biases' % name]
      if name == 'conv0_a':
        want = (3, 3, 3, 2, f)
      elif name == 'conv_lom':
        want = (1, 1, 1, f, 1)
      else:
        want = (3, 3, 3, f, f)
      if tuple(w.shape) != want or tuple(b.shape) != (want[-1],):
        raise ValueError('%s: weights %r biases %r, expected %r' %
                         (name, w.shape, b.shape, want))
    self.variables = variables

  def init_random(self, seed: int = 0, stddev: float = 0.01):
    """TruncatedNormal(stddev=0.01)-like init (reference convstack_3d.py:24-25)."""
    rng = np.random.RandomState(seed)
    f = self.features
    variables = {}
    for name in conv_scopes(self.depth):
      if name == 'conv0_a':
        shape = (3, 3, 3, 2, f)
      elif name == 'conv_lom':
        shape = (1, 1, 1, f, 1)
      else:
        shape = (3, 3, 3, f, f)
      w = np.clip(rng.normal(0, stddev, shape), -2 * stddev, 2 * stddev)
      variables['seed_update/%s/weights' % name] = w.astype(np.float32)
      variables['seed_update/%s/biases' % name] = np.zeros(shape[-1:],
                                                           np.float32)
    self.variables = variables

  def weights_blob(self) -> np.ndarray:
    """Flat f32 blob in the order ffn_engine_set_weights expects."""
    if self.variables is None:
      raise ValueError('model has no weights (load_checkpoint / init_random)')
    parts = []
    for name in conv_scopes(self.depth):
      parts.append(np.ascontiguousarray(
          self.variables['seed_update/%s/weights' % name],
          dtype=np.float32).ravel())
      parts.append(np.ascontiguousarray(
          self.variables['seed_update/%s/biases' % name],
          dtype=np.float32).ravel())
    return np.concatenate(parts)
