import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run via gpurun)')


def has_gpu():
  try:
    import torch
    return torch.cuda.is_available()
  except Exception:  # pylint:disable=broad-except
    return False


@pytest.fixture(scope='session')
def fib25_variables():
  with np.load(os.path.join(GOLDEN, 'fib25_weights.npz')) as d:
    return {k: d[k] for k in d.files}


@pytest.fixture(scope='session')
def fib25_blob(fib25_variables):
  from oracle import ffn_oracle
  return ffn_oracle.weights_blob(fib25_variables, 12)


@pytest.fixture(scope='session')
def fib25_model(fib25_variables):
  from ffn_amd.training.models import convstack_3d
  m = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8],
                                       batch_size=1, depth=12)
  m.set_variables(fib25_variables)
  return m
