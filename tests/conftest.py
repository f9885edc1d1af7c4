import functools
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


@functools.lru_cache(maxsize=None)
def has_gpu():
  """True if the HIP runtime sees a device (asked of libamdhip64 itself, the
  library the product links, not of torch)."""
  import ctypes
  for name in ('libamdhip64.so', '/opt/rocm/lib/libamdhip64.so'):
    try:
      hip = ctypes.CDLL(name)
    except OSError:
      continue
    n = ctypes.c_int(0)
    try:
      return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:  # pylint:disable=broad-except
      return False
  return False


def pytest_collection_modifyitems(config, items):
  """`gpu` tests are skipped (not failed) on a box without a ROCm device."""
  del config
  if has_gpu():
    return
  skip = pytest.mark.skip(reason='no ROCm device visible')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


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
