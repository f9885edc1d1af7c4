#!/usr/bin/env python3
"""Batch-1 time of the conv stack against the number of conv32m chunks (128
voxels each) of the FoV: 256 CUs host two workgroups each, so how much of a
33^3 layer's time (281 chunks) is the 25 CUs that run two workgroups?  (Variant
9, conv32mt, where the FoV has 257 .. 512 chunks: the step it removes.)

  python tools/gpu_chunks_vs_cus.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ffn_amd import engine as hip_engine  # noqa: E402
from ffn_amd.training.models import convstack_3d  # noqa: E402
from oracle import ffn_oracle  # noqa: E402  (random_weights only)


def main():
  depth = 12
  variables = ffn_oracle.random_weights(depth, seed=3, stddev=0.03)
  rows = []
  for fov in ([33, 33, 25], [33, 33, 29], [33, 31, 31], [33, 33, 30], [33, 33, 31],
              [33, 33, 33], [33, 33, 35], [33, 33, 41], [33, 33, 47]):
    try:
      m = convstack_3d.ConvStack3DFFNModel(fov_size=fov, deltas=[8, 8, 8],
                                           depth=depth)
      m.set_variables(variables)
      eng = hip_engine.HipEngine.from_model(m, max_batch=1)
    except Exception as e:  # pylint:disable=broad-except
      print('fov %s: %r' % (fov, e))
      continue
    zyx = fov[::-1]
    rng = np.random.RandomState(0)
    img = rng.normal(0, 1, [1] + zyx).astype(np.float32)
    seed = rng.normal(0, 1, [1] + zyx).astype(np.float32)
    v = int(np.prod(fov))
    chunks = (v + 127) // 128
    for variant in (9, 8, 6):
      try:
        eng.set_option('conv_variant', variant)
      except Exception:  # pylint:disable=broad-except
        continue
      eng.predict(seed, img)
      ts = []
      for _ in range(7):
        eng.forward_resident(1, 3)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.forward_resident(1, 100)
        eng.synchronize()
        ts.append((time.perf_counter() - t0) / 100)
      us = float(np.median(ts)) * 1e6
      print('fov %s: %6d voxels, %3d chunks of 128 (variant %d): %7.1f us / stack, '
            '%5.2f us / layer, %6.1f ns / chunk-layer' %
            (fov, v, chunks, variant, us, us / 24, us / 24 / chunks * 1e3), flush=True)
    eng.close()


if __name__ == '__main__':
  main()
