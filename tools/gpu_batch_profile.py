#!/usr/bin/env python3
"""Kernel-only run of the conv stack at one batch size and conv_variant, for
rocprofv3 (kernel stats / --pmc passes of the BATCHED configuration):

  rocprofv3 --kernel-trace --stats -- python tools/gpu_batch_profile.py --batch 8 --variant 8
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ffn_amd import engine as hip_engine  # noqa: E402
from ffn_amd.training.models import convstack_3d  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=8)
  ap.add_argument('--variant', type=int, default=8)
  ap.add_argument('--repeats', type=int, default=40)
  args = ap.parse_args()
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33],
                                           deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  eng = hip_engine.HipEngine.from_model(model, max_batch=args.batch)
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, (args.batch, 33, 33, 33)).astype(np.float32)
  seed = rng.normal(0, 1, (args.batch, 33, 33, 33)).astype(np.float32)
  eng.set_option('batch_chunks', 0)
  eng.set_option('conv_variant', args.variant)
  eng.predict(seed, img)
  eng.forward_resident(args.batch, 3)
  eng.synchronize()
  t0 = time.perf_counter()
  eng.forward_resident(args.batch, args.repeats)
  eng.synchronize()
  dt = (time.perf_counter() - t0) / args.repeats
  flop = 2.0 * 27 * 32 * 32 * 33**3 * 23 * args.batch
  print('variant %d batch %d: %.1f us per stack, %.2f us per FoV-layer, %.1f TFLOP/s '
        'algorithmic (x3 executed on the fp16 pipe: %.3f of 2,500)' %
        (args.variant, args.batch, dt * 1e6, dt * 1e6 / 25 / args.batch * (25 / 25.0),
         flop / dt / 1e12, 3 * flop / dt / 2.5e15))
  eng.close()


if __name__ == '__main__':
  main()
