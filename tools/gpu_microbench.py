#!/usr/bin/env python3
"""Kernel-only timing of the conv stack (no host traffic, no Python per step).

  python tools/gpu_microbench.py [--batch 1 4 32] [--repeats 50]
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
  ap.add_argument('--batch', type=int, nargs='+', default=[1, 2, 4, 8, 32])
  ap.add_argument('--repeats', type=int, default=50)
  args = ap.parse_args()
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33],
                                           deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  maxb = max(args.batch)
  eng = hip_engine.HipEngine.from_model(model, max_batch=maxb)
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, (maxb, 33, 33, 33)).astype(np.float32)
  seed = rng.normal(0, 1, (maxb, 33, 33, 33)).astype(np.float32)
  eng.predict(seed, img)  # fills the staging buffers
  flop = 2.0 * (2 * 27 * 32 + 23 * 27 * 32 * 32 + 32) * 33**3
  for variant, w8 in ((1, 0), (2, 0), (3, 0), (3, 1), (4, 0), (4, 1)):
    eng.set_option('conv_variant', variant)
    eng.set_option('waves8', w8)
    for b in args.batch:
      eng.forward_resident(b, 3)
      eng.synchronize()
      t0 = time.perf_counter()
      eng.forward_resident(b, args.repeats)
      eng.synchronize()
      dt = (time.perf_counter() - t0) / args.repeats
      eng.set_profiling(1)
      eng.get_profile(reset=True)
      eng.forward_resident(b, 10)
      ms, n = eng.get_profile(reset=True)
      eng.set_profiling(0)
      print('variant %d%s batch %2d: %8.1f us/stack  %8.1f FoV/s  %6.2f TFLOP/s '
            ' conv32 avg %.2f us (%d launches, %.1f TF/s in-kernel)' %
            (variant, '+w8' if w8 else '', b, dt * 1e6, b / dt, b * flop / dt / 1e12,
             ms / n * 1e3, n, b * 2.0 * 27 * 32 * 32 * 33**3 / (ms / n * 1e-3) / 1e12))
  # stateless boundary (ffn_predict): host seed + image in, host logits out
  eng.set_option('conv_variant', 4)
  eng.set_option('waves8', 1)
  for b in (1, maxb):
    eng.predict(seed[:b], img[:b])
    t0 = time.perf_counter()
    for _ in range(args.repeats):
      eng.predict(seed[:b], img[:b])
    dt = (time.perf_counter() - t0) / args.repeats
    print('variant 4+w8 ffn_predict (PCIe-inclusive, %d x 3 x 144 KB) batch %2d: '
          '%8.1f us/call  %8.1f FoV/s' % (b, b, dt * 1e6, b / dt))
  for policy in (1, 2, 0):
    eng.set_option('store_policy', policy)
    for b in (1, 8):
      eng.forward_resident(b, 3)
      eng.synchronize()
      t0 = time.perf_counter()
      eng.forward_resident(b, args.repeats)
      eng.synchronize()
      dt = (time.perf_counter() - t0) / args.repeats
      print('variant 4+w8 store_policy %d batch %d: %8.1f us/stack' % (
          policy, b, dt * 1e6))
  # in-kernel clocks of the compact kernels' first workgroup
  eng.set_option('debug_clock', 1)
  for variant, abls, nmfma in ((4, (0,), 13.5 * 27.0), (3, (0,), 27 * 27.0),
                               (2, (0, 8, 16, 24), 972.0)):
    eng.set_option('conv_variant', variant)
    for abl in abls:
      eng.set_option('ablate', abl)
      eng.forward_resident(1, 3)
      c = eng.debug_clocks()
      print('variant %d issue experiment %d (8 = no A reads, 16 = no B loads):'
            % (variant, abl))
      for w in range(4):
        tot, wall = c[w, 3] - c[w, 0], (c[w, 5] - c[w, 4]) * 10.0
        print(' clock wave %d: stage %d  loop %d  epilogue %d  total %d shader '
              'cycles; wall %.0f ns -> %.2f GHz; %.2f cyc/MFMA' % (
                  w, c[w, 1] - c[w, 0], c[w, 2] - c[w, 1], c[w, 3] - c[w, 2],
                  tot, wall, tot / max(wall, 1), (c[w, 2] - c[w, 1]) / nmfma))
  eng.set_option('ablate', 0)
  eng.set_option('debug_clock', 0)
  # phase ablation of the pipelined conv_b kernel (11 of the 23 convs per stack)
  eng.set_option('conv_variant', 1)
  print('ablation (conv_b launches only; 1=no staging loads 2=no MFMA loop '
        '4=no epilogue traffic):')
  for b in (1, 8):
    base = None
    for mask in (0, 1, 2, 4, 5, 6, 7):
      eng.set_option('ablate', mask)
      eng.forward_resident(b, 3)
      eng.synchronize()
      t0 = time.perf_counter()
      eng.forward_resident(b, args.repeats)
      eng.synchronize()
      dt = (time.perf_counter() - t0) / args.repeats
      if mask == 0:
        base = dt
      print('  batch %d ablate %d: %8.1f us/stack  -> conv_b delta %+.2f us/launch'
            % (b, mask, dt * 1e6, (dt - base) * 1e6 / 11))
    eng.set_option('ablate', 0)
  eng.close()


if __name__ == '__main__':
  main()
