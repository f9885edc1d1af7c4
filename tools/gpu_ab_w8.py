#!/usr/bin/env python3
"""Same-process A/B of conv32w8 modes (option "waves8"): 1 = segment conversion
between barriers, 2 = conversion interleaved with the taps.  Prints
bit-equality of the logits, us/stack at batch 1 and 8, and in-kernel clocks.
(Mode 3 of profiles/r01_ab_w8_weights_via_lds_hybrid.txt -- weights of dz
segments 1, 2 through LDS -- was measured with this script and not kept.)

  python tools/gpu_ab_w8.py [mode_a mode_b]      (default 1 2)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ffn_amd import engine as hip_engine  # noqa: E402
from ffn_amd.training.models import convstack_3d  # noqa: E402


def main():
  modes = tuple(int(v) for v in sys.argv[1:3]) or (1, 2)
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33],
                                           deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  eng = hip_engine.HipEngine.from_model(model, max_batch=8)
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, (8, 33, 33, 33)).astype(np.float32)
  seed = rng.normal(0, 1, (8, 33, 33, 33)).astype(np.float32)
  outs = {}
  for w in modes:
    eng.set_option('waves8', w)
    outs[w] = np.array(eng.predict(seed, img))
  print('logits bit-equal (waves8 %d vs %d):' % modes,
        np.array_equal(outs[modes[0]], outs[modes[1]]),
        'max abs diff %g' % np.max(np.abs(outs[modes[0]] - outs[modes[1]])))
  repeats = 200
  for rnd in range(3):
    for w in modes:
      eng.set_option('waves8', w)
      line = 'round %d waves8 %d:' % (rnd, w)
      for b in (1, 8):
        eng.forward_resident(b, 5)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.forward_resident(b, repeats if b == 1 else repeats // 4)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / (repeats if b == 1 else repeats // 4)
        line += '  batch %d %7.1f us/stack' % (b, dt * 1e6)
      print(line)
  eng.set_option('debug_clock', 1)
  for w in modes:
    eng.set_option('waves8', w)
    eng.forward_resident(1, 3)
    c = eng.debug_clocks()
    for k in range(2):
      print('waves8 %d clock wave %d: stage %d  loop %d  epilogue %d  total %d' % (
          w, k, c[k, 1] - c[k, 0], c[k, 2] - c[k, 1], c[k, 3] - c[k, 2],
          c[k, 3] - c[k, 0]))
  eng.set_option('debug_clock', 0)


if __name__ == '__main__':
  main()
