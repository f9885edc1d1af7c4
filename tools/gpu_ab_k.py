#!/usr/bin/env python3
"""Same-process A/B of conv_variant 4 (conv32w8) and 5 (conv32k): logits vs the
exact-f32 variant 2, microseconds per stack at several batch sizes (interleaved
rounds), in-kernel clock stamps of one mid-stack launch.

  python tools/gpu_ab_k.py [--batch 1 8 32] [--rounds 5] [--repeats 40]
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
  ap.add_argument('--batch', type=int, nargs='+', default=[1, 8, 32])
  ap.add_argument('--rounds', type=int, default=5)
  ap.add_argument('--repeats', type=int, default=40)
  ap.add_argument('--variants', type=int, nargs='+', default=[4, 5])
  ap.add_argument('--ablate', action='store_true')
  args = ap.parse_args()
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33],
                                           deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  maxb = max(args.batch)
  eng = hip_engine.HipEngine.from_model(model, max_batch=maxb)
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, (maxb, 33, 33, 33)).astype(np.float32)
  seed = rng.normal(0, 1, (maxb, 33, 33, 33)).astype(np.float32)
  eng.set_option('batch_chunks', 0)  # variant 6 / 7 are compared in their pure forms
  eng.set_option('conv_variant', 2)
  ref = eng.predict(seed[:2], img[:2])
  for v in args.variants:
    eng.set_option('conv_variant', v)
    out = eng.predict(seed[:2], img[:2])
    out2 = eng.predict(seed[:2], img[:2])
    print('variant %d: max |logit - exact f32 kernel| %.3g, deterministic %s' %
          (v, np.abs(out - ref).max(), np.array_equal(out, out2)))
  eng.predict(seed, img)  # fills the staging buffers of every slot
  flop = 2.0 * 27 * 32 * 32 * 33**3 * 23
  for b in args.batch:
    times = {v: [] for v in args.variants}
    for _ in range(args.rounds):
      for v in args.variants:
        eng.set_option('conv_variant', v)
        eng.forward_resident(b, 3)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.forward_resident(b, args.repeats)
        eng.synchronize()
        times[v].append((time.perf_counter() - t0) / args.repeats)
    for v in args.variants:
      t = np.array(times[v]) * 1e6
      med = float(np.median(t))
      print('batch %2d variant %d: median %7.1f us/stack (min %7.1f)  %6.2f us per '
            'FoV-layer  %6.1f TF/s algorithmic (conv32 only)' %
            (b, v, med, t.min(), med / 23 / b * (23.0 / 25.0),
             b * flop / (med * 1e-6) / 1e12))
  eng.set_option('debug_clock', 1)
  for v, nmfma in ((4, 13.5 * 27.0), (5, 210.0), (6, 210.0), (7, 126.0), (8, 162.0),
                   (9, 162.0)):
    if v not in args.variants:
      continue
    eng.set_option('conv_variant', v)
    for layer in ((3, 4, 22) if v >= 6 else (3,)):  # conv_a, conv_b, fused head
      eng.set_option('debug_layer', layer)
      eng.forward_resident(1, 3)
      c = eng.debug_clocks()
      for w in range(4):
        tot, wall = c[w, 3] - c[w, 0], (c[w, 5] - c[w, 4]) * 10.0
        print('variant %d layer %d wave %d: stage %d  loop %d  epilogue %d  total %d '
              'shader cycles; wall %.0f ns -> %.2f GHz; %.2f cyc/MFMA' % (
                  v, layer, w, c[w, 1] - c[w, 0], c[w, 2] - c[w, 1],
                  c[w, 3] - c[w, 2], tot, wall, tot / max(wall, 1),
                  (c[w, 2] - c[w, 1]) / nmfma))
    eng.set_option('debug_layer', 3)
  if 5 in args.variants and args.ablate:
    # issue experiments on conv32k (results are wrong, timing only)
    eng.set_option('debug_clock', 1)
    eng.set_option('conv_variant', 5)
    eng.set_option('fuse_head', 0)
    for abl in (0, 1, 2, 3, 4, 7):
      eng.set_option('ablate', abl)
      eng.forward_resident(1, 3)
      eng.synchronize()
      t0 = time.perf_counter()
      eng.forward_resident(1, args.repeats)
      eng.synchronize()
      dt = (time.perf_counter() - t0) / args.repeats
      eng.forward_resident(1, 2)
      c = eng.debug_clocks()
      w = 0
      print('variant 5 ablate %d (1 no X reads, 2 no conversion, 4 no W loads): '
            '%7.1f us/stack; wave 0 stage %d loop %d epilogue %d' %
            (abl, dt * 1e6, c[w, 1] - c[w, 0], c[w, 2] - c[w, 1],
             c[w, 3] - c[w, 2]))
    eng.set_option('ablate', 0)
    eng.set_option('fuse_head', 1)
  if 5 in args.variants:
    # conv32k phase stamps: entry -> first barrier -> dz = 0 barrier -> dz = +1
    # barrier (-> loop end / exit are in the table above)
    eng.set_option('debug_clock', 2)
    eng.set_option('conv_variant', 5)
    eng.forward_resident(1, 3)
    c = eng.debug_clocks()
    for w in range(4):
      print('variant 5 wave %d: entry -> B0 %d  -> B1 %d  -> B2 %d shader cycles' %
            (w, c[w, 1] - c[w, 0], c[w, 2] - c[w, 1], c[w, 3] - c[w, 2]))
  eng.set_option('debug_clock', 0)
  eng.close()


if __name__ == '__main__':
  main()
