#!/usr/bin/env python3
"""When and where the workgroups of ONE conv launch run (debug_clock 2): start /
end of every workgroup on the 100 MHz wall clock, the CU it ran on.

  python tools/gpu_wg_timeline.py [--variants 8 9] [--layers 3 4]
"""
import argparse
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ffn_amd import engine as hip_engine  # noqa: E402
from ffn_amd.training.models import convstack_3d  # noqa: E402


def describe(rec, label, mains_per_xcd=None, tails_per_xcd=None):
  ran = rec[:, 1] > 0
  r = rec[ran]
  t0 = r[:, 0].min()
  start = (r[:, 0] - t0) * 10.0   # ns
  end = (r[:, 1] - t0) * 10.0
  hw, xcc = r[:, 2], r[:, 3] & 0xf
  cu = (hw >> 8) & 0xf
  sh = (hw >> 12) & 0x1
  se = (hw >> 13) & 0x7
  where = xcc * 1000 + se * 100 + sh * 10 + cu   # a CU's identity
  per_cu = collections.Counter(where.tolist())
  print('%s: %d workgroups on %d distinct CUs (%d CUs host 2, %d host 3+); launch '
        'spans %.2f us' % (label, len(r), len(per_cu),
                           sum(1 for v in per_cu.values() if v == 2),
                           sum(1 for v in per_cu.values() if v > 2), end.max() / 1e3))
  print('   start: median %.2f  p90 %.2f  max %.2f us;  end: median %.2f  p90 %.2f  '
        'max %.2f us;  duration: median %.2f  p90 %.2f  max %.2f us' % (
            np.median(start) / 1e3, np.percentile(start, 90) / 1e3, start.max() / 1e3,
            np.median(end) / 1e3, np.percentile(end, 90) / 1e3, end.max() / 1e3,
            np.median(end - start) / 1e3, np.percentile(end - start, 90) / 1e3,
            (end - start).max() / 1e3))
  if mains_per_xcd is not None:
    blk = np.flatnonzero(ran)
    idx = blk >> 3
    is_tail = idx >= mains_per_xcd
    shared = np.array([per_cu[w] > 1 for w in where.tolist()])
    for name, sel in (('main alone on its CU', ~is_tail & ~shared),
                      ('main sharing its CU', ~is_tail & shared),
                      ('tail', is_tail)):
      if sel.any():
        print('   %-22s n=%3d  start median %.2f max %.2f | duration median %.2f '
              'max %.2f | end median %.2f max %.2f us' % (
                  name, sel.sum(), np.median(start[sel]) / 1e3, start[sel].max() / 1e3,
                  np.median((end - start)[sel]) / 1e3, (end - start)[sel].max() / 1e3,
                  np.median(end[sel]) / 1e3, end[sel].max() / 1e3))
    tails_with = collections.Counter()
    for w, t in zip(where.tolist(), is_tail.tolist()):
      tails_with[w] += int(t)
    print('   CUs by number of tail workgroups: %s' %
          dict(collections.Counter(tails_with.values())))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--variants', type=int, nargs='+', default=[8, 9])
  ap.add_argument('--layers', type=int, nargs='+', default=[3, 4])
  ap.add_argument('--batch', type=int, default=1,
                  help='FoVs per launch (<= 14: 4,096 workgroup stamps)')
  args = ap.parse_args()
  B = args.batch
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33],
                                           deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  eng = hip_engine.HipEngine.from_model(model, max_batch=B)
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, (B, 33, 33, 33)).astype(np.float32)
  seed = rng.normal(0, 1, (B, 33, 33, 33)).astype(np.float32)
  eng.predict(seed, img)
  if B > 1:
    eng.set_option('batch_chunks', 0)
  eng.set_option('debug_clock', 2)
  for v in args.variants:
    eng.set_option('conv_variant', v)
    for layer in args.layers:
      eng.set_option('debug_layer', layer)
      for rep in range(2):
        eng.forward_resident(B, 3)
        rec = eng.debug_workgroups(512 if B == 1 else 4096)
        if rep == 1:
          describe(rec, 'variant %d layer %d batch %d' % (v, layer, B),
                   32 if v == 9 and B == 1 else None, 13 if v == 9 and B == 1 else None)
          if B > 1:
            # the in-kernel stamps of workgroup 0 (it shares its CU from the start)
            eng.set_option('debug_clock', 1)
            eng.forward_resident(B, 2)
            c = eng.debug_clocks()
            eng.set_option('debug_clock', 2)
            for w in range(4):
              print('   workgroup 0 wave %d: prologue %d  loop %d  epilogue %d  total %d '
                    'cycles, wall %.2f us' % (w, c[w, 1] - c[w, 0], c[w, 2] - c[w, 1],
                                              c[w, 3] - c[w, 2], c[w, 3] - c[w, 0],
                                              (c[w, 5] - c[w, 4]) / 100.0))
          ran = rec[:, 1] > 0
          end = (rec[ran, 1] - rec[ran, 0].min()) * 10.0 / 1e3
          hist, edges = np.histogram(end, bins=12)
          print('   end times (us): ' + '  '.join(
              '%.1f-%.1f:%d' % (edges[k], edges[k + 1], hist[k])
              for k in range(len(hist)) if hist[k]))
  if 9 in args.variants and B == 1:
    # the shader-clock stamps of ONE tail workgroup (debug_clock 3): entry ->
    # first barrier (W0 + dz = -1 landed) -> last tap -> exit
    eng.set_option('debug_clock', 3)
    eng.set_option('conv_variant', 9)
    for layer in args.layers:
      eng.set_option('debug_layer', layer)
      eng.forward_resident(1, 3)
      c = eng.debug_clocks()
      for w in range(4):
        print('variant 9 layer %d tail chunk 0 wave %d: stage %d  taps %d  epilogue %d  '
              'total %d shader cycles; wall %.0f ns' % (
                  layer, w, c[w, 1] - c[w, 0], c[w, 2] - c[w, 1], c[w, 3] - c[w, 2],
                  c[w, 3] - c[w, 0], (c[w, 5] - c[w, 4]) * 10.0))
  eng.close()


if __name__ == '__main__':
  main()
