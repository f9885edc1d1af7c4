#!/usr/bin/env python
"""conv32hs (engine option flow = 3: 64-voxel workgroups, two hand-off chains per
SIMD) against conv32ps (flow = 2) on one box: microseconds per stack, and from the
debug_clock 4 stamps of ONE stack the per-layer period, where a workgroup's layer
goes (wait / stage / taps / epilogue + drain / publish), how the two workgroups of a
CU sit against each other in time, and whether the dispatcher paired them as meant.

  FFN_AMD_LIB=.../libffn_hip_NAME.so python tools/gpu_half_trace.py [--flow-debug N] [--check]
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


def stack_us(eng, reps=400):
  t_end = time.perf_counter() + 2.0
  while time.perf_counter() < t_end:  # clocks up
    eng.forward_resident(1, 20)
    eng.synchronize()
  eng.synchronize()
  t0 = time.perf_counter()
  eng.forward_resident(1, reps)
  eng.synchronize()
  return (time.perf_counter() - t0) / reps * 1e6


def mate_overlap(raw, n, nl=23):
  """How long a workgroup's taps take against how much of them runs while the OTHER
  workgroup of its CU is in its taps too."""
  tr = raw[:, :nl, :6]
  ids = (raw[:, 1, 6] * 100.0).astype(np.int64)
  xcc = (raw[:, 1, 7] * 100.0).astype(np.int64)
  cu = ((ids >> 8) & 0xff) | (xcc << 8)
  by_cu = {}
  for c in range(n):
    if tr[c, 1, 0] > 0:
      by_cu.setdefault(int(cu[c]), []).append(c)
  pairs = [v for v in by_cu.values() if len(v) == 2]
  if not pairs:
    return
  a = tr[[v[0] for v in pairs]]
  b = tr[[v[1] for v in pairs]]
  half = len(pairs)
  durs, ovs = [], []
  for x, y in ((a, b), (b, a)):
    for l in range(1, nl - 1):
      ov = np.zeros(half)
      for lb in range(nl):
        lo = np.maximum(x[:, l, 2], y[:, lb, 2])
        hi = np.minimum(x[:, l, 3], y[:, lb, 3])
        ov += np.clip(hi - lo, 0, None)
      durs.append(x[:, l, 3] - x[:, l, 2])
      ovs.append(ov)
  durs, ovs = np.concatenate(durs), np.concatenate(ovs)
  coef = np.linalg.lstsq(np.vstack([np.ones_like(ovs), ovs]).T, durs, rcond=None)[0]
  print('taps of a workgroup ~ %.2f us + %.2f x (us of them under the CU-mate\'s taps); mean '
        'overlap %.2f us' % (coef[0], coef[1], ovs.mean()))
  for lo_, hi_ in ((0, 0.1), (0.1, 0.8), (0.8, 1.5), (1.5, 2.2), (2.2, 9)):
    m = (ovs >= lo_) & (ovs < hi_)
    if m.sum():
      print('  overlap %.1f - %.1f us: %5d workgroup-layers, taps median %.2f us' % (
          lo_, hi_, m.sum(), np.median(durs[m])))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--flow-debug', type=int, default=0)
  ap.add_argument('--rounds', type=int, default=3)
  ap.add_argument('--check', action='store_true', help='logits of flow 3 against flow 2')
  ap.add_argument('--dump', default='', help='save the raw stamps (.npy)')
  ap.add_argument('--pace', default='', help='flow_pace values (10-ns ticks) to time, comma-separated')
  ap.add_argument('--trace-pace', type=int, default=0, help='flow_pace of the traced stack')
  args = ap.parse_args()
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  eng = hip_engine.HipEngine.from_model(model, max_batch=1)
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
  seed = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
  ref = eng.predict(seed, img)
  lib = os.path.basename(os.environ.get('FFN_AMD_LIB', 'libffn_hip.so'))
  if args.check:
    eng.set_option('conv_variant', 10)
    got = eng.predict(seed, img)
    print('%s: flow 3 against flow 2: max |d logit| %.3g, identical %s, timeouts %d' % (
        lib, np.abs(got - ref).max(), np.array_equal(got, ref),
        eng.get_option('stat_flow_timeouts')))
  eng.set_option('flow_debug', args.flow_debug)
  for r in range(args.rounds):
    eng.set_option('conv_variant', 9)
    a = stack_us(eng)
    eng.set_option('conv_variant', 10)
    b = stack_us(eng)
    print('%s round %d: us per stack (conv0_a + resident launch)  variant 9: %.2f   variant 10: %.2f  '
          '(%+.1f %%)' % (lib, r, a, b, (b / a - 1) * 100), flush=True)
  for pv in [int(x) for x in args.pace.split(',') if x]:
    eng.set_option('flow_pace', pv)
    eng.set_option('flow_pace_spread', pv)
    print('%s variant 10 paced at %d ticks (%.2f us per conv; 24 x = %.1f us): %.2f us per stack' % (
        lib, pv, pv / 100.0, 24 * pv / 100.0, stack_us(eng, 200)), flush=True)
  eng.set_option('flow_pace', args.trace_pace)
  eng.set_option('flow_pace_spread', args.trace_pace)
  print('timeouts', eng.get_option('stat_flow_timeouts'), '; traced stack: flow_pace', args.trace_pace)
  eng.set_option('debug_clock', 4)
  eng.forward_resident(1, 1)
  eng.synchronize()
  n = int(os.environ.get('FFN_H_SLOTS', '450'))
  raw = eng.debug_flow_trace(n).astype(np.float64) / 100.0  # us
  eng.set_option('debug_clock', 0)
  if args.dump:
    np.save(args.dump, raw)
  nl = 23
  tr = raw[:, :nl, :6]
  live = tr[:, 1, 0] > 0
  t00 = tr[live][:, 0, 0].min()
  tr = np.where(tr > 0, tr - t00, np.nan)
  print('variant 10, flow_debug %d: %d workgroups stamped; span %.1f us' % (
      args.flow_debug, live.sum(), np.nanmax(tr[:, nl - 1, 5])))
  per = np.diff(np.nanmedian(tr[:, :, 0], axis=0))
  print('median entry-to-entry period per layer: %s; mean (layers 2..) %.2f us' % (
      ' '.join('%.2f' % v for v in per), per[1:].mean()))
  names = ['wait', 'stage', 'taps', 'epilogue+drain', 'publish']
  half = n // 2
  for name, rows in (('first slots ', slice(0, half)), ('second slots', slice(half, n)),
                     ('first  64 chunks', slice(0, 64)), ('last 64 of the first half', slice(half - 64, half)),
                     ('first 64 of the second half', slice(half, half + 64)), ('last 64 chunks', slice(n - 64, n))):
    x = tr[rows, 2:nl - 1, :]
    d = np.diff(x, axis=2)
    p = np.nanmedian(np.diff(tr[rows, 1:nl - 1, 0], axis=1))
    gap = np.nanmedian(tr[rows, 2:nl - 1, 0] - tr[rows, 1:nl - 2, 5])
    print('%s | period %5.2f | %s | published -> next entry %.2f' % (
        name, p, ' / '.join('%s %.2f' % (nm, np.nanmedian(d[:, :, k]))
                            for k, nm in enumerate(names)), gap))
  # the two workgroups of a CU: HW_ID CU_ID bits 8-11, SH 12, SE 13-15; XCC_ID apart
  ids = (raw[:, 1, 6] * 100.0).astype(np.int64)
  xcc = (raw[:, 1, 7] * 100.0).astype(np.int64)
  cu = ((ids >> 8) & 0xff) | (xcc << 8)
  simd = (ids >> 4) & 3
  by_cu = {}
  for c in range(n):
    if live[c]:
      by_cu.setdefault(int(cu[c]), []).append(c)
  sizes = np.bincount([len(v) for v in by_cu.values()])
  print('CUs used %d; workgroups per CU histogram %s' % (len(by_cu), sizes.tolist()))
  meant = sum(1 for v in by_cu.values() if len(v) == 2 and abs(v[0] - v[1]) >= n // 3)
  print('CUs whose two workgroups are chunks at least %d apart: %d' % (n // 3, meant))
  offs = []
  for v in by_cu.values():
    if len(v) == 2:
      a, b = sorted(v)
      offs.append(np.nanmedian(tr[b, 2:nl - 1, 2] - tr[a, 2:nl - 1, 2]))  # taps start
  offs = np.array(offs)
  print('start of taps, second workgroup of a CU minus first (median over layers): median %.2f, '
        '10 %% %.2f, 90 %% %.2f us' % (np.nanmedian(offs), np.nanpercentile(offs, 10),
                                        np.nanpercentile(offs, 90)))
  mate_overlap(raw, n)
  print('timeouts', eng.get_option('stat_flow_timeouts'))
  eng.close()


if __name__ == '__main__':
  main()
