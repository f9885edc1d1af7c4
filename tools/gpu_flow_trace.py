#!/usr/bin/env python
"""Timeline of the resident conv stack (engine option flow = 2) from its
debug_clock 4 stamps: per conv of the stack and class of workgroup (main /
tail), where a layer's time goes -- waiting for the input tiles, staging the
first segment, the taps, the epilogue + drain of the write-through stores, the
publish -- and the period at which the layers follow each other.

  python tools/gpu_flow_trace.py [--flow-debug N]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffn_amd import engine as hip_engine  # noqa: E402
from ffn_amd.training.models import convstack_3d  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--flow', type=int, default=2)
  ap.add_argument('--flow-debug', type=int, default=0)
  ap.add_argument('--option', action='append', default=[], help='name=value')
  args = ap.parse_args()
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8],
                                           depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  eng = hip_engine.HipEngine.from_model(model, max_batch=1)
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
  seed = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
  eng.predict(seed, img)
  eng.set_option('flow', args.flow)
  eng.set_option('flow_debug', args.flow_debug)
  for item in args.option:
    name, _, value = item.partition('=')
    eng.set_option(name, int(value))
  import time
  t_end = time.perf_counter() + 3.0
  while time.perf_counter() < t_end:  # clocks up
    eng.forward_resident(1, 20)
    eng.synchronize()
  eng.synchronize()
  t0 = time.perf_counter()
  eng.forward_resident(1, 400)
  eng.synchronize()
  print('library %s: %.2f us per stack (conv0_a + the resident launch, 400 back to back)'
        % (os.path.basename(os.environ.get('FFN_AMD_LIB', 'libffn_hip.so')),
           (time.perf_counter() - t0) / 400 * 1e6))
  eng.set_option('debug_clock', 4)
  eng.forward_resident(1, 1)
  eng.synchronize()
  nslots = 256 + 100
  tr = eng.debug_flow_trace(nslots).astype(np.float64) / 100.0  # us
  eng.set_option('debug_clock', 0)
  nl = 23
  hw = eng_hw = None
  raw = tr
  tr = tr[:, :nl, :6]
  t00 = tr[:, 0, 0].min()
  tr = np.where(tr > 0, tr - t00, np.nan)
  main, tail = tr[:256], tr[256:]
  names = ['wait for tiles', 'stage dz=-1', 'taps', 'epilogue+drain', 'barrier+publish']
  print('flow %d flow_debug %d; all times in us; stack span (first entry -> last end) %.1f'
        % (args.flow, args.flow_debug, np.nanmax(tr[:, nl - 1, 5])))
  print('layer | entry: first / median / last | end(published): first / median / last |'
        ' main: ' + ' / '.join(names) + ' (median) | tail: same')
  for l in range(nl):
    e0 = tr[:, l, 0]
    e5 = tr[:, l, 5]
    def phases(x):
      d = np.diff(x[:, l, :], axis=1)
      if l == 0:
        d[:, 0] = 0
        d[:, 1] = x[:, l, 2] - x[:, l, 0]
      return ' / '.join('%5.2f' % v for v in np.nanmedian(d, axis=0))
    print('%2d | %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f | %s | %s' %
          (l, np.nanmin(e0), np.nanmedian(e0), np.nanmax(e0), np.nanmin(e5),
           np.nanmedian(e5), np.nanmax(e5), phases(main), phases(tail)))
  # by region of the FoV: the bottom / middle / top main chunks and the tail tiles
  # (the FoV's last voxels): who runs at which period, and where its time goes
  print('region           | period | wait / stage / taps / epilogue+drain (median over layers 2..%d)'
        % (nl - 2))
  for name, rows in (('main   0 -  31', slice(0, 32)), ('main  96 - 127', slice(96, 128)),
                     ('main 192 - 223', slice(192, 224)), ('main 224 - 255', slice(224, 256)),
                     ('main 240 - 255', slice(240, 256)), ('tail   0 -  49', slice(256, 306)),
                     ('tail  50 -  99', slice(306, 356))):
    x = tr[rows, 2:nl - 1, :]
    per_r = np.nanmedian(np.diff(tr[rows, 1:nl - 1, 0], axis=1))
    d = np.diff(x, axis=2)
    print('%s |  %5.2f | %s' % (name, per_r, ' / '.join(
        '%5.2f' % np.nanmedian(d[:, :, k]) for k in range(4))))
  gap = tr[:256, 2:nl - 1, 0] - tr[:256, 1:nl - 2, 5]  # published (conv l) -> entry (conv l + 1)
  body = tr[:256, 1:nl - 2, 5] - tr[:256, 1:nl - 2, 0]
  print('main workgroups, per workgroup and layer: entry -> published median %.2f; published -> '
        'next entry median %.2f (10 %% %.2f, 90 %% %.2f); their sum = the period'
        % (np.nanmedian(body), np.nanmedian(gap), np.nanpercentile(gap, 10),
           np.nanpercentile(gap, 90)))
  per = np.diff(np.nanmedian(tr[:256, :, 0], axis=0))
  print('median entry-to-entry period of the main workgroups, per layer: %s; mean %.2f us'
        % (' '.join('%.2f' % v for v in per), per[1:].mean()))
  # per main workgroup: its slowest-layer body and when its tiles were seen
  body = tr[:256, 1:nl - 1, 5] - tr[:256, 1:nl - 1, 1]  # tiles seen -> published
  print('main workgroups, tiles seen -> published: median %.2f, 10 %% %.2f, 90 %% %.2f, max %.2f'
        % (np.nanmedian(body), np.nanpercentile(body, 10), np.nanpercentile(body, 90),
           np.nanmax(body)))
  bodyt = tr[256:, 1:nl - 1, 5] - tr[256:, 1:nl - 1, 1]
  print('tail workgroups, tiles seen -> published: median %.2f, 90 %% %.2f, max %.2f'
        % (np.nanmedian(bodyt), np.nanpercentile(bodyt, 90), np.nanmax(bodyt)))
  # hop: for main chunk c at layer l, when did the last of its input tiles get
  # published (chunks c - 10 .. c + 10 of layer l - 1, clipped; + all tails for
  # the last chunks) vs when it saw them
  hop = []
  for l in range(1, nl):
    pub = tr[:, l - 1, 5]
    for c in range(256):
      lo, hi = max(0, c * 128 - 1123) // 128, min(35936, c * 128 + 127 + 1123) // 128
      srcs = list(range(lo, min(hi, 255) + 1))
      if hi >= 256:
        t_lo = max(0, (c * 128 - 1123 - 32768) // 32)
        t_hi = min(99, (c * 128 + 127 + 1123 - 32768) // 32)
        srcs += [256 + t for t in range(t_lo, t_hi + 1)]
      hop.append(tr[c, l, 1] - np.nanmax(pub[srcs]))
  hop = np.array(hop)
  print('last input tile published -> tiles seen by the consumer (main): median %.2f, 10 %% '
        '%.2f, 90 %% %.2f' % (np.nanmedian(hop), np.nanpercentile(hop, 10),
                               np.nanpercentile(hop, 90)))
  # which main workgroups share their CU with a tail workgroup?  HW_ID: CU_ID bits
  # 8-11, SH_ID 12, SE_ID 13-15 (gfx9); XCC_ID separately
  ids = (raw[:, 1, 6] * 100.0).astype(np.int64)  # (stamps were divided by 100)
  xcc = (raw[:, 1, 7] * 100.0).astype(np.int64)
  cu = ((ids >> 8) & 0xff) | (xcc << 8)
  tail_cus = set(cu[256:].tolist())
  shared = np.array([c in tail_cus for c in cu[:256]])
  b_all = tr[:256, 1:nl - 1, 5] - tr[:256, 1:nl - 1, 1]
  taps = tr[:256, 1:nl - 1, 3] - tr[:256, 1:nl - 1, 2]
  print('main workgroups that share their CU with a tail workgroup: %d of 256; body '
        '(tiles seen -> published) median %.2f against %.2f alone; taps %.2f against %.2f'
        % (shared.sum(), np.nanmedian(b_all[shared]), np.nanmedian(b_all[~shared]),
           np.nanmedian(taps[shared]), np.nanmedian(taps[~shared])))
  print('timeouts', eng.get_option('stat_flow_timeouts'))
  eng.close()


if __name__ == '__main__':
  main()
