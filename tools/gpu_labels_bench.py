#!/usr/bin/env python3
"""Kernel-only rates of the label / seed kernels (HIP events inside the library).

  python tools/gpu_labels_bench.py [--size 250 400]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ffn_amd import labels, seeding, synthetic  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--size', type=int, nargs='+', default=[250, 400])
  args = ap.parse_args()
  ops = labels.default_ops(0)
  seeder = seeding.default_seeder(0)
  for n in args.size:
    shape = (n, n, n)
    vol = synthetic.cells_volume(shape, seed=1234)
    fg = (vol > 110).astype(np.uint64)
    for conn in (1, 3):
      cc = ops.connected_components(fg, conn)
      ms, nbytes = ops.last_timing()
      print('%d^3 u64 connected_components conn=%d: %d comps, %.2f ms, %.0f GB/s '
            'algorithmic' % (n, conn, int(cc.max()), ms, nbytes / ms / 1e6))
    cc32 = cc.astype(np.uint32)
    ops.connected_components(cc32, 1)
    ms, nbytes = ops.last_timing()
    print('%d^3 u32 connected_components: %.2f ms, %.0f GB/s' %
          (n, ms, nbytes / ms / 1e6))
    b = np.roll(cc, 7, axis=2)
    pa, pb, cnt, slots = ops.pair_counts(cc, b)
    ms, nbytes = ops.last_timing()
    print('%d^3 u64 pair_counts: %d pairs, %.3f ms, %.0f GB/s' %
          (n, pa.size, ms, nbytes / ms / 1e6))
    ops.apply_pair_labels(slots, pa)
    ms, nbytes = ops.last_timing()
    print('%d^3 u64 apply_pair_labels: %.3f ms, %.0f GB/s' %
          (n, ms, nbytes / ms / 1e6))
    keys = np.arange(1, int(cc.max()) + 1, dtype=np.uint64)
    ops.remap(cc, keys, keys[::-1].copy())
    ms, nbytes = ops.last_timing()
    print('%d^3 u64 remap: %.3f ms, %.0f GB/s' % (n, ms, nbytes / ms / 1e6))
    image = synthetic.normalize(vol)
    seeds = seeder.peaks(image)
    ms, vox = seeder.last_timing()
    print('%d^3 PolicyPeaks: %d seeds, %.2f ms, %.0f Mvox/s' %
          (n, len(seeds), ms, vox / ms / 1e3))


if __name__ == '__main__':
  main()
