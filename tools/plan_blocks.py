#!/usr/bin/env python3
"""Work-sheet of round 1 (3-D output-block splits, then DESIGN.md §3.1): what 3-D output blocks would stage per
workgroup, against today's dense 144-voxel runs, for a given FoV.

For every split of the FoV into nz x ny x nx near-equal boxes it prints the
number of workgroups, the largest box (the critical path: tiles of 16 voxels),
the mean MFMA-tile occupancy, and the activation rows (128 B each) a workgroup
stages -- (dz + 2)(dy + 2)(dx + 2) for a box, three 214-row segments today.

  python tools/plan_blocks.py [fz fy fx]
"""
import itertools
import math
import sys


def parts(n, k):
  base, extra = divmod(n, k)
  return [base + 1] * extra + [base] * (k - extra)


def main():
  f = [int(v) for v in sys.argv[1:4]] or [33, 33, 33]
  vox = f[0] * f[1] * f[2]
  ideal_tiles = vox / 16.0
  rows_now = 3 * (144 + 2 * (f[2] + 2) + 2 * math.ceil(144 / f[2]))
  print('FoV %r: %d voxels = %.1f tiles; today: %d workgroups x 9 tiles, ~%d '
        'staged rows each' % (f, vox, ideal_tiles, math.ceil(vox / 144), rows_now))
  out = []
  for nz, ny, nx in itertools.product(range(1, 17), repeat=3):
    wgs = nz * ny * nx
    if not 180 <= wgs <= 256:
      continue
    pz, py, px = parts(f[0], nz), parts(f[1], ny), parts(f[2], nx)
    big = pz[0] * py[0] * px[0]
    tiles_max = math.ceil(big / 16)
    tiles_sum = sum(math.ceil(a * b * c / 16) for a in pz for b in py for c in px)
    rows = (pz[0] + 2) * (py[0] + 2) * (px[0] + 2)
    out.append((tiles_max, rows, wgs, (nz, ny, nx), (pz[0], py[0], px[0]),
                ideal_tiles / tiles_sum, ideal_tiles / (wgs * tiles_max)))
  out.sort()
  print('split        box      wgs  max tiles  staged rows  tile occupancy  '
        'critical-path efficiency')
  for tiles_max, rows, wgs, split, box, occ, eff in out[:12]:
    print('%-12r %-8r %3d  %9d  %11d  %14.2f  %24.2f' % (
        split, box, wgs, tiles_max, rows, occ, eff))


if __name__ == '__main__':
  main()
