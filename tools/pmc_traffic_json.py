#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter dumps (two separate passes of
the same bench.py command) -> the per-launch HBM traffic of the conv kernels as
a JSON that bench.py reads for `roofline.traffic`.

  python tools/pmc_traffic_json.py FETCH.csv WRITE.csv OUT.json [--depth 12]
      [--fov 33 33 33]

Units (MI355X_MICROARCH.md, HBM section): both counters are KiB per dispatch; on
gfx950 FETCH_SIZE counts the 16-B-per-lane streaming reads these kernels make at
HALF their bytes, so fetched bytes = 2 x FETCH_SIZE x 1024.
"""
import argparse
import collections
import csv
import json


def per_kernel(path, counter):
  agg = collections.defaultdict(list)
  for r in csv.DictReader(open(path)):
    if r.get('Counter_Name') == counter:
      agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
  return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('fetch_csv')
  ap.add_argument('write_csv')
  ap.add_argument('out_json')
  ap.add_argument('--depth', type=int, default=12)
  ap.add_argument('--fov', type=int, nargs=3, default=[33, 33, 33])
  ap.add_argument('--command', default='')
  ap.add_argument('--commit', default='', help='git revision of the tree the capture ran on')
  ap.add_argument('--sq-csv', default='', help='counter dump of an SQ pass '
                  '(SQ_VALU_MFMA_BUSY_CYCLES ...): per-launch means go into the JSON')
  ap.add_argument('--script', default='tools/gpu_profile_r6.sh')
  ap.add_argument('--csrc-sha', default='', help='ffn_amd._lib.csrc_sha() of the tree')
  args = ap.parse_args()
  fetch = per_kernel(args.fetch_csv, 'FETCH_SIZE')
  write = per_kernel(args.write_csv, 'WRITE_SIZE')
  kernels = {}
  for name in sorted(set(fetch) | set(write)):
    if 'conv' not in name and 'faces' not in name and 'paste' not in name:
      continue
    f, nf = fetch.get(name, (0.0, 0))
    w, nw = write.get(name, (0.0, 0))
    kernels[name] = {'fetch_size_kib_raw': round(f, 1), 'write_size_kib': round(w, 1),
                     'dispatches_fetch_pass': nf, 'dispatches_write_pass': nw,
                     'hbm_bytes_per_launch': int((2 * f + w) * 1024)}
  vox = args.fov[0] * args.fov[1] * args.fov[2]
  n = 2 * args.depth - 1
  act = vox * 128  # one 32-channel activation tensor: 128 B per voxel
  wts = 28 * 4096  # one conv's packed weight fragments
  # per conv: its input once + its weights; its output once (the last writes the
  # logits only: 4 B per voxel).  The resident stack (conv32ps) keeps the f32
  # residual stream of the main workgroups in registers: no X traffic.
  algorithmic_stack = n * (act + wts) + (n - 1) * act + vox * 4
  stack = [k for k in kernels if 'conv32ps' in k]
  out = {
      'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, '
                '--kernel-trace only; %s; %s' % (args.script, args.command),
      'commit': args.commit or None,
      'csrc_sha': args.csrc_sha or None,
      'units': 'KiB per dispatch; fetched bytes = 2 x FETCH_SIZE x 1024 on gfx950 '
               '(MI355X_MICROARCH.md, HBM section)',
      'conv_variant': 9,
      'per_kernel': kernels,
      'algorithmic_bytes_per_stack': int(algorithmic_stack),
      'algorithmic_note': '%d convs: each reads its %d-voxel 32-channel input '
                          '(128 B / voxel) and 112 KiB of weight fragments once and '
                          'writes its output once; the last writes 4 B / voxel of '
                          'logits; the residual stream stays in registers'
                          % (n, vox),
  }
  if stack:
    out['kernel'] = stack[0]
    out['traffic_bytes_per_launch'] = kernels[stack[0]]['hbm_bytes_per_launch']
    out['algorithmic_bytes_per_launch'] = int(algorithmic_stack)
    out['traffic_over_algorithmic'] = round(
        out['traffic_bytes_per_launch'] / algorithmic_stack, 3)
  if args.sq_csv and stack:
    # matrix-pipe occupancy of the stack kernel: busy cycles summed over the SIMDs
    sq = collections.defaultdict(list)
    for r in csv.DictReader(open(args.sq_csv)):
      if r['Kernel_Name'].split('(')[0] == stack[0]:
        sq[r['Counter_Name']].append(float(r['Counter_Value']))
    out['sq_per_launch'] = {c: round(sum(v) / len(v), 1) for c, v in sorted(sq.items())}
    out['sq_dispatches'] = len(next(iter(sq.values()))) if sq else 0
  with open(args.out_json, 'w') as f:
    json.dump(out, f, indent=1)
  print(json.dumps({k: out.get(k) for k in (
      'kernel', 'traffic_bytes_per_launch', 'algorithmic_bytes_per_launch',
      'traffic_over_algorithmic')}))


if __name__ == '__main__':
  main()
