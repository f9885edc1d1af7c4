#!/usr/bin/env python
"""When the roles of ONE single-FoV step ran (engine option debug_fused_trace): the
resident stack, then inside the fused launch the faces block (-> record published), the
paste blocks and the next step's conv0_a blocks -- in-kernel wall-clock stamps relative to
the stack's first workgroup, a few steps sampled out of a running segment of the 250^3
bench volume.

  python tools/gpu_step_trace.py [--samples 12] [--ahead 1]

--ahead 1 (engine option stack_ahead): the stack inside the traced window is the NEXT step's,
queued behind the fused launch; the stamps are then relative to the faces block's entry, and
the figure of interest is 'next stack: first workgroup entry' against 'last conv0_a block end'.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--samples', type=int, default=12)
  ap.add_argument('--ahead', type=int, default=0)
  ap.add_argument('--twice', type=int, default=0,
                  help='engine option debug_fused_twice: the fused launch made twice, the second one traced')
  args = ap.parse_args()
  from ffn_amd import synthetic
  from ffn_amd.inference import executor, inference, inference_utils, movement
  bargs = bench.build_parser().parse_args([])
  bench.configure(bargs)
  model = bench.load_model()
  request = bench.make_request()
  counters = inference_utils.Counters()
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), model, model.info, None,
                                  counters, 1)
  eng = exe.engine
  eng.set_option('stack_ahead', args.ahead)
  eng.set_option('debug_fused_twice', args.twice)
  image = synthetic.normalize(bench.bench_volume((250, 250, 250), 1234))
  canvas = inference.DeviceCanvas(
      model.info, exe.get_client(counters, direct=True), image, request.inference_options,
      counters=counters, movement_policy_fn=movement.get_policy_fn(request, model.info))
  names = {4: 'faces block entry', 8: 'record published', 5: 'first paste block entry',
           9: 'last paste block end', 6: 'first conv0_a block entry',
           10: 'last conv0_a block end', 11: 'stack: last workgroup end',
           12: 'faces: step count known', 13: 'faces: face maxima reduced',
           14: 'faces: record built', 16: 'conv0_a: position chosen (last block)',
           17: 'conv0_a: tile staged (last block)', 18: 'conv0_a: MFMAs done (last block)',
           19: 'faces: every load issued', 20: 'conv0_a block 0: loads issued',
           21: 'conv0_a block 0: count known', 22: 'conv0_a block 0: position chosen',
           23: 'conv0_a: last block entry', 24: 'conv0_a block 0: kernel arguments in'}
  if args.ahead:
    names[25] = 'the step\'s own stack: last workgroup end'
    names[26] = 'the step\'s own stack: first workgroup entry'
    names[7] = 'NEXT stack: first workgroup entry'
    names[11] = 'NEXT stack: last workgroup end'
  rows = []
  pair_rows = []
  import json
  fx = np.load(bench.full_fixture(1234))
  origins = json.loads(str(fx['origins']))
  pos, iters = max(((tuple(v[0]), v[1]) for v in origins.values()), key=lambda t: t[1])
  print('segment from %s (%d steps in the reference-minted run)' % (pos, iters))
  rng = np.random.RandomState(0)
  for k in range(args.samples):
    eng.set_option('debug_fused_trace', int(rng.randint(4, 12)))
    canvas.init_seed(pos)
    n_steps = canvas.segment_at(pos)
    eng.synchronize()
    if k == 0:
      print('first sampled segment: %s steps; update_at-calls %d' % (
          n_steps, counters['update_at-calls'].value))
    row = {k2: eng.get_option('debug_fused_stamp_%d' % k2) / 100.0 for k2 in names}
    rows.append(row)
    pair_rows.append(eng.get_option('debug_fused_stamp_27') if args.ahead else 0)
  if args.ahead:
    order = [26, 25, 4, 19, 12, 13, 14, 8, 5, 9, 6, 23, 24, 20, 21, 22, 16, 17, 18, 10, 7, 11]
    print('stack_ahead: us after the faces block of the step entered (median, min .. max over %d '
          'sampled steps); used %d, wasted %d' % (len(rows), eng.get_option('stat_ahead_used'),
                                                 eng.get_option('stat_ahead_wasted')))
    for k2 in order:
      v = np.array([r[k2] for r in rows])
      print('  %-40s %8.2f   (%.2f .. %.2f)' % (names[k2], np.median(v), v.min(), v.max()))
    pairs = [int(round(eng_pairs)) for eng_pairs in pair_rows]
    print('  main workgroups of the step\'s own stack that shared their CU with another main workgroup: %s' % pairs)
    gap = np.array([r[7] - r[10] for r in rows])
    print('  last conv0_a block end -> next stack\'s first workgroup: %.2f us (%.2f .. %.2f)' % (
        np.median(gap), gap.min(), gap.max()))
    print('  turn-around (record published -> next stack entry), run-long mean: %.2f us' % (
        eng.get_option('stat_turn_gpu_ns') / 1e3))
  else:
    order = [11, 4, 19, 12, 13, 14, 8, 5, 9, 6, 20, 21, 22, 16, 17, 18, 10]
    print('us after the first workgroup of the step\'s resident stack started (median, min .. max '
          'over %d sampled steps)' % len(rows))
    for k2 in order:
      v = np.array([r[k2] for r in rows])
      print('  %-40s %8.2f   (%.2f .. %.2f)' % (names[k2], np.median(v), v.min(), v.max()))
    end = np.array([r[11] for r in rows])
    for k2 in (4, 19, 12, 13, 14, 8, 5, 9, 6, 20, 21, 22, 16, 17, 18, 10):
      v = np.array([r[k2] for r in rows]) - end
      print('  after the stack\'s end: %-40s %7.2f' % (names[k2], np.median(v)))
  canvas.close()
  eng.close()


if __name__ == '__main__':
  main()
