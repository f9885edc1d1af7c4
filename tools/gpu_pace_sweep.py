#!/usr/bin/env python
"""The resident conv stack (conv32ps, flow = 2) paced: microseconds per stack (conv0_a +
the resident launch, back to back on one FoV) against engine option flow_pace (10-ns
ticks between two convs of a workgroup) and flow_pace_tail (the tail workgroups' offset
inside the period).  0 = free-running.

  python tools/gpu_pace_sweep.py [--pace 0,660,...] [--tail 0,200,...]
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


def stack_us(eng, reps=300):
  eng.forward_resident(1, 100)
  eng.synchronize()
  t0 = time.perf_counter()
  eng.forward_resident(1, reps)
  eng.synchronize()
  return (time.perf_counter() - t0) / reps * 1e6


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--pace', default='0,600,640,680,720,760')
  ap.add_argument('--tail', default='0')
  ap.add_argument('--spread', default='0,100', help='flow_pace_spread as percent of the pace')
  ap.add_argument('--rounds', type=int, default=2)
  args = ap.parse_args()
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  eng = hip_engine.HipEngine.from_model(model, max_batch=1)
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
  seed = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
  ref = eng.predict(seed, img)
  print('measured beat (flow_pace -1): %d ticks; the tuner saw %.2f us free-running, %.2f at its best beat'
        % (eng.get_option('flow_pace_now'), eng.get_option('flow_pace_free_ns') / 1e3,
           eng.get_option('flow_pace_best_ns') / 1e3))
  t_end = time.perf_counter() + 2.0
  while time.perf_counter() < t_end:
    eng.forward_resident(1, 20)
    eng.synchronize()
  paces = [int(x) for x in args.pace.split(',')]
  tails = [int(x) for x in args.tail.split(',')]
  for r in range(args.rounds):
    eng.set_option('flow_pace', -1)
    eng.set_option('flow_pace_spread', -1)
    eng.set_option('flow_pace_tail', 0)
    print('round %d measured beat %d: %.2f us per stack' % (r, eng.get_option('flow_pace_now'),
                                                            stack_us(eng)), flush=True)
    for p in paces:
      row = []
      for t in (tails if p else [0]):
        for sp in ([int(x) for x in args.spread.split(',')] if p else [0]):
          eng.set_option('flow_pace', p)
          eng.set_option('flow_pace_tail', t)
          eng.set_option('flow_pace_spread', p * sp // 100)
          row.append('tail %d spread %d %%: %6.2f' % (t, sp, stack_us(eng)))
      print('round %d pace %4d | %s' % (r, p, ' | '.join(row)), flush=True)
  eng.set_option('flow_pace', paces[-1])
  got = eng.predict(seed, img)
  print('paced logits identical to free-running:', np.array_equal(got, ref),
        '; timeouts', eng.get_option('stat_flow_timeouts'))
  eng.close()


if __name__ == '__main__':
  main()
