#!/usr/bin/env python
"""Same-process A/B of engine option sets ("arms") on the resident conv stack:
microseconds per stack at several batch sizes (arms interleaved round by round:
boxes behind gpurun differ by several per cent, a process does not), bit
equality of the logits between the arms, optionally the in-kernel clock stamps
of one launch.

  python tools/gpu_ab.py --arm conv_variant=9 --arm conv_variant=9,use_graph=1
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
from oracle import ffn_oracle  # noqa: E402  (random weights for other geometries)


def parse_arm(text):
  out = []
  for item in text.split(','):
    if item:
      name, _, value = item.partition('=')
      out.append((name, int(value)))
  return out


def apply(eng, arm):
  for name, value in arm:
    eng.set_option(name, value)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--arm', action='append', default=[],
                  help='name=value[,name=value...]; the first arm is the base')
  ap.add_argument('--batch', type=int, nargs='+', default=[1])
  ap.add_argument('--rounds', type=int, default=7)
  ap.add_argument('--repeats', type=int, default=60)
  ap.add_argument('--fov', type=int, nargs=3, default=[33, 33, 33], help='xyz')
  ap.add_argument('--deltas', type=int, nargs=3, default=[8, 8, 8], help='xyz')
  ap.add_argument('--depth', type=int, default=12)
  ap.add_argument('--warm-seconds', type=float, default=4.0)
  ap.add_argument('--no-verify', action='store_true',
                  help='skip the logit comparison of the arms (timing ablations '
                  'produce garbage, and a NaN would send the engine to its '
                  'exact-f32 fallback)')
  ap.add_argument('--clocks', action='store_true',
                  help='in-kernel clock stamps of layer 3 per arm (debug_clock 1)')
  args = ap.parse_args()
  arms = [parse_arm(a) for a in args.arm] or [[]]
  model = convstack_3d.ConvStack3DFFNModel(fov_size=args.fov, deltas=args.deltas,
                                           depth=args.depth)
  if args.fov == [33, 33, 33] and args.depth == 12:
    model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  else:
    model.set_variables(ffn_oracle.random_weights(args.depth, seed=18, stddev=0.03))
  maxb = max(args.batch)
  eng = hip_engine.HipEngine.from_model(model, max_batch=maxb)
  zyx = args.fov[::-1]
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, [maxb] + zyx).astype(np.float32)
  seed = rng.normal(0, 1, [maxb] + zyx).astype(np.float32)
  nb = min(2, maxb)
  base = None
  eng.predict(seed, img)  # fills the staging buffers of every slot
  for k, arm in enumerate(arms):
    if args.no_verify:
      break
    apply(eng, arm)
    out = eng.predict(seed[:nb], img[:nb])
    again = eng.predict(seed[:nb], img[:nb])
    one = eng.predict(seed[:1], img[:1])
    if base is None:
      base = (out, one)
    print('arm %d %s: deterministic %s, vs arm 0: n=%d %s (max |d| %.3g), n=1 %s '
          '(max |d| %.3g)' % (k, dict(arm), np.array_equal(out, again), nb,
                              'bit-identical' if np.array_equal(out, base[0])
                              else 'differs', np.abs(out - base[0]).max(),
                              'bit-identical' if np.array_equal(one, base[1])
                              else 'differs', np.abs(one - base[1]).max()))
  # a GPU that has been idle starts at low clocks (a whole session of this
  # script can be shorter than the ramp): run for a few seconds first
  t_warm = time.perf_counter()
  while time.perf_counter() - t_warm < args.warm_seconds:
    eng.forward_resident(maxb, 20)
    eng.synchronize()
  layers = 2 * args.depth - 1
  vox = zyx[0] * zyx[1] * zyx[2]
  flop = 2.0 * 27 * 32 * 32 * vox * layers
  for b in args.batch:
    times = [[] for _ in arms]
    for _ in range(args.rounds):
      for k, arm in enumerate(arms):
        apply(eng, arm)
        eng.forward_resident(b, 3)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.forward_resident(b, args.repeats)
        eng.synchronize()
        times[k].append((time.perf_counter() - t0) / args.repeats)
    for k, arm in enumerate(arms):
      t = np.array(times[k]) * 1e6
      med = float(np.median(t))
      # (the stack = conv0_a + the conv launches; conv0_a's share is ~1 / 25)
      print('batch %2d arm %d %s: median %7.1f us/stack (min %7.1f, max %7.1f)  '
            '%6.2f us per FoV-launch  %6.1f TF/s algorithmic' %
            (b, k, dict(arm), med, t.min(), t.max(), med / (layers + 1) / b,
             b * flop / (med * 1e-6) / 1e12))
  if args.clocks:
    for k, arm in enumerate(arms):
      apply(eng, arm)
      eng.set_option('debug_clock', 1)
      eng.forward_resident(1, 2)
      c = eng.debug_clocks()
      eng.set_option('debug_clock', 0)
      for w in range(4):
        print('arm %d wave %d: prologue %d  loop %d  epilogue %d  total %d cycles, '
              'wall %.2f us' % (k, w, c[w, 1] - c[w, 0], c[w, 2] - c[w, 1],
                                c[w, 3] - c[w, 2], c[w, 3] - c[w, 0],
                                (c[w, 5] - c[w, 4]) / 100.0))
  try:
    print('stat_flow_timeouts', eng.get_option('stat_flow_timeouts'))
  except Exception as exc:  # an older library
    print('stat_flow_timeouts: n/a (%s)' % exc)
  eng.close()


if __name__ == '__main__':
  main()
