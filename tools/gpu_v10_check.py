#!/usr/bin/env python
"""conv_variant 10 (conv32hs / conv32h: 80-voxel workgroups, two hand-off chains per SIMD)
on the GPU: logits against the C oracle and against conv_variant 9; the resident stack
against its per-layer launches (same bits); microseconds per stack, free-running and
paced (engine option flow_pace, 10-ns ticks).

  python tools/gpu_v10_check.py [--pace 0,520,560,...] [--no-oracle]
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
  ap.add_argument('--pace', default='0,480,520,560,600,640,680')
  ap.add_argument('--no-oracle', action='store_true')
  ap.add_argument('--rounds', type=int, default=2)
  args = ap.parse_args()
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  eng = hip_engine.HipEngine.from_model(model, max_batch=1)
  rng = np.random.RandomState(0)
  img = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
  seed = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
  ref9 = eng.predict(seed, img)
  eng.set_option('conv_variant', 10)
  eng.set_option('flow', 0)
  per_layer = eng.predict(seed, img)
  eng.set_option('flow', 2)
  resident = eng.predict(seed, img)
  print('variant 10 per-layer launches vs variant 9: max |d| %.3g' % np.abs(per_layer - ref9).max())
  print('variant 10 resident vs its per-layer launches: identical %s (max |d| %.3g); timeouts %d'
        % (np.array_equal(resident, per_layer), np.abs(resident - per_layer).max(),
           eng.get_option('stat_flow_timeouts')))
  if not args.no_oracle:
    from oracle import ffn_oracle
    with np.load(os.path.join(ROOT, 'tests/golden/fib25_weights.npz')) as d:
      variables = {k: d[k] for k in d.files}
    blob = ffn_oracle.weights_blob(variables, 12)
    want = ffn_oracle.forward(img[0], seed[0], blob, 12)
    print('max |d logit| vs the C oracle: variant 9 %.3g, variant 10 %.3g' % (
        np.abs(ref9[0] - want).max(), np.abs(resident[0] - want).max()))
    bad = np.argwhere(np.abs(resident[0] - want) > 1e-4)
    if len(bad):
      print('  %d voxels off by > 1e-4; first %s; dense index %% 80 histogram of the bad ones: %s'
            % (len(bad), bad[:5].tolist(),
               np.bincount((bad[:, 0] * 1089 + bad[:, 1] * 33 + bad[:, 2]) % 80 // 16,
                           minlength=5).tolist()))
  t_end = time.perf_counter() + 2.0
  while time.perf_counter() < t_end:
    eng.forward_resident(1, 20)
    eng.synchronize()
  for r in range(args.rounds):
    eng.set_option('conv_variant', 9)
    eng.set_option('flow_pace', 0)
    base = stack_us(eng)
    eng.set_option('conv_variant', 10)
    row = []
    for p in [int(x) for x in args.pace.split(',')]:
      eng.set_option('flow_pace', p)
      row.append('%d: %.1f' % (p, stack_us(eng)))
    print('round %d us per stack: variant 9 %.1f | variant 10 by flow_pace %s' % (
        r, base, ' | '.join(row)), flush=True)
  got = eng.predict(seed, img)
  print('paced logits identical to free-running:', np.array_equal(got, resident),
        '; timeouts', eng.get_option('stat_flow_timeouts'))
  eng.close()


if __name__ == '__main__':
  main()
