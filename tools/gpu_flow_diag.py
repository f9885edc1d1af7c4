#!/usr/bin/env python
"""Where do the logits of a FLOW mode differ from the plain launches?  (debugging
aid of round 4: per z plane / per 128-voxel chunk counts of differing voxels)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ffn_amd import engine as hip_engine  # noqa: E402
from ffn_amd.training.models import convstack_3d  # noqa: E402
from oracle import ffn_oracle  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--depth', type=int, nargs='+', default=[2, 3, 12])
  ap.add_argument('--flow', type=int, default=2)
  ap.add_argument('--flow-debug', type=int, nargs='+', default=[16])
  args = ap.parse_args()
  for depth in args.depth:
    model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8],
                                             depth=depth)
    if depth == 12:
      model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
    else:
      model.set_variables(ffn_oracle.random_weights(depth, seed=18, stddev=0.05))
    eng = hip_engine.HipEngine.from_model(model, max_batch=1)
    rng = np.random.RandomState(0)
    img = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
    seed = rng.normal(0, 1, [1, 33, 33, 33]).astype(np.float32)
    eng.set_option('flow', 0)
    base = eng.predict(seed, img)[0]
    for dbg in args.flow_debug:
      eng.set_option('flow', args.flow)
      eng.set_option('flow_debug', dbg)
      for rep in range(3):
        got = eng.predict(seed, img)[0]
        bad = (got != base).reshape(-1)
        nb = int(bad.sum())
        print('depth %d flow %d debug %d rep %d: %d voxels differ, max |d| %.3g' %
              (depth, args.flow, dbg, rep, nb, np.abs(got - base).max()))
        if nb:
          idx = np.nonzero(bad)[0]
          chunks = np.bincount(idx // 128, minlength=281)
          print('  dense index range %d .. %d; chunks with errors: %d of 281 '
                '(main < 256: %d, tail: %d); first chunks %s' %
                (idx.min(), idx.max(), int((chunks > 0).sum()),
                 int((chunks[:256] > 0).sum()), int((chunks[256:] > 0).sum()),
                 np.nonzero(chunks)[0][:24].tolist()))
          print('  differing voxels per z plane:', np.bincount(idx // 1089, minlength=33).tolist())
      print('  timeouts', eng.get_option('stat_flow_timeouts'))
    eng.set_option('flow', 0)
    eng.close()


if __name__ == '__main__':
  main()
