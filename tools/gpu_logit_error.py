#!/usr/bin/env python3
"""Logit error of the split-product kernels on REAL canvas states (not random
FoVs): drives the 250^3 bench workload with the exact-f32 kernel, samples the
(image, seed) FoV in front of selected steps, and runs the stateless predict on
each sample with every variant.  Prints max |logit - exact f32 kernel| together
with the magnitude of the inputs; saves the samples for offline analysis.

  python tools/gpu_logit_error.py [--steps 1600] [--every 100]
"""
import argparse
import functools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import test_gpu_round2 as t2  # noqa: E402


class _Stop(Exception):
  pass


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=1600)
  ap.add_argument('--every', type=int, default=100)
  ap.add_argument('--variants', type=int, nargs='+', default=[6, 8, 9])
  ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out',
                                                'r02_fov_samples.npz'))
  args = ap.parse_args()
  from ffn_amd import synthetic
  from ffn_amd.inference import executor, inference_utils
  from ffn_amd.inference import seed as seed_lib
  from ffn_amd.training.models import convstack_3d
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33],
                                           deltas=[8, 8, 8], depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), model,
                                  model.info, None, inference_utils.Counters(),
                                  1, device_id=0)
  eng = exe.engine
  g = np.load(os.path.join(ROOT, 'tests/golden/ref_canvas_cells250.npz'))
  vol = synthetic.cells_volume((250, 250, 250), seed=1234)
  image = synthetic.normalize(vol)
  eng.set_option('conv_variant', 2)
  canvas = t2._device_canvas(exe, model, image)
  pad = float(canvas.options.pad_value)
  pad_logit = np.float32(pad)  # (the Canvas keeps its options in logit space)
  samples = []
  inner = canvas.update_at
  count = [0]

  def rec(pos):
    k = count[0]
    if k % args.every == 0 or k == 1554:
      lo = [int(p) - 16 for p in pos]
      sl = tuple(slice(l, l + 33) for l in lo)
      seed = np.array(canvas.seed[sl], np.float32)
      samples.append((k, tuple(int(p) for p in pos), image[sl].copy(), seed))
    count[0] += 1
    if k >= args.steps:
      raise _Stop()
    return inner(pos)

  canvas.update_at = rec
  try:
    canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                     coords=g['seeds']))
  except _Stop:
    pass
  canvas.close()
  print('%d samples' % len(samples))
  keep = {}
  for k, pos, img, seed in samples:
    s = np.where(np.isnan(seed), pad_logit, seed).astype(np.float32)
    eng.set_option('conv_variant', 2)
    ref = eng.predict(s[None], img[None])[0]
    line = 'step %5d pos %-15s max|seed| %7.2f max|logit| %7.2f ' % (
        k, pos, np.abs(s).max(), np.abs(ref).max())
    for v in args.variants:
      eng.set_option('conv_variant', v)
      out = eng.predict(s[None], img[None])[0]
      line += ' v%d: %.3g' % (v, np.abs(out - ref).max())
    print(line)
    keep['img_%d' % k] = img
    keep['seed_%d' % k] = s
    keep['ref_%d' % k] = ref
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  np.savez_compressed(args.out, **keep)
  eng.close()


if __name__ == '__main__':
  main()
