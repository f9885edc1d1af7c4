#!/usr/bin/env python3
"""Is a run's departure from a reference-minted trajectory the kernel's error or
the loop's amplification of everybody's rounding?  Drives a whole-volume fixture's
workload on the GPU with the default kernels, samples the (image, seed) FoV in
front of selected steps -- by default around the place where the second 250^3
phantom's runs drift apart (move scores agree to 1.6e-5 up to step 800, differ by
0.13 at step 950: profiles/r05_full250_second_phantom.txt) -- and evaluates each
sample THREE ways: the GPU's stateless predict, torch-CPU / oneDNN f32 (the
forward the fixture was minted with) and torch f64.  Prints max |GPU - f64| next
to max |oneDNN f32 - f64|: on identical inputs both are rounding-sized.

  python tools/gpu_step_error_f64.py [--fixture _onednn_full_s4321] [--lo 780 --hi 960 --every 10]
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
  ap.add_argument('--fixture', default='_onednn_full_s4321')
  ap.add_argument('--lo', type=int, default=780)
  ap.add_argument('--hi', type=int, default=960)
  ap.add_argument('--every', type=int, default=10)
  ap.add_argument('--early', type=int, nargs='*', default=[0, 100, 400])
  args = ap.parse_args()
  from ffn_amd import synthetic
  from ffn_amd.inference import executor, inference_utils
  from ffn_amd.inference import seed as seed_lib
  from ffn_amd.training.models import convstack_3d
  from oracle import ffn_oracle
  with np.load(os.path.join(ROOT, 'tests/golden/fib25_weights.npz')) as d:
    variables = {k: d[k] for k in d.files}
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8],
                                           depth=12)
  model.set_variables(variables)
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), model, model.info, None,
                                  inference_utils.Counters(), 1, device_id=0)
  eng = exe.engine
  g = np.load(os.path.join(ROOT, 'tests/golden/ref_canvas_cells250%s.npz' % args.fixture))
  vol_seed = int(g['volume_seed']) if 'volume_seed' in g.files else 1234
  image = synthetic.normalize(synthetic.cells_volume((250, 250, 250), seed=vol_seed))
  want_steps = [tuple(int(v) for v in p) for p in g['steps']]
  canvas = t2._device_canvas(exe, model, image)
  pad_logit = np.float32(canvas.options.pad_value)
  wanted = set(args.early) | set(range(args.lo, args.hi + 1, args.every))
  samples, count = [], [0]
  inner = canvas.update_at

  def rec(pos):
    k = count[0]
    if k in wanted:
      sl = tuple(slice(int(p) - 16, int(p) + 17) for p in pos)
      samples.append((k, tuple(int(p) for p in pos), image[sl].copy(),
                      np.array(canvas.seed[sl], np.float32)))
    count[0] += 1
    if k > args.hi:
      raise _Stop()
    return inner(pos)

  canvas.update_at = rec
  try:
    canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                     coords=g['seeds']))
  except _Stop:
    pass
  canvas.close()
  print('fixture ref_canvas_cells250%s.npz, volume seed %d; %d samples; default kernels '
        '(conv_variant %d, flow %d)' % (args.fixture, vol_seed, len(samples),
                                        eng.get_option('conv_variant'),
                                        eng.get_option('flow')))
  print('step   pos              same pos as the reference run   max|logit|   '
        'max|GPU - f64|   max|oneDNN f32 - f64|   max|GPU - oneDNN f32|')
  worst = [0.0, 0.0]
  sq = {'GPU split products (conv_variant 9)': [], 'GPU exact f32, sequential (conv_variant 2)': [],
        'torch-CPU / oneDNN f32': [], 'C oracle f32, sequential': []}
  blob = ffn_oracle.weights_blob(variables, 12)
  for k, pos, img, seed in samples:
    s = np.where(np.isnan(seed), pad_logit, seed).astype(np.float32)
    gpu = eng.predict(s[None], img[None])[0]
    eng.set_option('conv_variant', 2)
    gpu2 = eng.predict(s[None], img[None])[0]
    eng.set_option('conv_variant', 9)
    f32 = ffn_oracle.forward_torch(img, s, variables, 12, threads=16)
    f64 = ffn_oracle.forward_torch(img, s, variables, 12, threads=16, f64=True)
    c32 = ffn_oracle.forward(img, s, blob, 12)
    for name, arr in zip(sq, (gpu, gpu2, f32, c32)):
      sq[name].append((arr.astype(np.float64) - f64).ravel())
    e_gpu, e_f32 = float(np.abs(gpu - f64).max()), float(np.abs(f32 - f64).max())
    worst = [max(worst[0], e_gpu), max(worst[1], e_f32)]
    print('%5d  %-16s %-30s %8.2f     %.3g        %.3g                %.3g' % (
        k, pos, k < len(want_steps) and want_steps[k] == pos, np.abs(f64).max(), e_gpu,
        e_f32, float(np.abs(gpu - f32).max())))
  print('worst over the samples: GPU vs f64 %.3g, oneDNN f32 vs f64 %.3g' % tuple(worst))
  print('error against the f64 forward (its logits rounded to f32) over all %d x 35,937 logits:'
        % len(samples))
  for name, errs in sq.items():
    e = np.concatenate(errs)
    print('  %-44s rms %.3g   mean %+.3g   99.9 %% %.3g   max %.3g   exactly equal %.1f %%' % (
        name, np.sqrt(np.mean(e * e)), e.mean(), np.percentile(np.abs(e), 99.9),
        np.abs(e).max(), 100.0 * np.mean(e == 0)))
  eng.close()


if __name__ == '__main__':
  main()
