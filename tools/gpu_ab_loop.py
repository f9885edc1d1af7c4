#!/usr/bin/env python3
"""Same-process A/B of engine option sets on the END-TO-END single-canvas loop
(`ffn_canvas_segment_at`, the drive of bench.py's headline line): one segment of
the 250^3 bench volume is advanced in legs of --leg-steps FoV steps, the arms
taking turns leg by leg, so that clock drift and the box are common to all arms
(two bench.py runs of the SAME options differ by up to 3 %).

  python tools/gpu_ab_loop.py --arm speculate=0 --arm speculate=1,fuse_paste=1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from ffn_amd import synthetic  # noqa: E402
from ffn_amd.inference import executor  # noqa: E402
from ffn_amd.inference import inference  # noqa: E402
from ffn_amd.inference import inference_utils  # noqa: E402
from ffn_amd.inference import movement  # noqa: E402
from ffn_amd.training.models import convstack_3d  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--arm', action='append', default=[])
  ap.add_argument('--leg-steps', type=int, default=150)
  ap.add_argument('--legs', type=int, default=12, help='per arm')
  ap.add_argument('--warm-legs', type=int, default=4)
  args = ap.parse_args()
  arms = [[(kv.split('=')[0], int(kv.split('=')[1])) for kv in a.split(',') if kv]
          for a in args.arm] or [[]]
  model = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33], deltas=[8, 8, 8],
                                           depth=12)
  model.load_checkpoint(os.path.join(ROOT, 'tests/golden/fib25_weights.npz'))
  counters = inference_utils.Counters()
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), model, model.info,
                                  None, counters, 1, device_id=0)
  eng = exe.engine
  request = bench.make_request()
  vol = synthetic.cells_volume((250, 250, 250), seed=1234)
  seeds = np.load(os.path.join(ROOT, 'tests/golden/ref_canvas_cells250_onednn.npz'))[
      'seeds']
  sub = counters.get_sub_counters()
  canvas = inference.DeviceCanvas(
      model.info, exe.get_client(sub, direct=True), synthetic.normalize(vol),
      request.inference_options, counters=sub,
      movement_policy_fn=movement.get_policy_fn(request, model.info))
  assert canvas._native_loop_ok()
  times = [[] for _ in arms]
  board = [[] for _ in arms]
  import itertools
  seed_iter = itertools.cycle([tuple(int(v) for v in s) for s in seeds])
  state = {'active': False, 'start': None}

  def leg(n):
    """n FoV steps (several segments if need be); seconds."""
    done = 0
    t0 = time.perf_counter()
    while done < n:
      if not state['active']:
        state['start'] = next(seed_iter)
        if not canvas.is_valid_pos(state['start'], ignore_move_threshold=True):
          continue
        got = canvas._segment_at_native(state['start'], max_steps=n - done)
      else:
        got = canvas._segment_at_native(state['start'], max_steps=n - done,
                                        resume=True)
      state['active'] = canvas._native_active
      done += got
    return time.perf_counter() - t0, done

  for _ in range(args.warm_legs):
    leg(args.leg_steps)
  for r in range(args.legs):
    for k, arm in enumerate(arms):
      for name, value in arm:
        eng.set_option(name, value)
      with bench.BoardSampler(0, period_s=0.002) as bs:
        dt, done = leg(args.leg_steps)
      times[k].append(dt / done * 1e6)
      if bs.power_w:
        half = len(bs.power_w) // 2
        board[k].append((float(np.median(bs.power_w[half:])), float(np.median(bs.sclk_mhz[half:]))))
  for k, arm in enumerate(arms):
    t = np.array(times[k])
    print('arm %d %s: median %.2f us/step (min %.2f, max %.2f, mean %.2f) over %d '
          'legs of %d steps  = %.0f FoV-steps/s' % (
              k, dict(arm), np.median(t), t.min(), t.max(), t.mean(), len(t),
              args.leg_steps, 1e6 / np.median(t)))
    if board[k]:
      b = np.array(board[k])
      print('      board %.0f W, sclk %.0f MHz (medians of the legs\' second halves; hwmon)' % (
          np.median(b[:, 0]), np.median(b[:, 1])))
  eng.close()


if __name__ == '__main__':
  main()
