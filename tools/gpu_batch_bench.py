#!/usr/bin/env python3
"""Config C3 shape: B concurrent canvases on one MI355X, FoV steps batched by the
executor's server thread into single ffn_canvas_step(n, ...) calls.

  python tools/gpu_batch_bench.py --canvases 8 --batch 8 --size 160 --steps 400
"""
import argparse
import functools
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from ffn_amd import synthetic  # noqa: E402
from ffn_amd.inference import executor, inference, inference_utils  # noqa: E402
from ffn_amd.inference import movement, seed as seed_lib  # noqa: E402


class _Done(Exception):
  pass


def run_driver(args, model, request, counters):
  """Same workload through the single-threaded MultiCanvasDriver."""
  for overlap in (False, True):
    exe = executor.HipBatchExecutor(executor.ExecutorInterface(), model,
                                    model.info, None, counters, args.batch)
    jobs = []
    for k in range(args.canvases):
      vol = synthetic.normalize(synthetic.cells_volume((args.size,) * 3,
                                                       seed=100 + k))
      sub = counters.get_sub_counters()
      canvas = inference.DeviceCanvas(
          model.info, exe.get_client(sub, direct=True), vol,
          request.inference_options, counters=sub,
          movement_policy_fn=movement.get_policy_fn(request, model.info))
      jobs.append((canvas, functools.partial(seed_lib.PolicyGrid3d, step=16,
                                             offsets=(0, 8, 4, 12))))
    drv = inference.MultiCanvasDriver(exe.engine, args.batch, overlap=overlap,
                                      max_steps_per_canvas=args.steps)
    t0 = time.perf_counter()
    drv.run(jobs)
    exe.engine.synchronize()
    dt = time.perf_counter() - t0
    print('driver overlap=%d canvases %d batch %d: %d steps in %.2f s = %.1f '
          'FoV-steps/s; %d engine calls (mean fill %.2f)' %
          (overlap, args.canvases, args.batch, drv.steps, dt, drv.steps / dt,
           drv.calls, drv.steps / max(drv.calls, 1)))
    for canvas, _ in jobs:
      canvas.close()
    exe.engine.close()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--canvases', type=int, default=8)
  ap.add_argument('--batch', type=int, default=8)
  ap.add_argument('--size', type=int, default=160)
  ap.add_argument('--steps', type=int, default=400, help='per canvas')
  ap.add_argument('--mode', choices=['threads', 'driver'], default='threads')
  args = ap.parse_args()
  model = bench.load_model()
  request = bench.make_request()
  counters = inference_utils.Counters()
  if args.mode == 'driver':
    return run_driver(args, model, request, counters)
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), model,
                                  model.info, None, counters, args.batch,
                                  expected_clients=args.canvases)
  exe.start_server()
  vols = [synthetic.normalize(synthetic.cells_volume((args.size,) * 3,
                                                     seed=100 + k))
          for k in range(args.canvases)]
  start = threading.Barrier(args.canvases + 1)
  done_steps = [0] * args.canvases

  def work(k):
    sub = counters.get_sub_counters()

    class C(inference.DeviceCanvas):

      def update_at(self, pos):
        r = super().update_at(pos)
        done_steps[k] += 1
        if done_steps[k] >= args.steps:
          raise _Done()
        return r

    canvas = C(model.info, exe.get_client(sub), vols[k],
               request.inference_options, counters=sub,
               movement_policy_fn=movement.get_policy_fn(request, model.info))
    start.wait()
    try:
      canvas.segment_all(seed_policy=functools.partial(
          seed_lib.PolicyGrid3d, step=16, offsets=(0, 8, 4, 12)))
    except _Done:
      pass
    canvas._deregister_client()

  threads = [threading.Thread(target=work, args=(k,), daemon=True)
             for k in range(args.canvases)]
  for t in threads:
    t.start()
  start.wait()
  t0 = time.perf_counter()
  for t in threads:
    t.join()
  dt = time.perf_counter() - t0
  exe.stop_server()
  total = sum(done_steps)
  calls = counters['executor-inference-calls'].value
  print('canvases %d batch %d: %d steps in %.2f s = %.1f FoV-steps/s; '
        '%d engine calls (mean fill %.2f)' %
        (args.canvases, args.batch, total, dt, total / dt, calls,
         total / max(calls, 1)))


if __name__ == '__main__':
  main()
