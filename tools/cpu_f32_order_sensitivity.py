#!/usr/bin/env python3
"""How far do two CORRECT f32 CPU implementations of the conv stack drift apart
over a whole 250^3 run?  The reference's FoV loop feeds every step's logits back
into the next steps' inputs, so float noise is not just added up: it can be
amplified until a threshold decision flips.  This tool runs the oracle canvas
loop (oracle/ffn_oracle.OracleCanvas, pinned to the reference's Canvas by
tests/test_oracle.py) on the cells250 fixture workload with the torch-CPU /
oneDNN conv stack -- plain f32, a different summation order than the C oracle,
which is what TensorFlow's CPU kernels would also be -- and compares it with
the reference-minted fixture (reference Canvas + C oracle forward):
first differing FoV position, move-score differences along the way, IoU.

  python tools/cpu_f32_order_sensitivity.py [--threads 8] [--max-steps 4000]
"""
import argparse
import functools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ffn_amd import synthetic  # noqa: E402
from oracle import ffn_oracle  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--threads', type=int, default=os.cpu_count() or 1)
  ap.add_argument('--max-steps', type=int, default=10**9)
  args = ap.parse_args()
  g = np.load(os.path.join(ROOT, 'tests/golden/ref_canvas_cells250.npz'))
  with np.load(os.path.join(ROOT, 'tests/golden/fib25_weights.npz')) as d:
    variables = {k: d[k] for k in d.files}
  blob = ffn_oracle.weights_blob(variables, 12)
  vol = synthetic.cells_volume((250, 250, 250), seed=1234)
  image = synthetic.normalize(vol)
  oc = ffn_oracle.OracleCanvas(image, blob, 12, (33, 33, 33), (8, 8, 8),
                               ffn_oracle.Options())
  oc.forward_fn = functools.partial(ffn_oracle.forward_torch,
                                    variables=variables, depth=12,
                                    threads=args.threads)

  class _Stop(Exception):
    pass

  inner = oc.update_at
  n = [0]

  def counted(pos):
    if n[0] >= args.max_steps:
      raise _Stop()
    n[0] += 1
    return inner(pos)

  oc.update_at = counted
  t0 = time.time()
  try:
    oc.segment_all(g['seeds'])
  except _Stop:
    pass
  wall = time.time() - t0
  want_steps = [tuple(int(v) for v in p) for p in g['steps']]
  want_moves, off = [], 0
  for nm in g['n_moves']:
    nm = int(nm)
    want_moves.append([(float(g['move_scores'][off + j]),
                        tuple(int(v) for v in g['move_coords'][off + j]))
                       for j in range(nm)])
    off += nm
  got_steps = [tuple(int(v) for v in p) for p, _ in oc.trace]
  got_moves = [[(s, tuple(int(o[i]) + int(p[i]) for i in range(3)))
                for s, o in m] for p, m in oc.trace]
  m = min(len(got_steps), len(want_steps))
  bad_step = next((k for k in range(m) if got_steps[k] != want_steps[k]), None)
  bad_move = next((k for k in range(m)
                   if [c for _, c in got_moves[k]] != [c for _, c in want_moves[k]]),
                  None)
  upto = bad_move if bad_move is not None else m
  errs = np.zeros(max(upto, 1))
  for k in range(upto):
    for (s, _), (w, _) in zip(got_moves[k], want_moves[k]):
      errs[k] = max(errs[k], abs(s - w))
  print('torch-oneDNN f32 forward vs the fixture (C oracle forward): %d steps '
        '(fixture %d), %.0f s on %d threads' % (len(got_steps), len(want_steps),
                                                wall, args.threads))
  print('first FoV position mismatch at step %s; first move-list mismatch at '
        'step %s' % (bad_step, bad_move))
  for lim in (2e-5, 1e-4, 1e-3):
    idx = np.nonzero(errs > lim)[0]
    print('move-score difference > %g first at step %s (%d steps in all)' %
          (lim, idx[0] if idx.size else None, idx.size))
  print('difference by 100-step block (max): %s' % ' '.join(
      '%.1e' % errs[b:b + 100].max() for b in range(0, upto, 100)))
  if len(got_steps) >= len(want_steps) or args.max_steps >= 10**9:
    seg = oc.segmentation
    want = g['segmentation'].astype(np.int32)
    inter = np.sum((seg > 0) & (want > 0) & (seg == want))
    union = np.sum((seg > 0) | (want > 0))
    print('labelled IoU %.6f; voxels %d vs %d' % (
        inter / max(union, 1), int((seg > 0).sum()), int((want > 0).sum())))


if __name__ == '__main__':
  main()
