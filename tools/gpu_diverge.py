#!/usr/bin/env python3
"""Where (and by how much) a split-product kernel's 250^3 run leaves the
reference-minted trajectory (tests/golden/ref_canvas_cells250.npz): first FoV
position that differs, first queued-move list that differs (with the scores on
both sides and their distance to the move threshold), IoU of the final
segmentations.

  python tools/gpu_diverge.py [--variants 2 4 6]
"""
import argparse
import functools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import test_gpu_round2 as t2  # noqa: E402  (canvas / request helpers)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--variants', type=int, nargs='+', default=[2, 9])
  ap.add_argument('--fixture', default='',
                  help="suffix of tests/golden/ref_canvas_cells250<suffix>.npz: '' "
                  "(C oracle forward), '_onednn', '_f64', '_onednn_full' (the "
                  'WHOLE volume: every seed of the grid, tools/make_golden.py '
                  '--only cells250 --forward onednn --num-seeds 0 --tag _full)')
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
  g = np.load(os.path.join(ROOT, 'tests/golden/ref_canvas_cells250%s.npz' %
                           args.fixture))
  print('fixture ref_canvas_cells250%s.npz: %d FoV steps, %d seeds, forward %s, '
        'minted in %.0f s' % (args.fixture, len(g['steps']), len(g['seeds']),
                              str(g['forward']), float(g['mint_wall_seconds'])))
  vol_seed = int(g['volume_seed']) if 'volume_seed' in g.files else 1234
  vol = synthetic.cells_volume((250, 250, 250), seed=vol_seed)
  print('volume: cells 250^3, seed %d' % vol_seed)
  want_steps = [tuple(int(v) for v in p) for p in g['steps']]
  want_moves, off = [], 0
  for nm in g['n_moves']:
    nm = int(nm)
    want_moves.append([(float(g['move_scores'][off + j]),
                        tuple(int(v) for v in g['move_coords'][off + j]))
                       for j in range(nm)])
    off += nm
  want_seg = g['segmentation'].astype(np.int32)
  for variant in args.variants:
    exe.engine.set_option('conv_variant', variant)
    canvas = t2._device_canvas(exe, model, synthetic.normalize(vol))
    thr = canvas.movement_policy.score_threshold
    deltas = canvas.movement_policy.deltas
    got_steps, got_moves, got_all = [], [], []
    inner = canvas.update_at

    def rec(pos):
      pred = inner(pos)
      got_steps.append(tuple(int(v) for v in pos))
      got_moves.append(sorted(
          ((s, tuple(int(v) + int(p) for v, p in zip(o, pos)))
           for s, o, _ in pred.scored_move_offsets(deltas, thr)), reverse=True))
      # every face, threshold or not
      got_all.append(sorted(
          ((s, tuple(int(v) + int(p) for v, p in zip(o, pos)))
           for s, o, _ in pred.scored_move_offsets(deltas, -1e30)), reverse=True))
      return pred

    canvas.update_at = rec
    canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                     coords=g['seeds']))
    n = min(len(got_steps), len(want_steps))
    bad_step = next((k for k in range(n) if got_steps[k] != want_steps[k]), None)
    bad_move = next((k for k in range(n)
                     if [c for _, c in got_moves[k]] != [c for _, c in want_moves[k]]),
                    None)
    max_err = 0.0
    upto = bad_move if bad_move is not None else n
    errs = np.zeros(upto)
    for k in range(upto):
      for (s, _), (w, _) in zip(got_moves[k], want_moves[k]):
        errs[k] = max(errs[k], abs(s - w))
    max_err = float(errs.max()) if upto else 0.0
    for lim in (2e-5, 1e-4, 1e-3):
      idx = np.nonzero(errs > lim)[0]
      print('  variant %d: move-score err > %g first at step %s (%d steps in all)' %
            (variant, lim, idx[0] if idx.size else None, idx.size))
    print('  variant %d: err by 100-step block (max): %s' % (
        variant, ' '.join('%.1e' % errs[b:b + 100].max()
                          for b in range(0, upto, 100))))
    print('  variant %d: engine now on conv_variant %d, range fallbacks %d' % (
        variant, exe.engine.get_option('conv_variant'),
        exe.engine.range_fallbacks))
    seg = np.asarray(canvas.segmentation)
    inter = np.sum((seg > 0) & (want_seg > 0) & (seg == want_seg))
    union = np.sum((seg > 0) | (want_seg > 0))
    fg_inter = np.sum((seg > 0) & (want_seg > 0))
    # id-agnostic: every reference object against the object of this run that
    # covers most of it (two runs that part ways number their segments apart)
    both = (seg > 0) & (want_seg > 0)
    keys, cnt = np.unique(want_seg[both].astype(np.int64) * (1 << 32) +
                          seg[both].astype(np.int64), return_counts=True)
    size_w = np.bincount(want_seg[want_seg > 0].ravel())
    size_g = np.bincount(seg[seg > 0].ravel())
    best = {}
    for kk, c in zip(keys, cnt):
      w, gid = int(kk >> 32), int(kk & 0xffffffff)
      iou = c / float(size_w[w] + size_g[gid] - c)
      if iou > best.get(w, 0.0):
        best[w] = iou
    ref_ids = np.nonzero(size_w)[0]
    matched = sum(best.get(int(w), 0.0) * size_w[w] for w in ref_ids) / max(
        float(size_w.sum()), 1.0)
    print('variant %d: objects %d (reference %d); size-weighted best-match IoU per '
          'reference object %.6f; objects matched at IoU >= 0.999: %d' %
          (variant, int(np.count_nonzero(size_g)), len(ref_ids), matched,
           sum(1 for w in ref_ids if best.get(int(w), 0.0) >= 0.999)))
    print('variant %d: %d steps (reference %d); first position mismatch at step '
          '%s; first move-list mismatch at step %s; max move-score err before '
          'it %.3g; labelled IoU %.6f, foreground IoU %.6f, voxels %d vs %d' %
          (variant, len(got_steps), len(want_steps), bad_step, bad_move,
           max_err, inter / max(union, 1), fg_inter / max(union, 1),
           int((seg > 0).sum()), int((want_seg > 0).sum())))
    if bad_move is not None:
      k = bad_move
      print('  threshold %.9f; FoV %s' % (thr, got_steps[k]))
      print('  reference moves:', [(round(s, 7), c) for s, c in want_moves[k]])
      print('  this run, all six faces:',
            [(round(s, 7), c, round(s - thr, 7)) for s, c in got_all[k]])
    canvas.close()
  exe.engine.close()


if __name__ == '__main__':
  main()
