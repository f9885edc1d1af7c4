"""bench.py --mode sharded: ONE volume cut into overlapping sub-boxes, dealt to the ranks, assembled
on the devices (BASELINE configs[2] / [3] / [4] in shape) -- the run, its rank bookkeeping and its JSON line."""
import functools
import json
import os
import sys
import time

import numpy as np

import bench as B


def run_sharded(args, rank, local_rank, world):
  """--mode sharded: BASELINE configs[3] in shape -- ONE volume, cut into
  overlapping sub-boxes by `tile_volume` (ffn/utils/bounding_box.py:250-412),
  dealt round-robin to the ranks, every rank advancing its sub-boxes
  concurrently (`Runner.run_many`, one batched engine call per round), then the
  TIMED assembly on the devices: id offsets (all_gather), owned cores into one
  int32 volume, all_reduce(MAX) over RCCL, margin histograms, union-find
  edges (all_gather), relabel.  No voxel crosses PCIe before the result is
  asked for."""
  import tempfile
  import torch
  import torch.distributed as dist
  from ffn_amd import distributed as ffn_dist
  from ffn_amd import synthetic
  from ffn_amd.inference import runner as runner_lib

  if not torch.cuda.is_available():
    raise RuntimeError('bench.py needs an MI355X: no CPU fallback exists')
  dev_index, device = B.rank_devices(args, local_rank)  # device: where collectives run
  torch.cuda.set_device(dev_index)
  if world > 1:
    B.init_group(args, dev_index)
  comm = B.Comm(rank, world, device)

  def barrier():
    comm.barrier()
    torch.cuda.synchronize()

  n = args.sharded_volume
  shape = tuple(args.sharded_volume_zyx) if args.sharded_volume_zyx else (n, n, n)
  # ONE volume for the job: rank 0 builds it, the others map it (a per-rank
  # build costs a nearest-centre query per voxel -- 10^9 at 1024^3 -- per rank)
  t_setup0 = time.perf_counter()
  shm = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
  vol_path = os.path.join(shm, 'ffn_amd_bench_cells_%s_%s.npy' % (
      'x'.join(str(v) for v in shape), os.environ.get('MASTER_PORT', str(os.getpid()))))
  vol = synthetic.shared_volume(
      lambda: B.bench_volume(shape, 4321), vol_path, rank,
      barrier if world > 1 else None)
  t_volume = time.perf_counter() - t_setup0
  # (after the volume: its nearest-centre queries use every CPU the job has)
  request = B.make_request()
  request.seed_policy = 'PolicyPeaks'
  out_dir = tempfile.mkdtemp(prefix='ffn_sharded_%d_' % rank)
  request.segmentation_output_dir = out_dir
  if B.CONFIG == 'c1':
    request.model_checkpoint_path = os.path.join(B.ROOT, 'tests', 'golden',
                                                 'fib25_weights.npz')
  else:
    request.model_checkpoint_path = os.path.join(out_dir, 'weights.npz')
    np.savez(request.model_checkpoint_path, **B.model_variables())
  run = runner_lib.Runner(device_id=dev_index)
  run.start(request, batch_size=args.sharded_batch, direct=True,
            image_volume=vol)
  eng = run.executor.engine
  if args.conv_variant is not None:
    eng.set_option('conv_variant', args.conv_variant)
  for name, value in B._engine_options(args):
    eng.set_option(name, value)
  sub = tuple(args.sharded_sub_zyx) if args.sharded_sub_zyx else (
      (args.sharded_sub,) * 3)
  ov = tuple(B.FOV)
  boxes = ffn_dist.tile_volume(shape, sub, ov, back_shift=True)
  # sub-boxes are taken by the ranks as their canvas slots free up (the cost of
  # a box is heavy-tailed); ids follow the box index, so the assembled volume
  # does not depend on the deal
  dealer = (ffn_dist.BoxDealer(boxes, rank, world, device=device)
            if args.sharded_deal == 'dynamic'
            else iter(ffn_dist.assign_round_robin(boxes, rank, world)))
  asm = ffn_dist._assembly_for(device)
  asm.job_boxes = boxes
  mine, results = [], []

  def subvolumes():
    for b in dealer:
      mine.append(b)
      results.append(None)
      yield b.corner, b.size

  def collect(index, canvas):
    results[index] = (mine[index], asm.labels(canvas.segmentation))

  # kernel-only rate of the batched step (resident FoVs, no canvas): what the
  # conv chain takes per FoV and launch at this batch -> the batched roofline
  kernel_reps = 20

  def time_stack():
    eng.forward_resident(args.sharded_batch, 3)
    eng.synchronize()
    tk = time.perf_counter()
    eng.forward_resident(args.sharded_batch, kernel_reps)
    eng.synchronize()
    return (time.perf_counter() - tk) / kernel_reps * 1e6

  stack_us_run_weights = time_stack()
  stack_us = stack_us_run_weights
  batched_ghz = B.sample_shader_clock(eng, args.sharded_batch)
  if B.CONFIG != 'c1':  # (dense_random_blob: why)
    eng.set_weights(B.dense_random_blob())
    stack_us = time_stack()
    board = B.board_under_stack(eng, args.sharded_batch)
    eng.set_weights(B.load_model().weights_blob())
  else:
    board = B.board_under_stack(eng, args.sharded_batch)
  t_setup = time.perf_counter() - t_setup0
  eng.set_option('stat_reset', 0)
  barrier()
  t0 = time.perf_counter()
  run.run_many(subvolumes(), batch_size=args.sharded_batch, save=False,
               on_done=collect, groups=args.sharded_groups,
               max_steps_per_canvas=args.sharded_max_steps or None)
  torch.cuda.synchronize()
  t_seg_local = time.perf_counter() - t0
  barrier()
  t_seg = time.perf_counter() - t0
  steps = run.counters['update_at-calls'].value
  voxels = run.counters['voxels-segmented'].value
  if args.sharded_deal == 'dynamic':
    dealer.check_complete()
  conv_variant = eng.get_option('conv_variant')  # after the run: what it used
  step_calls = eng.get_option('stat_step_calls')
  step_items = eng.get_option('stat_step_items')
  step_hist = {k: eng.get_option('stat_hist_%d' % k)
               for k in range(1, args.sharded_batch + 1)}
  kw = dict(num_boxes=len(boxes), collective=args.sharded_collective)
  # timed assembly, in two parts (after one untimed pass: allocations, code
  # objects and the RCCL communicator are set up by the first call)
  merged, _, _, _ = ffn_dist.merge_segmentations(
      results, shape, rank, world, device, assembly=asm, keep_on_device=True, **kw)
  del merged
  barrier()
  tm = time.perf_counter()
  merged, offsets, held, _ = ffn_dist.merge_segmentations(
      results, shape, rank, world, device, assembly=asm, keep_on_device=True, **kw)
  barrier()
  merge_ms = (time.perf_counter() - tm) * 1e3
  plain_ids = int(len(np.unique(asm.to_host(merged)))) - 1
  del merged
  barrier()
  tr = time.perf_counter()
  merged, offsets, edges, roots = ffn_dist.reconcile_segmentations(
      results, shape, rank, world, device, keep_on_device=True, assembly=asm, **kw)
  barrier()
  reconcile_total_ms = (time.perf_counter() - tr) * 1e3
  final_ids = int(len(np.unique(asm.to_host(merged)))) - 1
  totals = sharded_totals(comm, steps, voxels, t_seg_local, len(mine))
  merge_bytes = dict(ffn_dist.merge_collective_bytes(shape, boxes, world),
                     used=args.sharded_collective)
  if rank == 0 and world > 1:
    try:
      os.remove(vol_path)
    except OSError:
      pass
  check = None
  if world == 1 and not args.no_assembly_check and args.assembly_check_stride > 1:
    check = sampled_assembly_check(held, offsets, merged, edges, shape,
                                   args.assembly_check_stride)
  elif world == 1 and not args.no_assembly_check:
    # checker leg (untimed): the assembly against its numpy specification
    t_check = time.perf_counter()
    from oracle import labels_oracle
    host_results = [(b, np.asarray(asm.to_host(seg))) for b, seg in held]
    host_results.sort(key=lambda r: r[0].index)  # ids follow the box index
    want, want_edges, _ = labels_oracle.reconcile(
        host_results, shape, ffn_dist.MIN_OVERLAP_VOXELS,
        ffn_dist.MIN_OVERLAP_FRACTION)
    got = np.asarray(asm.to_host(merged))
    check = {'ids_expected': int(len(np.unique(want)) - 1),
             'ids_got': int(len(np.unique(got)) - 1),
             'volume_equal': bool(np.array_equal(got, want)),
             'edges_equal': bool(np.array_equal(edges, want_edges)),
             'voxels_labelled': int((got > 0).sum()),
             'what': 'device assembly + reconciliation of every sub-box against '
                     'oracle/labels_oracle.reconcile (numpy, single process) on '
                     'the same sub-box labels',
             'seconds': round(time.perf_counter() - t_check, 1)}
  run.stop_executor()
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()
  if rank != 0:
    return
  m = dict(shape=shape, boxes=boxes, sub=sub, ov=ov, t_seg=t_seg,
           conv_variant=conv_variant, step_calls=step_calls, step_items=step_items,
           step_hist=step_hist, stack_us=stack_us,
           stack_us_run_weights=stack_us_run_weights, kernel_reps=kernel_reps,
           t_setup=t_setup, t_volume=t_volume, merge_ms=merge_ms,
           reconcile_total_ms=reconcile_total_ms, plain_ids=plain_ids,
           final_ids=final_ids, edges_n=len(edges), check=check,
           driver_calls=run.last_driver.calls,
           driver_library_seconds=run.last_driver.library_seconds,
           driver_segments_ended=run.last_driver.segments_ended,
           merge_bytes=merge_bytes, batched_ghz=batched_ghz, board=board,
           clock_spread=B.SHADER_CLOCK_SPREAD.get(args.sharded_batch))
  print(json.dumps(sharded_line(args, world, totals, m)))


def sampled_assembly_check(held, offsets, merged, edges, shape, stride):
  """Checker leg for volumes too large for the whole-volume comparison (configs[4]
  at 256 x 2048 x 2048: the numpy specification would sort a billion voxel pairs):
  the id offsets of EVERY sub-box, and for every `stride`-th sub-box (by index)
  (a) the merge edges its margin contributes -- oracle/labels_oracle.margin_edges
  on the volume assembled from all cores, against the device's edges that start
  in that sub-box's id range -- and (b) the final labels of its core: its own
  labels + offset, relabelled through the oracle's union-find over the (device's)
  edge list, against the device's volume.  held: [(SubBox, device labels)] in
  this rank's order, offsets: the device's, same order."""
  from oracle import labels_oracle
  t0 = time.perf_counter()
  order = sorted(range(len(held)), key=lambda j: held[j][0].index)
  to_host = lambda a: a.cpu().numpy() if hasattr(a, 'cpu') else np.asarray(a)
  host = {held[j][0].index: to_host(held[j][1]) for j in order}
  boxes = {held[j][0].index: held[j][0] for j in order}
  n = len(order)
  maxes = [int(host[i].max()) if host[i].size else 0 for i in range(n)]
  want_off = np.concatenate([[0], np.cumsum(maxes)])[:-1]
  offsets_equal = all(int(offsets[j]) == int(want_off[held[j][0].index]) for j in order)
  # the volume as assembled from the cores (ids in the global space, no merges yet)
  plain = np.zeros(tuple(shape), np.int32)
  for i in range(n):
    b, seg = boxes[i], host[i]
    lo = [c - k for c, k in zip(b.core_lo, b.corner)]
    hi = [c - k for c, k in zip(b.core_hi, b.corner)]
    core = seg[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    plain[b.core_lo[0]:b.core_hi[0], b.core_lo[1]:b.core_hi[1],
          b.core_lo[2]:b.core_hi[2]] = np.where(core > 0, core + int(want_off[i]), 0)
  uf = labels_oracle.UnionFind()
  for x, y, _ in sorted(map(tuple, np.asarray(edges).tolist())):
    uf.union(int(x), int(y))
  roots = {x: uf.find(x) for x in list(uf.parent)}
  keys = np.array(sorted(k for k, v in roots.items() if k != v), np.uint64)
  vals = np.array([roots[int(k)] for k in keys], np.uint64)
  got = to_host(merged)
  edges = np.asarray(edges, np.int64).reshape(-1, 3)
  sampled = list(range(0, n, stride))
  edges_equal = volume_equal = True
  edges_checked = voxels_checked = 0
  for i in sampled:
    b, seg = boxes[i], host[i]
    shifted = np.where(seg > 0, seg.astype(np.int64) + int(want_off[i]), 0)
    sel = tuple(slice(c, c + k) for c, k in zip(b.corner, b.size))
    lo = [c - k for c, k in zip(b.core_lo, b.corner)]
    hi = [c - k for c, k in zip(b.core_hi, b.corner)]
    want_e = labels_oracle.margin_edges(shifted, plain[sel], lo, hi,
                                        ffn_dist_min_overlap()[0], ffn_dist_min_overlap()[1])
    mine = edges[(edges[:, 0] > want_off[i]) & (edges[:, 0] <= want_off[i] + maxes[i])]
    mine = mine[np.lexsort((mine[:, 2], mine[:, 1], mine[:, 0]))]
    edges_equal = edges_equal and bool(np.array_equal(mine, want_e))
    edges_checked += len(want_e)
    core = (slice(b.core_lo[0], b.core_hi[0]), slice(b.core_lo[1], b.core_hi[1]),
            slice(b.core_lo[2], b.core_hi[2]))
    want_core = labels_oracle.remap(plain[core], keys, vals, keep_missing=True)
    volume_equal = volume_equal and bool(np.array_equal(got[core], want_core))
    voxels_checked += int(want_core.size)
  return {
      'sampled': True, 'stride': stride, 'sub_boxes': n, 'sub_boxes_checked': len(sampled),
      'offsets_equal': bool(offsets_equal),
      'edges_equal': bool(edges_equal), 'edges_checked': int(edges_checked),
      'edges_total': int(len(edges)),
      'volume_equal': bool(volume_equal), 'voxels_checked': int(voxels_checked),
      'voxels_labelled': int((got > 0).sum()),
      'what': 'id offsets of every sub-box; for every %d-th sub-box (by index): the merge '
              'edges of its margin (oracle/labels_oracle.margin_edges on the volume '
              'assembled from all cores) against the device\'s edges from its id range, '
              'and the final labels of its core (own labels + offset, relabelled by the '
              'oracle\'s union-find over the edge list) against the device\'s volume'
              % stride,
      'seconds': round(time.perf_counter() - t0, 1)}


def ffn_dist_min_overlap():
  from ffn_amd import distributed as ffn_dist
  return ffn_dist.MIN_OVERLAP_VOXELS, ffn_dist.MIN_OVERLAP_FRACTION


def sharded_totals(comm, steps, voxels, busy_seconds, n_boxes):
  """Per-rank numbers of the sharded mode -> the job's: FoV steps and voxels add
  up, every rank's share is kept (`per_rank`: how even the dynamic deal was)."""
  rows = comm.all_gather([steps, voxels, busy_seconds, n_boxes])
  return {
      'steps': sum(r[0] for r in rows),
      'voxels': sum(r[1] for r in rows),
      'busy_max': max(r[2] for r in rows),
      'per_rank': None if comm.world == 1 else [
          {'fov_steps': int(r[0]), 'busy_seconds': round(r[2], 3),
           'sub_boxes': int(r[3])} for r in rows],
  }


def sharded_line(args, world, totals, m):
  """The one JSON line of the sharded mode from the job's totals and rank 0's
  measurements `m` (pure bookkeeping; tests/test_bench_ranks.py)."""
  steps_all, voxels_all = totals['steps'], totals['voxels']
  shape, boxes, sub, ov, t_seg = m['shape'], m['boxes'], m['sub'], m['ov'], m['t_seg']
  conv_variant, check, merge_bytes = m['conv_variant'], m['check'], m['merge_bytes']
  step_calls, step_items, step_hist = m['step_calls'], m['step_items'], m['step_hist']
  stack_us, stack_us_run_weights = m['stack_us'], m['stack_us_run_weights']
  kernel_reps, t_setup, t_volume = m['kernel_reps'], m['t_setup'], m['t_volume']
  merge_ms, reconcile_total_ms = m['merge_ms'], m['reconcile_total_ms']
  plain_ids, final_ids, edges_n = m['plain_ids'], m['final_ids'], m['edges_n']
  driver_calls, driver_library_seconds = m['driver_calls'], m['driver_library_seconds']
  driver_segments_ended = m['driver_segments_ended']
  # batched roofline: algorithmic flops of the conv launches of one stack at this
  # batch / the time of the whole resident stack (conv0_a included: it is
  # 1 / (2 depth) of the launches), against the ceiling of the arithmetic used
  fov_launch_us = stack_us / args.sharded_batch / (2 * B.DEPTH)
  # every flop of the resident stack (conv0_a's 0.124 GFLOP, the 2 depth - 1
  # convs, the fused head) over its wall time -- NOT 2 depth equal launches
  batched_tflops = args.sharded_batch * B.STEP_FLOPS / (stack_us * 1e-6) / 1e12
  batched_peak = B.PEAK_BF16_MFMA_TFLOPS / 3.0
  roofline = {
      'bound': 'mfma',
      'kernel': 'conv32m (conv_variant %s): the batched conv stack, %d FoVs per launch, one '
                'launch per conv; %s' % (
                    conv_variant, args.sharded_batch,
                    'FIB-25 weights' if B.CONFIG == 'c1' else
                    'dense seeded random weights loaded for the timing (the run\'s '
                    'constructed network is mostly zeros: less power, higher clock)'),
      'achieved': round(batched_tflops, 1),
      'peak': round(batched_peak, 1),
      'unit': 'TFLOP/s',
      'frac': round(batched_tflops / batched_peak, 4),
      'traffic': None,
      'flops_per_stack': args.sharded_batch * B.STEP_FLOPS,
      'avg_stack_us': round(stack_us, 1),
      'timing': 'wall clock over %d resident stacks of %d FoVs between device '
                'synchronisations (conv0_a + %d conv launches each, every flop of a step '
                'counted)' % (kernel_reps, args.sharded_batch, 2 * B.DEPTH - 1),
      'shader_clock_ghz': m.get('batched_ghz'),
      'shader_clock_samples': m.get('clock_spread'),
      'board': m.get('board'),
  }
  out = {
      'metric': 'FoV-steps/sec (one %s volume sharded by sub-box over %d GPU(s))'
                % ('x'.join(str(v) for v in shape), world),
      'value': round(steps_all / t_seg, 2),
      'unit': 'FoV-steps/s',
      'n_gpus': world,
      'steps': int(steps_all),
      'warmup': 0,
      'ms_per_step': round(1e3 * t_seg / max(steps_all, 1), 4),
      'higher_is_better': True,
      'scaling': 'strong',
      'vs_baseline': None,
      'dtype': 'f32 (split products on the fp16 MFMA)',
      'data': 'synthetic',
      'config': {
          'workload': ('%s: ONE synthetic cells %s (zyx) '
                       'uint8 volume, %d overlapping sub-boxes of %s (overlap = '
                       'FoV = %s), %s deal, %d canvases open per GPU in %d group(s) '
                       'of %d (= FoVs per engine call), GPU PolicyPeaks seeds, '
                       '%s; assembly on the devices'
                       % ('configs[2] / configs[3]-shaped' if B.CONFIG == 'c1' else
                          'configs[4]-shaped (depth %d, FoV zyx %s, deltas %s)'
                          % (B.DEPTH, list(B.FOV), list(B.DELTAS)),
                          'x'.join(str(v) for v in shape), len(boxes),
                          'x'.join(str(v) for v in sub),
                          'x'.join(str(v) for v in ov), args.sharded_deal,
                          args.sharded_batch * args.sharded_groups,
                          args.sharded_groups, args.sharded_batch,
                          'FIB-25 weights' if B.CONFIG == 'c1' else
                          'constructed flood-fill weights '
                          '(synthetic.flood_fill_weights)')),
          'volume': list(shape),
          'sub_boxes': len(boxes),
          'conv_variant': conv_variant,
          'engine_options': dict(B._engine_options(args)),
          'launch': {'collective_backend': args.collective_backend,
                     'ranks_share_gpus': bool(args.ranks_share_gpus)},
          'max_steps_per_canvas': args.sharded_max_steps or None,
          'parallelism': 'sub-boxes sharded over ranks; collectives only in the '
                         'final assembly (RCCL)',
      },
      'setup_seconds': {'total': round(t_setup, 2), 'volume': round(t_volume, 2),
                        'how': 'rank 0 builds the synthetic volume once, the '
                               'other ranks map it (/dev/shm); untimed'},
      'per_rank': totals['per_rank'],
      'host_loop': {
          'library_calls': driver_calls,
          'seconds_inside_library_calls': round(driver_library_seconds, 3),
          'segments_ended': driver_segments_ended,
          'note': 'rank 0, summed over the group threads: the rest of '
                  'groups x segmentation_seconds is Python between segments '
                  '(commit, seed policy, next init_seed) and canvas set-up',
      },
      'engine_calls': {
          'batched_steps': step_calls,
          'mean_fovs_per_step': round(step_items / max(step_calls, 1), 2),
          'steps_by_fovs': {str(k): v for k, v in step_hist.items() if v},
          'note': 'rank 0; what separates the end-to-end rate from the kernel '
                  'rate: steps with fewer FoVs than the batch (canvases between '
                  'segments, the tail of the job) and host turn-around',
      },
      'roofline': roofline,
      'batched_kernel': {
          'batch': args.sharded_batch,
          'us_per_stack': round(stack_us, 1),
          'weights': ('the run\'s (FIB-25)' if B.CONFIG == 'c1' else
                      'dense seeded random weights loaded for this timing; with '
                      'the run\'s mostly-zero constructed network the same stack '
                      'takes %.1f us (less power, higher clock)'
                      % stack_us_run_weights),
          'us_per_fov_launch': round(fov_launch_us, 3),
          'achieved': round(batched_tflops, 1),
          'peak': round(batched_peak, 1),
          'unit': 'TFLOP/s',
          'frac': round(batched_tflops / batched_peak, 4),
          'shader_clock_ghz': m.get('batched_ghz'),
          'frac_of_peak_at_that_clock': (
              round(batched_tflops / (batched_peak * m['batched_ghz'] / 2.4), 4)
              if m.get('batched_ghz') else None),
          'timing': 'wall clock over %d resident stacks of %d FoVs (conv0_a + '
                    '%d conv launches each): achieved = batch x %.2f GFLOP (all of '
                    'a step\'s flops) / stack time; us_per_fov_launch = stack / '
                    'batch / %d launches'
                    % (kernel_reps, args.sharded_batch, 2 * B.DEPTH - 1,
                       B.STEP_FLOPS / 1e9, 2 * B.DEPTH),
          'whole_run': {
              'what': 'every flop of the run\'s FoV steps over its end-to-end time',
              'tflops': round(steps_all / world * B.STEP_FLOPS / t_seg / 1e12, 1),
              'frac': round(steps_all / world * B.STEP_FLOPS / t_seg / 1e12 /
                            batched_peak, 4),
          },
      },
      'segmentation_seconds': round(t_seg, 3),
      'voxels_segmented_per_s': round(voxels_all / t_seg, 1),
      'merge_ms': round(merge_ms, 2),
      'reconcile_ms': round(reconcile_total_ms - merge_ms, 2),
      'assembly': {
          'merge_ms': round(merge_ms, 2),
          'merge_plus_reconcile_ms': round(reconcile_total_ms, 2),
          'how': 'all_reduce(id counts by sub-box) + cores -> one device int32 '
                 'volume + %s over RCCL; then margin pair histograms on the GPU, '
                 'all_gather(edges), union-find, table relabel in place; wall clock '
                 'between barriers, max over ranks'
                 % ('one broadcast per sub-box core from its owner'
                    if merge_bytes['used'] == 'broadcast' else
                    'all_reduce(MAX) of the zero-filled volume'),
          'collective_bytes': merge_bytes,
          'ids_before_reconcile': plain_ids,
          'ids_after_reconcile': final_ids,
          'merge_edges': int(edges_n),
          'check_vs_specification': check,
      },
  }
  return out
