"""bench.py at N > 1 without GPUs: world_size-2 / -4 gloo processes run the
rank bookkeeping of both modes -- per-rank numbers -> the job's totals -> rank
0's JSON line -- with tests/emulated_device.py standing in for the engine
(stream mode: a complete segment_all pass per rank; sharded mode: sub-boxes
dealt dynamically, assembled with either collective)."""

import functools
import json
import os
import socket
import time

import numpy as np
import pytest


def _spawn(fn, world, tmp_path, *extra):
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(fn, args=(world, port, str(tmp_path)) + extra, nprocs=world, join=True)


def _emulated_canvas(image, blob):
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  from ffn_amd.inference import request as request_lib
  from ffn_amd.training.model import ModelInfo
  from tests.emulated_device import EmulatedDeviceClient
  r = request_lib.InferenceRequest()
  o = r.inference_options
  o.init_activation, o.pad_value, o.move_threshold = 0.95, 0.05, 0.9
  o.segment_threshold, o.min_segment_size = 0.6, 1000
  o.min_boundary_dist.x = o.min_boundary_dist.y = o.min_boundary_dist.z = 1
  info = ModelInfo(deltas=(8, 8, 8), pred_mask_size=(33, 33, 33),
                   input_seed_size=(33, 33, 33), input_image_size=(33, 33, 33))
  client = EmulatedDeviceClient(inference_utils.Counters(), blob, 12, (33, 33, 33),
                                (8, 8, 8))
  return inference.make_canvas(info, client, image, r.inference_options,
                               counters=inference_utils.Counters(),
                               movement_policy_fn=movement.get_policy_fn(r, info))


def _blob():
  from oracle import ffn_oracle
  from tests.conftest import GOLDEN
  with np.load(os.path.join(GOLDEN, 'fib25_weights.npz')) as d:
    variables = {k: d[k] for k in d.files}
  ffn_oracle.set_threads(2)
  return ffn_oracle.weights_blob(variables, 12)


def _stream_worker(rank, world, port, tmpdir):
  import torch.distributed as dist
  import bench
  from ffn_amd import synthetic
  from ffn_amd.inference import seed as seed_lib
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  comm = bench.Comm(rank, world)  # CPU tensors over gloo
  args = bench.build_parser().parse_args(['--gpus', str(world), '--steps', '3',
                                          '--warmup', '1', '--volume', '56'])
  bench.configure(args)
  # every rank its own volume, as bench.py's stream mode has it
  vol = synthetic.cells_volume((56, 56, 56), seed=11 + rank, membrane_dilate=1)
  canvas = _emulated_canvas(synthetic.normalize(vol), _blob())
  policy = functools.partial(seed_lib.PolicyGrid3d, step=16, offsets=(0, 8))
  comm.barrier()
  t0 = time.perf_counter()
  canvas.segment_all(seed_policy=policy)
  t_local = time.perf_counter() - t0
  comm.barrier()
  t_all = time.perf_counter() - t0
  steps = canvas.counters['update_at-calls'].value
  voxels = canvas.counters['voxels-segmented'].value
  full = bench.full_volume_totals(comm, steps, voxels, t_local, t_all,
                                  len(canvas.origins), bench.VOLUME_ZYX)
  # the K timed steps: rank r claims (0.010 + 0.001 r) seconds for them
  local = {
      'full_volume': full, 'elapsed': 0.010 + 0.001 * rank, 'elapsed_local': 0.01,
      'voxels': 100 * (rank + 1), 'voxels_run': voxels, 'steps_run': steps,
      'seconds_run': t_local, 'conv_ms': 0.5, 'conv_launches': 3,
      'prof_samples_ms': np.array([0.17, 0.16, 0.18], np.float32), 'flow': 2,
      'conv_variant': 9, 'profile_every': 1, 'prewarm_steps': 0,
      'volume_passes_completed': 0, 'merge_ms': 1.0, 'merged_ids': 3,
      'counters': {k: c.value for k, c in canvas.counters},
      'speculation': {}, 'segments': len(canvas.origins),
  }
  res = bench.stream_totals(comm, local)
  np.save(os.path.join(tmpdir, 'rank_%d.npy' % rank),
          np.array([steps, voxels, t_local, t_all]))
  if rank == 0:
    with open(os.path.join(tmpdir, 'line.json'), 'w') as f:
      f.write(json.dumps(bench.stream_line(args, world, res)))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_stream_mode_rank_aggregation(tmp_path, world):
  """`value` = the FoV steps of every rank's complete pass over the slowest
  rank's clock; the K timed steps are the steady state next to it (MAX of the
  ranks' times); voxels add up."""
  _spawn(_stream_worker, world, tmp_path)
  ranks = [np.load(tmp_path / ('rank_%d.npy' % r)) for r in range(world)]
  with open(tmp_path / 'line.json') as f:
    line = json.load(f)
  steps = sum(int(r[0]) for r in ranks)
  voxels = sum(int(r[1]) for r in ranks)
  assert len({int(r[0]) for r in ranks}) > 1  # the ranks really did differ
  fv = line['full_volume']
  assert line['n_gpus'] == world and line['scaling'] == 'weak'
  assert fv['steps'] == steps and fv['voxels_segmented'] == voxels
  assert fv['rank0']['steps'] == int(ranks[0][0])
  t_all = max(float(r[3]) for r in ranks)
  assert fv['seconds'] >= max(float(r[2]) for r in ranks) - 0.05  # (other ranks' clocks)
  assert abs(fv['seconds'] - t_all) < 0.25
  assert line['value'] == fv['fov_steps_per_s'] == round(steps / fv['seconds'], 1)
  assert line['voxels_segmented_per_s'] == fv['voxels_segmented_per_s']
  assert line['ms_per_step'] == round(1e3 * world / line['value'], 4)
  ss = line['steady_state']
  slowest = 0.010 + 0.001 * (world - 1)
  assert ss['value'] == round(world * 3 / slowest, 2)
  assert ss['ms_per_step'] == round(1e3 * slowest / 3, 4)
  assert line['steps'] == 3 and line['warmup'] == 1
  assert line['voxels_leg']['voxels_segmented'] == voxels
  assert line['voxels_leg']['fov_steps'] == steps
  assert 'held %d voxels' % sum(100 * (r + 1) for r in range(world)) in (
      line['voxels_leg']['region'])
  rf = line['roofline']
  assert rf['bound'] == 'mfma' and rf['launches_per_step'] == 1
  assert abs(rf['avg_launch_us'] - 170.0) < 0.01
  assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3


_SHAPE = (56, 56, 90)
_SUB = (56, 56, 56)
_OV = (33, 33, 33)


def _sharded_worker(rank, world, port, tmpdir):
  import torch.distributed as dist
  import bench
  from ffn_amd import distributed as ffn_dist
  from ffn_amd import synthetic
  from ffn_amd.inference import seed as seed_lib
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  comm = bench.Comm(rank, world)
  args = bench.build_parser().parse_args(
      ['--gpus', str(world), '--mode', 'sharded', '--sharded-batch', '4',
       '--sharded-groups', '1'])
  bench.configure(args)
  vol = synthetic.cells_volume(_SHAPE, seed=21, membrane_dilate=1)
  image = synthetic.normalize(vol)
  boxes = ffn_dist.tile_volume(_SHAPE, _SUB, _OV, back_shift=True)
  blob = _blob()
  policy = functools.partial(seed_lib.PolicyGrid3d, step=16, offsets=(0, 8))
  results, steps, voxels = [], 0, 0
  comm.barrier()
  t0 = time.perf_counter()
  for b in ffn_dist.BoxDealer(boxes, rank, world):  # the dynamic deal, over the store
    sel = tuple(slice(c, c + n) for c, n in zip(b.corner, b.size))
    canvas = _emulated_canvas(image[sel], blob)
    canvas.segment_all(seed_policy=policy)
    seg = np.array(np.asarray(canvas.segmentation), np.int32)
    seg[seg < 0] = 0
    results.append((b, seg))
    steps += canvas.counters['update_at-calls'].value
    voxels += canvas.counters['voxels-segmented'].value
  t_local = time.perf_counter() - t0
  comm.barrier()
  t_seg = time.perf_counter() - t0
  merged = {}
  for coll in ('broadcast', 'all_reduce'):
    asm = ffn_dist._assembly_for('cpu')
    asm.job_boxes = boxes
    merged[coll], _, edges, _ = ffn_dist.reconcile_segmentations(
        results, _SHAPE, rank, world, 'cpu', assembly=asm, num_boxes=len(boxes),
        collective=coll)
  totals = bench.sharded_totals(comm, steps, voxels, t_local, len(results))
  np.savez(os.path.join(tmpdir, 'rank_%d.npz' % rank), steps=steps, voxels=voxels,
           taken=np.array([b.index for b, _ in results], np.int64), **merged)
  if rank == 0:
    m = dict(shape=_SHAPE, boxes=boxes, sub=_SUB, ov=_OV, t_seg=t_seg, conv_variant=9,
             step_calls=10, step_items=30, step_hist={1: 1, 2: 2, 3: 3, 4: 4},
             stack_us=500.0, stack_us_run_weights=500.0, kernel_reps=20, t_setup=1.0,
             t_volume=0.5, merge_ms=2.0, reconcile_total_ms=5.0, plain_ids=9,
             final_ids=int(len(np.unique(merged['broadcast'])) - 1),
             edges_n=len(edges), check=None, driver_calls=7,
             driver_library_seconds=1.5, driver_segments_ended=5,
             merge_bytes=dict(ffn_dist.merge_collective_bytes(_SHAPE, boxes, world),
                              used='broadcast'))
    with open(os.path.join(tmpdir, 'line.json'), 'w') as f:
      f.write(json.dumps(bench.sharded_line(args, world, totals, m)))
  dist.barrier()
  dist.destroy_process_group()


def test_sharded_mode_rank_aggregation(tmp_path):
  """One volume, sub-boxes taken by two ranks as they come free, each segmented
  on the emulated device; both collectives assemble the same reconciled volume
  on both ranks; the JSON line carries the job's totals, every rank's share and
  the bytes either collective would move."""
  from ffn_amd import distributed as ffn_dist
  world = 2
  _spawn(_sharded_worker, world, tmp_path)
  ranks = [np.load(tmp_path / ('rank_%d.npz' % r)) for r in range(world)]
  with open(tmp_path / 'line.json') as f:
    line = json.load(f)
  boxes = ffn_dist.tile_volume(_SHAPE, _SUB, _OV, back_shift=True)
  assert sorted(int(i) for r in ranks for i in r['taken']) == list(range(len(boxes)))
  for coll in ('broadcast', 'all_reduce'):
    assert np.array_equal(ranks[0][coll], ranks[1][coll])
  assert np.array_equal(ranks[0]['broadcast'], ranks[0]['all_reduce'])
  assert (ranks[0]['broadcast'] > 0).any()
  steps = sum(int(r['steps']) for r in ranks)
  assert line['n_gpus'] == world and line['scaling'] == 'strong'
  assert line['steps'] == steps > 0
  assert line['value'] == round(steps / line['segmentation_seconds'], 2)
  assert [p['fov_steps'] for p in line['per_rank']] == [int(r['steps']) for r in ranks]
  assert [p['sub_boxes'] for p in line['per_rank']] == [len(r['taken']) for r in ranks]
  cb = line['assembly']['collective_bytes']
  vol_bytes = 4 * int(np.prod(_SHAPE))
  assert cb['used'] == 'broadcast' and cb['volume_bytes'] == vol_bytes
  assert cb['all_reduce_sent_per_gpu'] == vol_bytes          # 2 (N-1)/N V at N = 2
  assert cb['broadcast_received_per_gpu'] == vol_bytes // 2  # cores partition the volume
  assert cb['ratio'] == 4.0


def test_sampled_assembly_check_agrees_with_the_whole_volume_check():
  """bench.py --assembly-check-stride k (configs[4] at its stated size): the
  sampled checker accepts what the whole-volume specification accepts, and sees
  a wrong offset, a missing edge and a wrong voxel."""
  import bench
  from scipy import ndimage
  from ffn_amd import distributed as ffn_dist
  from oracle import labels_oracle
  shape = (40, 72, 88)
  rng = np.random.RandomState(3)
  gt, _ = ndimage.label(ndimage.gaussian_filter(rng.rand(*shape), 2.0) > 0.5)
  boxes = ffn_dist.tile_volume(shape, (40, 40, 40), (12, 12, 12), back_shift=True)
  results = []
  for b in boxes:
    sel = tuple(slice(c, c + n) for c, n in zip(b.corner, b.size))
    local = gt[sel]
    ids = np.unique(local[local > 0])
    lut = np.zeros(int(gt.max()) + 1, np.int32)
    lut[ids] = rng.permutation(len(ids)) + 1  # ids local to the sub-box
    results.append((b, lut[local]))
  asm = ffn_dist._assembly_for('cpu')
  asm.job_boxes = boxes
  merged, offsets, held, _ = ffn_dist.merge_segmentations(
      results, shape, 0, 1, 'cpu', assembly=asm, keep_on_device=True,
      num_boxes=len(boxes))
  merged, offsets, edges, _ = ffn_dist.reconcile_segmentations(
      results, shape, 0, 1, 'cpu', keep_on_device=True, assembly=asm,
      num_boxes=len(boxes))
  assert len(edges) > 3
  want, want_edges, _ = labels_oracle.reconcile(
      sorted(results, key=lambda r: r[0].index), shape, ffn_dist.MIN_OVERLAP_VOXELS,
      ffn_dist.MIN_OVERLAP_FRACTION)
  assert np.array_equal(np.asarray(merged), want) and np.array_equal(edges, want_edges)
  for stride in (1, 2, 3):
    c = bench.sampled_assembly_check(held, offsets, merged, edges, shape, stride)
    assert c['offsets_equal'] and c['edges_equal'] and c['volume_equal'], (stride, c)
    assert c['sub_boxes_checked'] == len(range(0, len(boxes), stride))
  assert c['edges_checked'] < bench.sampled_assembly_check(
      held, offsets, merged, edges, shape, 1)['edges_checked'] == len(edges)
  bad_off = list(offsets)
  bad_off[1] += 1
  assert not bench.sampled_assembly_check(held, bad_off, merged, edges, shape, 2)[
      'offsets_equal']
  first_box_edges = np.nonzero(edges[:, 0] <= int(np.asarray(held[0][1]).max()))[0]
  assert len(first_box_edges)
  assert not bench.sampled_assembly_check(
      held, offsets, merged, np.delete(edges, first_box_edges[0], axis=0), shape, 2)[
          'edges_equal']
  wrong = np.array(np.asarray(merged))
  b0 = boxes[0]
  wrong[b0.core_lo[0], b0.core_lo[1], b0.core_lo[2]] += 1
  assert not bench.sampled_assembly_check(held, offsets, wrong, edges, shape, 2)[
      'volume_equal']


def test_self_launch_command(monkeypatch):
  """`python bench.py --gpus N` outside torch.distributed.run re-executes itself
  under it: one rank per GPU on this node, rendezvous on 127.0.0.1 (the container
  hostname may not resolve), dmabuf IPC for RCCL.  (What runs behind the exec needs
  GPUs; the command itself is checked here.)"""
  import sys
  import bench
  seen = {}

  def fake_execve(path, argv, env):
    seen.update(path=path, argv=list(argv), env=dict(env))
    raise SystemExit(0)

  monkeypatch.setattr(os, 'execve', fake_execve)
  monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '20',
                                    '--warmup', '5'])
  monkeypatch.delenv('WORLD_SIZE', raising=False)
  monkeypatch.delenv('HSA_ENABLE_IPC_MODE_LEGACY', raising=False)
  with pytest.raises(SystemExit):
    bench.main()
  argv = seen['argv']
  assert seen['path'] == sys.executable and argv[:3] == [sys.executable, '-m',
                                                         'torch.distributed.run']
  assert '--nnodes=1' in argv
  assert argv[argv.index('--nproc-per-node') + 1] == '4'
  assert argv[argv.index('--master-addr') + 1] == '127.0.0.1'
  assert int(argv[argv.index('--master-port') + 1]) > 0
  assert argv[-7].endswith('bench.py') and argv[-6:] == ['--gpus', '4', '--steps', '20',
                                                         '--warmup', '5']
  assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
  # under the launcher (WORLD_SIZE set) a mismatch is an error, not another exec
  monkeypatch.setenv('WORLD_SIZE', '2')
  monkeypatch.setenv('RANK', '0')
  with pytest.raises(SystemExit, match='--gpus 4 but WORLD_SIZE=2'):
    bench.main()
