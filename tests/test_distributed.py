"""N > 1 path on CPU: world_size-2 gloo processes run the sub-box sharding and
the final segmentation merge (the only collective of the path)."""

import os
import socket

import numpy as np

from ffn_amd import distributed as ffn_dist


def test_tiler_cores_partition_the_volume():
  cases = [((100, 90, 120), (64, 64, 64), (33, 33, 33)),
           ((250, 250, 250), (140, 140, 140), (40, 40, 40)),
           ((64, 64, 64), (64, 64, 64), (33, 33, 33)),
           ((72, 80, 112), (72, 80, 72), (33, 33, 33)),
           ((256, 2048 // 8, 2048 // 8), (128, 160, 160), (21, 41, 41))]
  for shape, sub, ov in cases:
    for back_shift in (False, True):
      boxes = ffn_dist.tile_volume(shape, sub, ov, back_shift=back_shift)
      if back_shift:  # every sub-box keeps the full size
        assert all(tuple(b.size) == tuple(min(s, n) for s, n in zip(sub, shape))
                   for b in boxes)
      cover = np.zeros(shape, np.int32)
      for b in boxes:
        for a in range(3):
          assert 0 <= b.corner[a] and b.corner[a] + b.size[a] <= shape[a]
          assert b.corner[a] <= b.core_lo[a] < b.core_hi[a] <= (
              b.corner[a] + b.size[a])
        cover[b.core_lo[0]:b.core_hi[0], b.core_lo[1]:b.core_hi[1],
              b.core_lo[2]:b.core_hi[2]] += 1
      assert cover.min() == 1 and cover.max() == 1
      deal = [ffn_dist.assign_round_robin(boxes, r, 3) for r in range(3)]
      assert sorted(b.index for d in deal for b in d) == list(range(len(boxes)))
  # box count per axis = ceil((n - overlap) / stride), as the reference's
  # OrderlyOverlappingCalculator (bounding_box.py:296-305)
  assert len(ffn_dist.tile_volume((111, 40, 40), (72, 40, 40), (33, 33, 33))) == 2
  assert len(ffn_dist.tile_volume((112, 40, 40), (72, 40, 40), (33, 33, 33))) == 3
  assert len(ffn_dist.tile_volume((105, 40, 40), (72, 40, 40), (33, 33, 33))) == 2


def _worker(rank, world, port, tmpdir):
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  shape = (40, 48, 56)
  boxes = ffn_dist.tile_volume(shape, (28, 32, 40), (8, 8, 8))
  mine = ffn_dist.assign_round_robin(boxes, rank, world)
  results = []
  for b in mine:
    rng = np.random.RandomState(100 + b.index)
    results.append((b, rng.randint(0, 4, b.size).astype(np.int32)))
  merged, offsets = ffn_dist.merge_segmentations(results, shape, rank, world,
                                                 device='cpu')
  np.save(os.path.join(tmpdir, 'merged_%d.npy' % rank), merged)
  np.save(os.path.join(tmpdir, 'offsets_%d.npy' % rank), np.array(offsets))
  dist.barrier()
  dist.destroy_process_group()


def test_merge_world2_gloo(tmp_path):
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  world = 2
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  merged = [np.load(tmp_path / ('merged_%d.npy' % r)) for r in range(world)]
  assert np.array_equal(merged[0], merged[1])
  # single-process recomputation of the expected union
  shape = (40, 48, 56)
  boxes = ffn_dist.tile_volume(shape, (28, 32, 40), (8, 8, 8))
  segs = {b.index: np.random.RandomState(100 + b.index).randint(
      0, 4, b.size).astype(np.int32) for b in boxes}
  base = 0
  offsets = {}
  for r in range(world):
    for b in ffn_dist.assign_round_robin(boxes, r, world):
      offsets[b.index] = base
      base += int(segs[b.index].max())
  want = np.zeros(shape, np.int32)
  for b in boxes:
    lo = [c - k for c, k in zip(b.core_lo, b.corner)]
    hi = [c - k for c, k in zip(b.core_hi, b.corner)]
    core = segs[b.index][lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    want[b.core_lo[0]:b.core_hi[0], b.core_lo[1]:b.core_hi[1],
         b.core_lo[2]:b.core_hi[2]] = np.where(core > 0,
                                               core + offsets[b.index], 0)
  assert np.array_equal(merged[0], want)
  # ids of different sub-boxes never collide
  ids = set()
  for b in boxes:
    mine = set(np.unique(want[b.core_lo[0]:b.core_hi[0],
                              b.core_lo[1]:b.core_hi[1],
                              b.core_lo[2]:b.core_hi[2]])) - {0}
    assert not (ids & mine)
    ids |= mine


def test_merge_world1_needs_no_process_group():
  shape = (20, 20, 20)
  boxes = ffn_dist.tile_volume(shape, (20, 20, 20), (4, 4, 4))
  seg = np.random.RandomState(0).randint(0, 3, shape).astype(np.int32)
  out, offs = ffn_dist.merge_segmentations([(boxes[0], seg)], shape, 0, 1)
  assert np.array_equal(out, seg) and offs == [0]
