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


# ---------------------------------------------------------------------------
# dynamic dealing (world 4): skewed sub-box costs, deal-independent result
# ---------------------------------------------------------------------------
_DEAL_SHAPE = (40, 48, 112)
_DEAL_SUB = (24, 28, 32)
_DEAL_OV = (8, 8, 8)
_HEAVY = (0, 4, 8, 12)  # static round-robin would put all four on rank 0


def _deal_cost(box):
  return 0.30 if box.index in _HEAVY else 0.02


def _deal_labels(box):
  return np.random.RandomState(300 + box.index).randint(
      0, 5, box.size).astype(np.int32)


def _deal_worker(rank, world, port, tmpdir, mode):
  import time
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  boxes = ffn_dist.tile_volume(_DEAL_SHAPE, _DEAL_SUB, _DEAL_OV, back_shift=True)
  if mode == 'static':
    dealer = iter(ffn_dist.assign_round_robin(boxes, rank, world))
  elif mode == 'known':  # the dealer is told the costs: largest first
    dealer = ffn_dist.BoxDealer(boxes, rank, world, cost=_deal_cost)
  else:  # costs unknown (equal estimates): index order, taken when free
    dealer = ffn_dist.BoxDealer(boxes, rank, world)
  dist.barrier()
  t0 = time.perf_counter()
  results = []
  for b in dealer:  # one canvas slot per rank; "segmenting" = its cost in time
    time.sleep(_deal_cost(b))
    results.append((b, _deal_labels(b)))
  busy_until = time.perf_counter() - t0
  dist.barrier()
  makespan = time.perf_counter() - t0
  if mode != 'static':
    dealer.check_complete()
  out = {}
  for coll in ('all_reduce', 'broadcast'):
    asm = ffn_dist._assembly_for('cpu')
    asm.job_boxes = boxes
    merged, _ = ffn_dist.merge_segmentations(
        results, _DEAL_SHAPE, rank, world, device='cpu', assembly=asm,
        num_boxes=len(boxes), collective=coll)
    out[coll] = merged
  np.savez(os.path.join(tmpdir, '%s_%d.npz' % (mode, rank)),
           busy=busy_until, makespan=makespan,
           taken=np.array([b.index for b, _ in results]), **out)
  dist.barrier()
  dist.destroy_process_group()


def test_dynamic_dealing_world4_skewed_costs(tmp_path):
  """Four heavy sub-boxes that a static round-robin deal would all give to rank
  0: with the dynamic deal no rank idles more than 20 % of the job (costs
  known: largest first) resp. more than one heavy box (costs unknown); the
  assembled volume is the same on every rank, for either collective, and does
  not depend on who segmented what."""
  import torch.multiprocessing as mp
  world = 4
  boxes = ffn_dist.tile_volume(_DEAL_SHAPE, _DEAL_SUB, _DEAL_OV, back_shift=True)
  assert len(boxes) >= 5 * world
  stats = {}
  for mode in ('static', 'known', 'unknown'):
    with socket.socket() as s:
      s.bind(('127.0.0.1', 0))
      port = s.getsockname()[1]
    mp.spawn(_deal_worker, args=(world, port, str(tmp_path), mode), nprocs=world,
             join=True)
    stats[mode] = [np.load(tmp_path / ('%s_%d.npz' % (mode, r)))
                   for r in range(world)]
  # every box was segmented exactly once, whatever the deal
  for mode, rs in stats.items():
    taken = sorted(int(i) for r in rs for i in r['taken'])
    assert taken == list(range(len(boxes))), mode
  def idle(rs):
    span = max(float(r['makespan']) for r in rs)
    return max(1.0 - float(r['busy']) / span for r in rs), span
  idle_static, span_static = idle(stats['static'])
  idle_known, span_known = idle(stats['known'])
  idle_unknown, span_unknown = idle(stats['unknown'])
  assert idle_static > 0.5  # the deal this replaces: three ranks wait for rank 0
  assert idle_known <= 0.20, idle_known
  assert idle_unknown <= 0.30 / span_unknown + 0.10, idle_unknown
  assert span_known < 0.6 * span_static and span_unknown < 0.75 * span_static
  # one volume, whoever segmented what
  want = None
  for mode, rs in stats.items():
    for r in rs:
      for coll in ('all_reduce', 'broadcast'):
        if want is None:
          want = r[coll]
        assert np.array_equal(r[coll], want), (mode, coll)
  # = the single-process assembly with ids following the sub-box index
  single, _ = ffn_dist.merge_segmentations(
      [(b, _deal_labels(b)) for b in boxes], _DEAL_SHAPE, 0, 1,
      num_boxes=len(boxes))
  assert np.array_equal(want, single)


def _shared_volume_worker(rank, world, port, tmpdir):
  import torch.distributed as dist
  from ffn_amd import synthetic
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  builds = []

  def build():
    builds.append(rank)
    return synthetic.cells_volume((24, 32, 40), seed=77)

  vol = synthetic.shared_volume(build, os.path.join(tmpdir, 'vol.npy'), rank,
                                dist.barrier)
  np.save(os.path.join(tmpdir, 'got_%d.npy' % rank), np.asarray(vol))
  np.save(os.path.join(tmpdir, 'builds_%d.npy' % rank), np.array(builds))
  dist.barrier()
  dist.destroy_process_group()


def test_shared_volume_is_built_once_world2(tmp_path):
  """bench.py --mode sharded under torch.distributed.run: rank 0 builds the
  synthetic volume, every other rank maps the same bytes (no per-rank build:
  10^9 nearest-centre queries per rank at BASELINE configs[3]'s 1024^3)."""
  import torch.multiprocessing as mp
  from ffn_amd import synthetic
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(_shared_volume_worker, args=(2, port, str(tmp_path)), nprocs=2,
           join=True)
  want = synthetic.cells_volume((24, 32, 40), seed=77)
  for r in range(2):
    assert np.array_equal(np.load(tmp_path / ('got_%d.npy' % r)), want)
  assert np.load(tmp_path / 'builds_0.npy').tolist() == [0]
  assert np.load(tmp_path / 'builds_1.npy').tolist() == []
  # a single rank needs no file and no barrier
  assert np.array_equal(synthetic.shared_volume(
      lambda: want, str(tmp_path / 'unused.npy')), want)
  assert not (tmp_path / 'unused.npy').exists()


class _FakeStore:
  """The fetch-and-add of a torch.distributed store, in one process."""

  def __init__(self):
    self.values = {}

  def add(self, key, amount):
    self.values[key] = self.values.get(key, 0) + amount
    return self.values[key]

  def set(self, key, value):
    self.values[key] = value

  def get(self, key):
    return self.values[key]


def test_box_dealer_second_job_starts_from_zero():
  """the dealer's counter in the store used one fixed key, so a
  second job over the same process group was dealt nothing (and the assembly
  returned zeros).  ADVICE r4: the per-job key came from a process-global count
  of dealers, which dealers made by only SOME ranks shifted apart.  Every deal
  now counts under a job number rank 0 draws from the store itself."""
  import pytest
  from ffn_amd import distributed as ffn_dist
  boxes = ffn_dist.tile_volume((64, 64, 96), (40, 40, 40), (8, 8, 8))
  store = _FakeStore()
  # two ranks of a world-2 job, simulated in turn: each makes one dealer per job
  for job in range(3):
    # a dealer only ONE "rank" makes in between (a local tiling) changes nothing
    assert len(list(ffn_dist.BoxDealer(boxes[:2]))) == 2
    a = ffn_dist.BoxDealer(boxes, 0, 2, store=store)
    b = ffn_dist.BoxDealer(boxes, 1, 2, store=store)
    assert a._key == b._key
    got = []
    for k in range(len(boxes) + 2):
      try:
        got.append(next(a if k % 3 else b))
      except StopIteration:
        break
    assert sorted(x.index for x in got) == list(range(len(boxes))), job
  # an explicit job tag does the same
  a = ffn_dist.BoxDealer(boxes, 0, 2, store=store, job='again')
  assert len(list(a)) == len(boxes)
  b = ffn_dist.BoxDealer(boxes, 1, 2, store=store, job='again')
  assert len(list(b)) == 0
  # one process: the completeness check counts what was taken
  c = ffn_dist.BoxDealer(boxes)
  next(c)
  with pytest.raises(RuntimeError, match='sub-boxes were taken'):
    c.check_complete()
  list(c)
  c.check_complete()


def test_merge_reports_a_sub_box_nobody_holds():
  """with the ids following the sub-box index, a box that no rank
  segmented used to become a silent hole (or broadcast(src=-1))."""
  import pytest
  from ffn_amd import distributed as ffn_dist
  shape = (32, 32, 48)
  boxes = ffn_dist.tile_volume(shape, (32, 32, 32), (8, 8, 8))
  assert len(boxes) == 2
  seg = np.ones(boxes[0].size, np.int32)
  with pytest.raises(RuntimeError, match='no rank holds sub-box'):
    ffn_dist.merge_segmentations([(boxes[0], seg)], shape, 0, 1,
                                 num_boxes=len(boxes))
  out, _ = ffn_dist.merge_segmentations([(boxes[0], seg)], shape, 0, 1,
                                        num_boxes=len(boxes), allow_missing=True)
  assert out.max() == 1 and (out == 0).any()


# ---------------------------------------------------------------------------
# a rank's own failure reaches every rank through check_complete
# ---------------------------------------------------------------------------
def _failing_worker(rank, world, port, tmpdir):
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  boxes = ffn_dist.tile_volume(_DEAL_SHAPE, _DEAL_SUB, _DEAL_OV, back_shift=True)
  # (constructing the dealer is no collective: rank 1 makes its own a little later)
  if rank == 1:
    import time
    time.sleep(0.3)
  dealer = ffn_dist.BoxDealer(boxes, rank, world)
  taken = [b.index for b in dealer]
  msg = ''
  try:
    dealer.check_complete(failed=rank == 1)  # rank 1 "had to skip a sub-box"
  except RuntimeError as e:
    msg = str(e)
  with open(os.path.join(tmpdir, 'fail_%d.txt' % rank), 'w') as f:
    f.write('%d|%s' % (len(taken), msg))
  dist.barrier()
  dist.destroy_process_group()


def test_a_ranks_failure_is_raised_on_every_rank(tmp_path):
  """`check_complete(failed)`: ONE all-reduce carries the boxes taken and the ranks'
  own failures; a rank that failed no longer leaves its peers blocked in a collective
  it never reaches (distributed.segment_volume)."""
  import socket
  import torch.multiprocessing as mp
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
  mp.spawn(_failing_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  rows = [open(tmp_path / ('fail_%d.txt' % r)).read().split('|') for r in range(2)]
  boxes = ffn_dist.tile_volume(_DEAL_SHAPE, _DEAL_SUB, _DEAL_OV, back_shift=True)
  assert sum(int(r[0]) for r in rows) == len(boxes)  # dealt exactly once over the ranks
  for r in rows:
    assert 'report a failed or skipped sub-box' in r[1], r
