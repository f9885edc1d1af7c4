"""Multi-GPU: sub-box sharding and the final segmentation merge over RCCL.

The FoV loop has no exchange step: the reference's unit of parallelism is the
subvolume ("embarrassingly parallel ... does not contain support for workload
distribution", reference doc/manual.md:107-117), and seeds inside one canvas
are strictly sequential (inference.py:341,573-581).  So one process per GPU
(torch.distributed, backend "nccl" = RCCL over xGMI) each segments its own
overlapping sub-boxes with NO data-path collective; the only communication is
the final assembly of one global label volume:

  1. all_gather of each rank's max local id  -> exclusive-scan id offsets
     (an 8-int collective);
  2. every rank writes its sub-boxes' *core* regions (the part of each sub-box
     it owns) with globally offset ids into a zero-filled full-volume int32
     array and the ranks `all_reduce(MAX)` it -- cores are disjoint, so MAX is
     a union.  (north_star asks for exactly this all-reduce; at 1024^3 int32 =
     4.3 GB a ring all-reduce is ~49 ms on xGMI, negligible against minutes of
     segmentation.  An all_gather of owned slabs would be ~3.5 ms; see
     DESIGN.md.)

Reconciliation of objects that cross a cut is "currently *not implemented*" in
the reference (doc/manual.md:119-127, which prescribes a union-find over the
sub-box id spaces).  `reconcile_segmentations` implements it (SURVEY.md 8f rank
1) without any extra exchange of voxel data: after the all-reduce every rank
holds the assembled volume, so the labels its neighbours gave to its own
margins are already local.  Each rank pairs its sub-boxes' own labelling with
the assembled one over the margin (GPU joint histogram, include/ffn_labels.h),
turns large overlaps into merge edges, the (tiny) edge lists are all-gathered,
every rank runs the same deterministic union-find and relabels the volume with
one table-driven GPU pass.

The tiler mirrors the semantics of the reference's
`OrderlyOverlappingCalculator` (ffn/utils/bounding_box.py:250-412): sub-boxes
of a fixed size with a fixed overlap, clipped to the outer box, enumerated in
z-major order.
"""

from __future__ import annotations

import dataclasses
from typing import List, Sequence, Tuple

import numpy as np


@dataclasses.dataclass(frozen=True)
class SubBox:
  """A sub-box (zyx) and the core region it owns inside the outer volume."""
  index: int
  corner: Tuple[int, int, int]
  size: Tuple[int, int, int]
  core_lo: Tuple[int, int, int]   # absolute coordinates
  core_hi: Tuple[int, int, int]   # exclusive


def tile_volume(shape_zyx: Sequence[int], sub_size_zyx: Sequence[int],
                overlap_zyx: Sequence[int],
                back_shift: bool = False) -> List[SubBox]:
  """Overlapping sub-boxes covering `shape`; cores partition the volume.

  Consecutive sub-boxes along an axis start `sub_size - overlap` apart.  The
  last one is clipped to the volume, or, with `back_shift`, moved back so that
  it keeps the full size (`back_shift_small_sub_boxes` of the reference's
  OrderlyOverlappingCalculator, bounding_box.py:276-280, :313-318).  A trailing
  box that would be no larger than the overlap is dropped, as in the reference
  (:296-299).  The core of a sub-box extends to the middle of each overlap
  zone, so cores tile the volume exactly once.
  """
  spans_per_axis = []
  for n, s, o in zip(shape_zyx, sub_size_zyx, overlap_zyx):
    if s <= o:
      raise ValueError('sub-box size must exceed the overlap')
    step = s - o
    count = max(1, -(-(n - o) // step))  # ceil((n - overlap) / stride)
    spans = []
    for k in range(count):
      start = k * step
      end = min(start + s, n)
      if back_shift and start + s > n:
        start, end = max(n - s, 0), n
      spans.append((start, end))
    spans_per_axis.append(spans)
  boxes = []
  idx = 0
  for kz, (z0, z1) in enumerate(spans_per_axis[0]):
    for ky, (y0, y1) in enumerate(spans_per_axis[1]):
      for kx, (x0, x1) in enumerate(spans_per_axis[2]):
        lo, hi = [], []
        for a, k in enumerate((kz, ky, kx)):
          spans = spans_per_axis[a]
          lo.append(0 if k == 0 else (spans[k][0] + spans[k - 1][1]) // 2)
          hi.append(shape_zyx[a] if k == len(spans) - 1 else
                    (spans[k + 1][0] + spans[k][1]) // 2)
        boxes.append(SubBox(idx, (z0, y0, x0), (z1 - z0, y1 - y0, x1 - x0),
                            tuple(lo), tuple(hi)))
        idx += 1
  return boxes


def assign_round_robin(boxes: Sequence[SubBox], rank: int,
                       world: int) -> List[SubBox]:
  """Static dealing of sub-boxes to ranks (deterministic, no communication)."""
  return [b for b in boxes if b.index % world == rank]


class BoxDealer:
  """Dynamic dealing of sub-boxes to the ranks of a job.

  Segment cost is heavy-tailed (the reference's own sample: max 6,330 against a
  median of 8 FoV steps per segment), so a static deal leaves ranks idle
  while one finishes its expensive boxes.  Here the boxes are ordered by
  estimated cost, largest first, and every rank takes the next one whenever
  one of its canvas slots frees up -- an atomic fetch-and-add on a counter in
  the job's `torch.distributed` store (the TCP store every process group
  already has; ~0.1 ms per box, no collective, nothing on the data path).
  With more boxes than ranks x slots the ranks finish within one box of each
  other.  world == 1 (or no store): plain iteration in the same order.

  Iterating yields the SubBox objects this rank was dealt; `taken` keeps them.

  Every deal of a process group has its OWN counter in the store: `key` + '/' +
  a job number that rank 0 draws from a second counter in the store and hands
  to the others THROUGH THE STORE (no process-group collective: constructing a
  dealer blocks nobody and touches no device), so a second job over the
  same process group -- or a process that restarts against a store that is
  still alive -- starts at zero instead of at an earlier job's final count, and
  dealers that only SOME ranks construct (a local tiling with world == 1, a rank
  that takes deal='static' once) cannot shift the ranks' keys apart.  `job`: an
  explicit tag instead (any value all ranks agree on; no collective then).

  `check_complete(failed)` (collective, after the job; on `device`): every box
  was taken exactly once over all ranks and no rank reports a failure of its
  own, else RuntimeError ON EVERY RANK.
  """

  def __init__(self, boxes: Sequence[SubBox], rank: int = 0, world: int = 1,
               store=None, key: str = 'ffn_amd/next_box', cost=None, job=None,
               device=None):
    self.device = device
    if cost is None:
      cost = lambda b: int(np.prod(b.size))
    # stable: equal costs keep the tiler's z-major order
    self.order = sorted(boxes, key=lambda b: -cost(b))
    self.rank, self.world = rank, world
    self.taken: List[SubBox] = []
    self._store = store
    self._local = 0
    if world > 1 and store is None:
      import torch.distributed as dist  # pylint:disable=g-import-not-at-top
      self._store = dist.distributed_c10d._get_default_store()
    if world > 1 and job is None:
      job = self._draw_job(key)
    self._key = '%s/%s' % (key, job)

  def _draw_job(self, key: str) -> int:
    """A job number all ranks agree on: rank 0 takes the next value of the
    store's '<key>/jobs' counter, the others receive it."""
    import torch  # pylint:disable=g-import-not-at-top
    import torch.distributed as dist  # pylint:disable=g-import-not-at-top
    n = int(self._store.add(key + '/jobs', 1)) if self.rank == 0 else 0
    # rank 0 publishes the number under the count of dealers THIS rank has made
    # (every rank of a job makes its dealer, as it must to be dealt anything)
    tag = key + '/job_of_round/%d' % int(self._store.add(key + '/round/%d' % self.rank, 1))
    if self.rank == 0:
      self._store.set(tag, str(n))
      return n
    return int(self._store.get(tag))

  def _collective_device(self):
    import torch.distributed as dist  # pylint:disable=g-import-not-at-top
    if dist.get_backend() != 'nccl':
      return 'cpu'
    return self.device if self.device is not None else 'cuda'

  def check_complete(self, failed: bool = False):
    """Collective over the job's ranks: every box of the deal was taken exactly
    once (a rank that used another key would have been dealt every box), and no
    rank comes with a failure of its own (`failed`: e.g. a sub-box it had to skip)
    -- the error is raised on EVERY rank, none is left waiting in a collective."""
    if self.world == 1:
      n, bad = len(self.taken), int(bool(failed))
    else:
      import torch  # pylint:disable=g-import-not-at-top
      import torch.distributed as dist  # pylint:disable=g-import-not-at-top
      t = torch.tensor([len(self.taken), int(bool(failed))], dtype=torch.int64,
                       device=self._collective_device())
      dist.all_reduce(t)
      n, bad = int(t[0].item()), int(t[1].item())
    if bad:
      raise RuntimeError('BoxDealer: %d rank(s) of the job report a failed or skipped '
                         'sub-box (see their own message)' % bad)
    if n != len(self.order):
      raise RuntimeError('BoxDealer: %d sub-boxes were taken over all ranks, the job '
                         'has %d (did every rank deal from the same key %r?)'
                         % (n, len(self.order), self._key))

  def _next_index(self) -> int:
    if self.world > 1:
      return int(self._store.add(self._key, 1)) - 1
    self._local += 1
    return self._local - 1

  def __iter__(self):
    return self

  def __next__(self) -> SubBox:
    k = self._next_index()
    if k >= len(self.order):
      raise StopIteration()
    self.taken.append(self.order[k])
    return self.order[k]


class _HostAssembly:
  """Assembly on host arrays (numpy; collectives on CPU tensors / gloo): the
  device-free path, also the specification of `_DeviceAssembly`."""

  on_device = False

  def __init__(self, device=None, ops=None):
    self.device = device
    self._ops = ops

  def labels(self, seg):
    seg = np.asarray(seg)
    out = np.array(seg, np.int32)
    out[out < 0] = 0
    return out

  def max_id(self, seg):
    return int(seg.max()) if seg.size else 0

  def zeros(self, shape):
    return np.zeros(tuple(int(v) for v in shape), np.int32)

  def place_core(self, out, box, seg, off):
    lo = [c - b for c, b in zip(box.core_lo, box.corner)]
    hi = [c - b for c, b in zip(box.core_hi, box.corner)]
    core = seg[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    out[box.core_lo[0]:box.core_hi[0], box.core_lo[1]:box.core_hi[1],
        box.core_lo[2]:box.core_hi[2]] = np.where(core > 0, core + off, 0)

  def all_reduce_max(self, out, world):
    if world > 1:
      import torch  # pylint:disable=g-import-not-at-top
      import torch.distributed as dist  # pylint:disable=g-import-not-at-top
      t = torch.from_numpy(out)  # shares memory: reduced in place (gloo)
      if self.device is not None and str(self.device) != 'cpu':
        t = t.to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[...] = t.cpu().numpy()
      else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return out

  def broadcast_cores(self, out, boxes, owner, rank):
    """One broadcast per OWNER: the cores a rank segmented, packed into one
    contiguous buffer in sub-box order (thousands of small sub-boxes would make a
    broadcast per core latency-bound)."""
    import torch  # pylint:disable=g-import-not-at-top
    import torch.distributed as dist  # pylint:disable=g-import-not-at-top
    for src in sorted(set(int(o) for o in owner)):
      theirs = [b for b in boxes if int(owner[b.index]) == src]
      sels = [tuple(slice(l, h) for l, h in zip(b.core_lo, b.core_hi)) for b in theirs]
      sizes = [int(np.prod([h - l for l, h in zip(b.core_lo, b.core_hi)])) for b in theirs]
      buf = np.empty(sum(sizes), out.dtype)
      if src == rank:
        pos = 0
        for sel, n in zip(sels, sizes):
          buf[pos:pos + n] = out[sel].ravel()
          pos += n
      t = torch.from_numpy(buf)
      dist.broadcast(t, src=src)
      if src != rank:
        pos = 0
        for sel, n in zip(sels, sizes):
          out[sel] = buf[pos:pos + n].reshape(out[sel].shape)
          pos += n
    return out

  def margin_pairs(self, box, seg, off, merged):
    own = np.where(seg > 0, seg.astype(np.int64) + off, 0).astype(np.uint32)
    sel = tuple(slice(c, c + n) for c, n in zip(box.corner, box.size))
    g = np.array(merged[sel], np.uint32)
    core = tuple(slice(int(l - b), int(h - b))
                 for l, h, b in zip(box.core_lo, box.core_hi, box.corner))
    own[core] = 0
    g[core] = 0
    if self._ops is not None:
      pa, pb, cnt, _ = self._ops.pair_counts(own, g)
      return pa, pb, cnt
    keys, cnt = np.unique(own.astype(np.uint64).ravel() |
                          (g.astype(np.uint64).ravel() << np.uint64(32)),
                          return_counts=True)
    return (keys & np.uint64(0xffffffff), keys >> np.uint64(32),
            cnt.astype(np.uint64))

  def remap(self, merged, keys, vals):
    if self._ops is not None:
      return self._ops.remap(merged, keys, vals, keep_missing=True)
    lut = dict(zip(keys.tolist(), vals.tolist()))
    out = merged.copy()
    for k, v in lut.items():
      out[merged == k] = v
    return out

  def to_host(self, merged):
    return merged


class _DeviceAssembly:
  """Assembly on the GPU: sub-box labels, the assembled volume and the RCCL
  all-reduce buffer are ONE set of device arrays (torch owns the memory, the
  label kernels of libffn_hip.so work on the raw pointers); no voxel crosses
  PCIe until the caller asks for the result."""

  on_device = True

  def __init__(self, device, ops=None):
    import torch  # pylint:disable=g-import-not-at-top
    self.torch = torch
    self.device = torch.device(device)
    if ops is None:
      from . import labels  # pylint:disable=g-import-not-at-top
      ops = labels.default_ops(self.device.index or 0)
    self._ops = ops

  def labels(self, seg):
    """int32 device labels (>= 0) of a sub-box: from a device canvas (copied on
    the device), a device tensor, or host data (one upload)."""
    torch = self.torch
    handle = getattr(seg, 'canvas_handle', None)
    if handle is not None:  # the segmentation of a live DeviceCanvas
      out = torch.empty(tuple(seg.shape), dtype=torch.int32, device=self.device)
      self._ops.copy_canvas(handle(), out.data_ptr())
      return out
    if isinstance(seg, torch.Tensor):
      return seg.to(self.device, torch.int32).clamp_(min=0)
    host = np.array(np.asarray(seg), np.int32)
    host[host < 0] = 0
    return torch.from_numpy(host).to(self.device)

  def max_id(self, seg):
    return int(seg.max().item()) if seg.numel() else 0

  def zeros(self, shape):
    return self.torch.zeros(tuple(int(v) for v in shape), dtype=self.torch.int32,
                            device=self.device)

  def place_core(self, out, box, seg, off):
    self.torch.cuda.synchronize(self.device)
    lo = [c - b for c, b in zip(box.core_lo, box.corner)]
    hi = [c - b for c, b in zip(box.core_hi, box.corner)]
    self._ops.place_core_device(seg.data_ptr(), seg.shape, lo, hi, off,
                                out.data_ptr(), out.shape, box.corner)

  def all_reduce_max(self, out, world):
    if world > 1:
      import torch.distributed as dist  # pylint:disable=g-import-not-at-top
      dist.all_reduce(out, op=dist.ReduceOp.MAX)  # RCCL, in place
      self.torch.cuda.synchronize(self.device)
    return out

  def broadcast_cores(self, out, boxes, owner, rank):
    """One RCCL broadcast per OWNER: the cores a rank segmented, packed into one
    contiguous device buffer in sub-box order (a core is a strided box of `out`;
    one broadcast per core would be latency-bound with thousands of sub-boxes)."""
    import torch.distributed as dist  # pylint:disable=g-import-not-at-top
    torch = self.torch
    for src in sorted(set(int(o) for o in owner)):
      theirs = [b for b in boxes if int(owner[b.index]) == src]
      sels = [tuple(slice(l, h) for l, h in zip(b.core_lo, b.core_hi)) for b in theirs]
      sizes = [int(np.prod([h - l for l, h in zip(b.core_lo, b.core_hi)])) for b in theirs]
      if src == rank:
        buf = torch.cat([out[sel].reshape(-1) for sel in sels])
      else:
        buf = torch.empty(sum(sizes), dtype=out.dtype, device=out.device)
      dist.broadcast(buf, src=src)
      if src != rank:
        pos = 0
        for sel, n in zip(sels, sizes):
          out[sel] = buf[pos:pos + n].view(out[sel].shape)
          pos += n
    torch.cuda.synchronize(self.device)
    return out

  def margin_pairs(self, box, seg, off, merged):
    self.torch.cuda.synchronize(self.device)
    lo = [c - b for c, b in zip(box.core_lo, box.corner)]
    hi = [c - b for c, b in zip(box.core_hi, box.corner)]
    return self._ops.margin_pairs_device(seg.data_ptr(), seg.shape, off, lo, hi,
                                         merged.data_ptr(), merged.shape,
                                         box.corner)

  def remap(self, merged, keys, vals):
    self.torch.cuda.synchronize(self.device)
    self._ops.remap_device(merged.data_ptr(), merged.numel(), keys, vals)
    return merged

  def to_host(self, merged):
    return merged.cpu().numpy()


def _assembly_for(device, ops=None):
  if device is not None and str(device) != 'cpu' and str(device).startswith('cuda'):
    return _DeviceAssembly(device, ops)
  return _HostAssembly(device, ops)


def merge_collective_bytes(shape_zyx, boxes, world: int):
  """What the two assembly collectives move, per GPU, for an int32 label volume
  of `shape_zyx` cut into `boxes` (ring algorithms over xGMI):
    all_reduce(MAX) of the zero-filled volume: 2 (N-1)/N x 4 V bytes sent (and as
      many received) by EVERY rank, whatever it owns;
    broadcast of the owned cores: every core crosses each ring link once --
      (N-1)/N x 4 V in total over the job, i.e. each rank receives the cores it
      does not own (~ (N-1)/N x 4 V) and sends on average 1/N of that.
  -> {'volume_bytes', 'all_reduce_sent_per_gpu', 'broadcast_received_per_gpu',
      'broadcast_sent_per_gpu_mean', 'ratio'}"""
  vol = 4 * int(np.prod([int(v) for v in shape_zyx]))
  core = 4 * sum(int(np.prod([h - l for l, h in zip(b.core_lo, b.core_hi)]))
                 for b in boxes)
  n = max(int(world), 1)
  ar = 2.0 * (n - 1) / n * vol
  bc_recv = (n - 1) / n * core
  return {
      'volume_bytes': vol,
      'all_reduce_sent_per_gpu': int(ar),
      'broadcast_received_per_gpu': int(bc_recv),
      'broadcast_sent_per_gpu_mean': int(bc_recv / n) if n > 1 else 0,
      'ratio': round(ar / (bc_recv / n), 1) if n > 1 and bc_recv else None,
  }


def merge_segmentations(local_results, shape_zyx, rank: int, world: int,
                        device=None, assembly=None, keep_on_device=False,
                        num_boxes=None, collective='all_reduce',
                        allow_missing=False):
  """Assembles one global int32 label volume from per-rank sub-box results.

  Args:
    local_results: list of (SubBox, segmentation ndarray of SubBox.size) that
      this rank produced; ids are local to each sub-box, 0 = background.
    shape_zyx: outer volume shape
    rank, world: torch.distributed rank / world size (world == 1: no
      collective at all)
    device: torch device for the collective buffers ('cuda:k' with the nccl
      backend, 'cpu' with gloo).  With a cuda device the whole assembly stays
      in HBM (`_DeviceAssembly`): the sub-box labels (device canvases are
      copied on the device, host arrays uploaded once), the assembled volume
      and the all-reduce buffer; with 'cpu' / None it runs on numpy arrays
      (`_HostAssembly`, the device-free path).
    assembly: an assembly object to use instead (tests)
    keep_on_device: return (assembled volume as the assembly holds it, offsets,
      the sub-box labels as the assembly holds them, the assembly) instead
    num_boxes: total number of sub-boxes of the job.  Given, the id offsets
      follow the sub-box INDEX (exclusive scan of the per-box id counts, one
      all_reduce of num_boxes integers): the global ids then do not depend on
      which rank segmented which box, i.e. on the timing of a dynamic deal.
      None: offsets by (rank, position in local_results), the static scheme.
    collective: 'all_reduce' = every rank writes its cores into a zero-filled
      volume, one all_reduce(MAX) (what north_star names; 2 (N-1)/N volumes
      over each ring link); 'broadcast' = one broadcast per owning rank of the
      cores it segmented, packed (needs num_boxes; (N-1)/N volumes in total, and no zero-filled
      buffer is reduced) -- for volumes of 1024^3 and more.
    allow_missing: with num_boxes, a sub-box that NO rank holds is an error
      (the job lost it) unless this is set (a deliberately partial assembly:
      its core stays 0; not with collective='broadcast').

  Returns:
    (global int32 ndarray, list of per-sub-box id offsets of this rank)
  """
  import torch
  import torch.distributed as dist

  asm = assembly if assembly is not None else _assembly_for(device)
  local_results = [(box, asm.labels(seg)) for box, seg in local_results]
  coll_dev = device if asm.on_device else None
  local_max = [asm.max_id(seg) for _, seg in local_results]
  owner = None
  if num_boxes is not None:
    # 1. global id space by sub-box index
    counts = torch.zeros(2 * num_boxes, dtype=torch.int64, device=coll_dev)
    for (box, _), m in zip(local_results, local_max):
      counts[box.index] = m
      counts[num_boxes + box.index] = rank + 1
    if world > 1:
      dist.all_reduce(counts, op=dist.ReduceOp.MAX)
    counts = counts.cpu().numpy()
    owner = counts[num_boxes:] - 1
    if not allow_missing and (owner < 0).any():
      # (a dealer that dealt nothing, a rank that failed: without this the
      # volume would silently have zero holes, or broadcast(src=-1) hang)
      raise RuntimeError(
          'merge_segmentations: no rank holds sub-box(es) %s of %d' %
          (np.nonzero(owner < 0)[0].tolist()[:20], num_boxes))
    starts = np.concatenate([[0], np.cumsum(counts[:num_boxes])])
    total = int(starts[-1])
    offsets = [int(starts[box.index]) for box, _ in local_results]
  else:
    # 1. global id space: offsets by exclusive scan over (rank, sub-box) order
    my_total = int(sum(local_max))
    if world > 1:
      t = torch.tensor([my_total], dtype=torch.int64, device=coll_dev)
      gathered = [torch.zeros_like(t) for _ in range(world)]
      dist.all_gather(gathered, t)
      totals = [int(g.item()) for g in gathered]
    else:
      totals = [my_total]
    base = int(sum(totals[:rank]))
    offsets = []
    for m in local_max:
      offsets.append(base)
      base += m
    total = int(sum(totals))
  if total >= 2**31:
    raise OverflowError('global id space exceeds int32')

  # 2. owned cores into a zero-filled volume, then the union over the ranks
  out = asm.zeros(shape_zyx)
  for (box, seg), off in zip(local_results, offsets):
    asm.place_core(out, box, seg, off)
  if collective == 'broadcast' and world > 1:
    if owner is None:
      raise ValueError("collective='broadcast' needs num_boxes")
    boxes = getattr(asm, 'job_boxes', None)
    if boxes is None:
      raise ValueError("collective='broadcast' needs assembly.job_boxes "
                       '(every sub-box of the job, by index)')
    out = asm.broadcast_cores(out, boxes, owner, rank)
  elif collective not in ('all_reduce', 'broadcast'):
    raise ValueError('unknown collective %r' % (collective,))
  else:
    out = asm.all_reduce_max(out, world)
  if keep_on_device:
    return out, offsets, local_results, asm
  return asm.to_host(out), offsets


class _UnionFind:
  """Union-find over global ids; the root of a set is its smallest id, so the
  result does not depend on the order in which ranks contributed edges."""

  def __init__(self):
    self.parent = {}

  def find(self, x):
    parent = self.parent
    root = parent.setdefault(x, x)
    while root != parent[root]:
      root = parent[root]
    while parent[x] != root:
      parent[x], x = root, parent[x]
    return root

  def union(self, x, y):
    rx, ry = self.find(x), self.find(y)
    if rx != ry:
      if rx < ry:
        self.parent[ry] = rx
      else:
        self.parent[rx] = ry


#: Merge criterion defaults, conservative on purpose: an object pair is joined
#: across a cut only if the two labellings agree on at least half of the
#: smaller one's margin voxels AND on a minimum voxel count.  A single shared
#: voxel must not be an edge -- the union-find is transitive, so one spurious
#: contact would chain distinct neurites into a global merger.
MIN_OVERLAP_VOXELS = 64
MIN_OVERLAP_FRACTION = 0.5


def margin_edges(ops, seg_global_ids, assembled_box, core_lo, core_hi,
                 min_overlap_voxels=MIN_OVERLAP_VOXELS,
                 min_overlap_fraction=MIN_OVERLAP_FRACTION):
  """Merge candidates between one sub-box's own labels and the assembled
  volume over the sub-box's margin (everything outside its core).

  Args:
    ops: ffn_amd.labels.LabelOps (GPU joint histogram)
    seg_global_ids: the sub-box labelling, ids already globally offset
    assembled_box: the assembled volume restricted to the sub-box
    core_lo, core_hi: the core in sub-box coordinates

  Returns:
    int64 [k, 3] array of (own id, assembled id, shared voxels), sorted, for
    pairs of different non-zero ids with shared >= min_overlap_voxels and
    shared >= min_overlap_fraction * min(voxels of either label in the margin).
  """
  a = np.array(seg_global_ids, np.uint32)
  g = np.array(assembled_box, np.uint32)
  core = tuple(slice(int(l), int(h)) for l, h in zip(core_lo, core_hi))
  a[core] = 0
  g[core] = 0
  pa, pb, cnt, _ = ops.pair_counts(a, g)
  return _edges_from_pairs(pa, pb, cnt, min_overlap_voxels,
                           min_overlap_fraction)


def _edges_from_pairs(pa, pb, cnt, min_overlap_voxels, min_overlap_fraction):
  """(own id, assembled id, voxels) pairs of a margin -> merge edges."""
  pa = np.asarray(pa, np.uint64)
  pb = np.asarray(pb, np.uint64)
  if pa.size == 0:
    return np.zeros((0, 3), np.int64)
  cnt = np.asarray(cnt).astype(np.int64)
  ua, ia = np.unique(pa, return_inverse=True)
  ub, ib = np.unique(pb, return_inverse=True)
  size_a = np.zeros(ua.size, np.int64)
  size_b = np.zeros(ub.size, np.int64)
  np.add.at(size_a, ia, cnt)
  np.add.at(size_b, ib, cnt)
  smaller = np.minimum(size_a[ia], size_b[ib])
  keep = ((pa != 0) & (pb != 0) & (pa != pb) & (cnt >= min_overlap_voxels) &
          (cnt >= min_overlap_fraction * smaller))
  edges = np.stack([pa[keep].astype(np.int64), pb[keep].astype(np.int64),
                    cnt[keep]], axis=1)
  return edges[np.lexsort((edges[:, 2], edges[:, 1], edges[:, 0]))]


def _all_gather_rows(rows: np.ndarray, world: int, device):
  """all_gather of int64 [k, 3] arrays with per-rank k (pads to the max k)."""
  import torch
  import torch.distributed as dist
  k = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
  ks = [torch.zeros_like(k) for _ in range(world)]
  dist.all_gather(ks, k)
  ks = [int(v.item()) for v in ks]
  kmax = max(max(ks), 1)
  mine = torch.zeros((kmax, 3), dtype=torch.int64, device=device)
  if rows.shape[0]:
    mine[:rows.shape[0]] = torch.from_numpy(np.ascontiguousarray(rows)).to(
        mine.device)
  parts = [torch.zeros_like(mine) for _ in range(world)]
  dist.all_gather(parts, mine)
  return np.concatenate([p[:n].cpu().numpy() for p, n in zip(parts, ks)])


def reconcile_segmentations(local_results, shape_zyx, rank: int, world: int,
                            device=None,
                            min_overlap_voxels: int = MIN_OVERLAP_VOXELS,
                            min_overlap_fraction: float = MIN_OVERLAP_FRACTION,
                            ops=None, keep_on_device=False, assembly=None,
                            num_boxes=None, collective='all_reduce'):
  """merge_segmentations + union-find reconciliation of objects that cross a
  cut between sub-boxes (doc/manual.md:119-127).

  Args:
    local_results, shape_zyx, rank, world, device: as merge_segmentations
    min_overlap_voxels, min_overlap_fraction: merge criterion (margin_edges)
    ops: label-operations object (default: the GPU `LabelOps` of `device`)

  Returns:
    (global int32 ndarray with merged ids, id offsets of this rank's sub-boxes,
     int64 [k, 3] array of all merge edges, {id: root id} for every id that
     took part in an edge)
  """
  asm = assembly if assembly is not None else _assembly_for(device, ops)
  merged, offsets, held, asm = merge_segmentations(
      local_results, shape_zyx, rank, world, device, assembly=asm,
      keep_on_device=True, num_boxes=num_boxes, collective=collective)
  mine = []
  for (box, seg), off in zip(held, offsets):
    pa, pb, cnt = asm.margin_pairs(box, seg, off, merged)
    mine.append(_edges_from_pairs(pa, pb, cnt, min_overlap_voxels,
                                  min_overlap_fraction))
  edges = (np.concatenate(mine) if mine else np.zeros((0, 3), np.int64))
  if world > 1:
    edges = _all_gather_rows(edges, world, device if asm.on_device else None)
  if edges.shape[0]:
    edges = edges[np.lexsort((edges[:, 2], edges[:, 1], edges[:, 0]))]
  uf = _UnionFind()
  for x, y, _ in edges:
    uf.union(int(x), int(y))
  roots = {x: uf.find(x) for x in list(uf.parent)}
  keys = np.array(sorted(k for k, v in roots.items() if k != v), np.uint64)
  if keys.size:
    vals = np.array([roots[int(k)] for k in keys], np.uint64)
    merged = asm.remap(merged, keys, vals)
  if keep_on_device:
    return merged, offsets, edges, roots
  return asm.to_host(merged), offsets, edges, roots


def segment_volume(runner, corner_zyx, size_zyx, sub_size_zyx, overlap_zyx,
                   rank: int = 0, world: int = 1, device=None,
                   batch_size=None, reconcile: bool = True,
                   min_overlap_voxels: int = MIN_OVERLAP_VOXELS,
                   min_overlap_fraction: float = MIN_OVERLAP_FRACTION,
                   save: bool = True, deal: str = 'dynamic',
                   collective: str = 'broadcast', store=None, deal_job=None):
  """Segments a whole bounding box on `world` GPUs (BASELINE configs C4 / C5).

  One process per GPU calls this with its rank.  The box is cut into
  overlapping sub-boxes (`tile_volume`; make them several times as many as
  ranks x batch_size) that the ranks take one by one as their canvas slots
  free up (`BoxDealer`, largest first; deal='static': round-robin); each rank
  segments its sub-boxes concurrently on its GPU (`Runner.run_many`: one
  batched engine call per round) with no communication; then the ranks
  assemble one global label volume over RCCL (`merge_segmentations`; the ids
  follow the sub-box index, so the result does not depend on the deal) and, if
  `reconcile`, merge objects cut by sub-box borders
  (`reconcile_segmentations`).

  Args:
    runner: a started `ffn_amd.inference.runner.Runner` (direct=True)
    corner_zyx, size_zyx: the bounding box inside the runner's image volume
    sub_size_zyx, overlap_zyx: sub-box tiling (overlap >= the model FoV)
    rank, world, device: torch.distributed coordinates (world == 1: no
      collective is issued)
    batch_size: sub-boxes advanced per engine call on one GPU
    deal: 'dynamic' or 'static'
    collective: 'broadcast' (default: every sub-box core sent once by its owner)
      or 'all_reduce' (north_star's wording: a zero-filled volume reduced with
      MAX) -- same result; `merge_collective_bytes` prices both
    store: the job's torch.distributed store (default: the default group's)
    deal_job: tag of this job's counter in the store (default: a number rank 0
      draws from the store and broadcasts, see BoxDealer)

  Returns:
    (global int32 label volume of shape size_zyx -- identical on every rank --,
     dict with 'boxes', 'mine', 'offsets', 'edges', 'roots', 'seconds')
  """
  import time  # pylint:disable=g-import-not-at-top
  corner_zyx = tuple(int(c) for c in corner_zyx)
  size_zyx = tuple(int(s) for s in size_zyx)
  # full-size sub-boxes at the back edge: a clipped sliver narrower than the
  # FoV could not host a single seed
  boxes = tile_volume(size_zyx, sub_size_zyx, overlap_zyx, back_shift=True)
  if deal == 'dynamic':
    dealer = BoxDealer(boxes, rank, world, store=store, job=deal_job, device=device)
  elif deal == 'static':
    dealer = iter(assign_round_robin(boxes, rank, world))
  else:
    raise ValueError('unknown deal %r' % (deal,))
  mine, results = [], []
  asm = _assembly_for(device)
  asm.job_boxes = boxes

  def subvolumes():
    for b in dealer:
      mine.append(b)
      results.append(None)
      yield tuple(c + o for c, o in zip(corner_zyx, b.corner)), b.size

  def collect(index, canvas):  # the canvas is closed right after this call
    # the -1 "excluded" markers (runner.py:452) are dropped; on a GPU the
    # labels go from the canvas into the assembly's own device array
    results[index] = (mine[index], asm.labels(canvas.segmentation))

  t0 = time.perf_counter()
  failure = None
  try:
    runner.run_many(subvolumes(), batch_size=batch_size, save=save,
                    on_done=collect)
  except Exception as err:  # pylint:disable=broad-except
    if deal != 'dynamic' or world == 1:
      raise
    failure = err  # the peers are told below, then it is raised again
  t_seg = time.perf_counter() - t0
  skipped = [b for b, r in zip(mine, results) if r is None]
  if deal == 'dynamic':
    # (collective) every sub-box was dealt exactly once AND no rank failed: a rank
    # that raised before this point would leave its peers blocked in here
    try:
      dealer.check_complete(failed=failure is not None or bool(skipped))
    except RuntimeError:
      if failure is not None:
        raise failure
      if not skipped:
        raise
  if failure is not None:
    raise failure
  for b in skipped:
    raise RuntimeError('sub-box %r was skipped (output exists / masked); '
                       'assemble from the saved files instead' % (b,))
  info = {'boxes': boxes, 'mine': mine}
  t0 = time.perf_counter()
  if reconcile:
    merged, offsets, edges, roots = reconcile_segmentations(
        results, size_zyx, rank, world, device, min_overlap_voxels,
        min_overlap_fraction, keep_on_device=True, assembly=asm,
        num_boxes=len(boxes), collective=collective)
    info.update(offsets=offsets, edges=edges, roots=roots)
  else:
    merged, offsets, _, _ = merge_segmentations(
        results, size_zyx, rank, world, device, assembly=asm,
        keep_on_device=True, num_boxes=len(boxes), collective=collective)
    info.update(offsets=offsets, edges=np.zeros((0, 3), np.int64), roots={})
  info['assemble_seconds'] = time.perf_counter() - t0
  info['segment_seconds'] = t_seg  # this rank's own: waiting at the collective
  info['local_results'] = results  # is in assemble_seconds
  info['merged_device'] = merged if asm.on_device else None
  return asm.to_host(merged), info
