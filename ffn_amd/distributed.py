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


def merge_segmentations(local_results, shape_zyx, rank: int, world: int,
                        device=None):
  """Assembles one global int32 label volume from per-rank sub-box results.

  Args:
    local_results: list of (SubBox, segmentation ndarray of SubBox.size) that
      this rank produced; ids are local to each sub-box, 0 = background.
    shape_zyx: outer volume shape
    rank, world: torch.distributed rank / world size (world == 1: no
      collective at all)
    device: torch device for the collective buffers ('cuda:k' with the nccl
      backend, 'cpu' with gloo)

  Returns:
    (global int32 ndarray, list of per-sub-box id offsets of this rank)
  """
  import torch
  import torch.distributed as dist

  # 1. global id space: offsets by exclusive scan over (rank, sub-box) order
  local_max = [int(seg.max()) if seg.size else 0 for _, seg in local_results]
  my_total = int(sum(local_max))
  if world > 1:
    t = torch.tensor([my_total], dtype=torch.int64, device=device)
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
  if sum(totals) >= 2**31:
    raise OverflowError('global id space exceeds int32')

  # 2. owned cores into a zero-filled volume, then union by all_reduce(MAX)
  out = np.zeros(tuple(shape_zyx), dtype=np.int32)
  for (box, seg), off in zip(local_results, offsets):
    lo = [c - b for c, b in zip(box.core_lo, box.corner)]
    hi = [c - b for c, b in zip(box.core_hi, box.corner)]
    core = seg[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].astype(np.int32)
    core = np.where(core > 0, core + off, 0).astype(np.int32)
    out[box.core_lo[0]:box.core_hi[0], box.core_lo[1]:box.core_hi[1],
        box.core_lo[2]:box.core_hi[2]] = core
  if world > 1:
    t = torch.from_numpy(out)
    if device is not None and str(device) != 'cpu':
      t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out = t.cpu().numpy()
  return out, offsets


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
  if pa.size == 0:
    return np.zeros((0, 3), np.int64)
  cnt = cnt.astype(np.int64)
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
                            ops=None):
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
  merged, offsets = merge_segmentations(local_results, shape_zyx, rank, world,
                                        device)
  if ops is None:
    from . import labels  # pylint:disable=g-import-not-at-top
    index = getattr(device, 'index', None)
    ops = labels.default_ops(index if index is not None else 0)
  mine = []
  for (box, seg), off in zip(local_results, offsets):
    own = np.where(seg > 0, seg.astype(np.int64) + off, 0)
    sel = tuple(slice(c, c + n) for c, n in zip(box.corner, box.size))
    lo = [c - b for c, b in zip(box.core_lo, box.corner)]
    hi = [c - b for c, b in zip(box.core_hi, box.corner)]
    mine.append(margin_edges(ops, own, merged[sel], lo, hi,
                             min_overlap_voxels, min_overlap_fraction))
  edges = (np.concatenate(mine) if mine else np.zeros((0, 3), np.int64))
  if world > 1:
    edges = _all_gather_rows(edges, world, device)
  if edges.shape[0]:
    edges = edges[np.lexsort((edges[:, 2], edges[:, 1], edges[:, 0]))]
  uf = _UnionFind()
  for x, y, _ in edges:
    uf.union(int(x), int(y))
  roots = {x: uf.find(x) for x in list(uf.parent)}
  keys = np.array(sorted(k for k, v in roots.items() if k != v), np.uint64)
  if keys.size:
    vals = np.array([roots[int(k)] for k in keys], np.uint64)
    merged = ops.remap(merged, keys, vals, keep_missing=True)
  return merged, offsets, edges, roots


def segment_volume(runner, corner_zyx, size_zyx, sub_size_zyx, overlap_zyx,
                   rank: int = 0, world: int = 1, device=None,
                   batch_size=None, reconcile: bool = True,
                   min_overlap_voxels: int = MIN_OVERLAP_VOXELS,
                   min_overlap_fraction: float = MIN_OVERLAP_FRACTION,
                   save: bool = True):
  """Segments a whole bounding box on `world` GPUs (BASELINE configs C4 / C5).

  One process per GPU calls this with its rank.  The box is cut into
  overlapping sub-boxes (`tile_volume`), dealt round-robin; each rank segments
  its sub-boxes concurrently on its GPU (`Runner.run_many`: one batched engine
  call per round) with no communication; then the ranks assemble one global
  label volume (all-reduce over RCCL) and, if `reconcile`, merge objects cut by
  sub-box borders (`reconcile_segmentations`).

  Args:
    runner: a started `ffn_amd.inference.runner.Runner` (direct=True)
    corner_zyx, size_zyx: the bounding box inside the runner's image volume
    sub_size_zyx, overlap_zyx: sub-box tiling (overlap >= the model FoV)
    rank, world, device: torch.distributed coordinates (world == 1: no
      collective is issued)
    batch_size: sub-boxes advanced per engine call on one GPU

  Returns:
    (global int32 label volume of shape size_zyx -- identical on every rank --,
     dict with 'boxes', 'mine', 'offsets', 'edges', 'roots')
  """
  corner_zyx = tuple(int(c) for c in corner_zyx)
  size_zyx = tuple(int(s) for s in size_zyx)
  # full-size sub-boxes at the back edge: a clipped sliver narrower than the
  # FoV could not host a single seed
  boxes = tile_volume(size_zyx, sub_size_zyx, overlap_zyx, back_shift=True)
  mine = assign_round_robin(boxes, rank, world)
  results = [None] * len(mine)

  def collect(index, canvas):  # the canvas is closed right after this call
    seg = np.array(np.asarray(canvas.segmentation), np.int32)
    seg[seg < 0] = 0  # the -1 "excluded" markers (runner.py:452)
    results[index] = (mine[index], seg)

  runner.run_many(
      [(tuple(c + o for c, o in zip(corner_zyx, b.corner)), b.size)
       for b in mine], batch_size=batch_size, save=save, on_done=collect)
  for b, r in zip(mine, results):
    if r is None:
      raise RuntimeError('sub-box %r was skipped (output exists / masked); '
                         'assemble from the saved files instead' % (b,))
  info = {'boxes': boxes, 'mine': mine}
  if reconcile:
    merged, offsets, edges, roots = reconcile_segmentations(
        results, size_zyx, rank, world, device, min_overlap_voxels,
        min_overlap_fraction)
    info.update(offsets=offsets, edges=edges, roots=roots)
  else:
    merged, offsets = merge_segmentations(results, size_zyx, rank, world,
                                          device)
    info.update(offsets=offsets, edges=np.zeros((0, 3), np.int64), roots={})
  info['local_results'] = results
  return merged, info

