"""Label routines: the oracle restatement against the
fixtures minted by the reference's own segmentation.py, the host-side tables of
ffn_amd.inference.segmentation on an emulated device, and the distributed
union-find reconciliation (world 2, gloo)."""

import os
import socket

import numpy as np
import pytest

from ffn_amd import distributed as ffn_dist
from oracle import labels_oracle
from tests.emulated_device import EmulatedLabelOps

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SPLIT_CASES = ['plain', 'zeros_min50', 'ties', 'big_ids', 'max_uint32']
CC_CASES = ['conn1', 'conn2_min30', 'conn3_min5', 'no_zero']


@pytest.fixture(scope='module')
def gold():
  return np.load(os.path.join(GOLDEN, 'ref_labels.npz'))


@pytest.mark.parametrize('name', SPLIT_CASES)
def test_oracle_split_by_intersection_matches_reference(gold, name):
  a = gold['split_%s_a' % name].copy()
  labels_oracle.split_segmentation_by_intersection(
      a, gold['split_%s_b' % name], int(gold['split_%s_min_size' % name]))
  assert np.array_equal(a, gold['split_%s_out' % name])


@pytest.mark.parametrize('name', CC_CASES)
def test_oracle_clean_up_matches_reference(gold, name):
  seg = gold['cc_%s_in' % name].copy()
  conn = int(gold['cc_%s_connectivity' % name])
  plain, first, sizes, _ = labels_oracle.connected_components(seg, conn)
  assert np.array_equal(plain, gold['cc_%s_plain' % name])
  flat = plain.ravel()
  for k in range(first.size):  # first voxel / size tables agree with the array
    assert flat[int(first[k])] == k + 1
    assert (k == 0 or first[k] > first[k - 1])
  assert np.array_equal(sizes, np.bincount(flat)[1:])
  orig, count = labels_oracle.clean_up_and_count(
      seg, True, conn, int(gold['cc_%s_min_size' % name]))
  assert np.array_equal(seg, gold['cc_%s_out' % name])
  ids = gold['cc_%s_ids' % name]
  assert sorted(int(k) for k in orig) == [int(v) for v in ids]
  assert [int(orig[k]) for k in sorted(orig)] == [
      int(v) for v in gold['cc_%s_orig' % name]]
  assert [int(count[k]) for k in sorted(count)] == [
      int(v) for v in gold['cc_%s_count' % name]]


def test_oracle_clear_dust_matches_reference(gold):
  got = labels_oracle.clear_dust(gold['dust_in'].copy(), 150)
  assert np.array_equal(got, gold['dust_out'])


# -- host tables of the product, device emulated --------------------------------

@pytest.fixture()
def emulated_ops(monkeypatch):
  from ffn_amd.inference import segmentation
  ops = EmulatedLabelOps(seed=3)
  monkeypatch.setattr(segmentation, '_ops', lambda device_id=None: ops)
  return ops


@pytest.mark.parametrize('name', SPLIT_CASES)
def test_product_split_tables_match_reference(gold, emulated_ops, name):
  from ffn_amd.inference import segmentation
  a = gold['split_%s_a' % name].copy()
  segmentation.split_segmentation_by_intersection(
      a, gold['split_%s_b' % name], int(gold['split_%s_min_size' % name]))
  assert np.array_equal(a, gold['split_%s_out' % name])


@pytest.mark.parametrize('name', CC_CASES)
def test_product_clean_up_tables_match_reference(gold, emulated_ops, name):
  from ffn_amd.inference import segmentation
  seg = gold['cc_%s_in' % name].copy()
  orig, count = segmentation.clean_up_and_count(
      seg, True, int(gold['cc_%s_connectivity' % name]),
      int(gold['cc_%s_min_size' % name]))
  assert np.array_equal(seg, gold['cc_%s_out' % name])
  assert [int(k) for k in sorted(orig)] == [int(v) for v in
                                            gold['cc_%s_ids' % name]]
  assert [int(orig[k]) for k in sorted(orig)] == [
      int(v) for v in gold['cc_%s_orig' % name]]
  assert [int(count[k]) for k in sorted(count)] == [
      int(v) for v in gold['cc_%s_count' % name]]
  got = segmentation.split_disconnected_components(
      gold['cc_%s_in' % name], int(gold['cc_%s_connectivity' % name]))
  assert np.array_equal(got, gold['cc_%s_plain' % name])


def test_product_error_behaviour_matches_reference(emulated_ops):
  from ffn_amd.inference import segmentation
  a = np.zeros((2, 2, 2), np.uint64)
  with pytest.raises(ValueError):
    segmentation.split_segmentation_by_intersection(
        a, np.zeros((2, 2, 3), np.uint64), 0)
  with pytest.raises(TypeError):
    segmentation.split_segmentation_by_intersection(
        a.astype(np.int32), a.astype(np.int32), 0)
  empty = np.zeros((0, 3, 3), np.uint64)
  assert segmentation.clean_up_and_count(empty) == ({}, {})
  z = np.zeros((2, 2, 2), np.uint64)
  o, c = segmentation.clean_up_and_count(z)
  assert o == {0: 0} and c == {0: 8}


def test_consensus_request_and_split(emulated_ops, gold, tmp_path):
  from ffn_amd.inference import consensus
  from ffn_amd.inference import request as request_lib
  from ffn_amd.inference import storage
  req = request_lib.ConsensusRequest()
  request_lib.parse_text('''
      segmentation1 { directory: "%s" split_cc: false }
      segmentation2 { directory: "%s" split_cc: false }
      type: CONSENSUS_SPLIT
      split_min_size: 50''' % (tmp_path / 's1', tmp_path / 's2'), req)
  assert req.type == request_lib.ConsensusRequest.CONSENSUS_SPLIT
  a, b = gold['split_zeros_min50_a'], gold['split_zeros_min50_b']
  for d, seg in (('s1', a), ('s2', b)):
    path = storage.segmentation_path(str(tmp_path / d), (0, 0, 0))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    origins = {int(i): storage.OriginInfo((1, 2, 3), int(i), 0.0)
               for i in np.unique(seg) if i}
    storage.save_subvolume(seg, origins, path, request=b'', counters='{}',
                           overlaps={})
  out, origins = consensus.compute_consensus((0, 0, 0), req)
  want = gold['split_zeros_min50_out']
  assert np.array_equal(out, want) and out.dtype == np.uint8
  kept = set(int(i) for i in np.unique(want) if i) & set(
      int(i) for i in np.unique(a))
  assert set(origins) == kept
  req.type = 5
  with pytest.raises(ValueError):
    consensus.compute_consensus_for_segmentations(a.copy(), b, req)


# -- union-find reconciliation -----------------------------------------------------

def _objects_volume(shape, seed):
  """Ground-truth objects: a few fat random tubes crossing the whole volume."""
  rng = np.random.RandomState(seed)
  vol = np.zeros(shape, np.int32)
  zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shape], indexing='ij')
  for oid in range(1, 7):
    p0 = rng.rand(3) * shape
    d = rng.randn(3)
    d /= np.linalg.norm(d)
    rel = np.stack([zz - p0[0], yy - p0[1], xx - p0[2]], -1)
    t = rel @ d
    dist = np.linalg.norm(rel - t[..., None] * d, axis=-1)
    vol[(dist < 4.5) & (vol == 0)] = oid
  return vol


def _sub_results(truth, boxes):
  """Per-sub-box segmentations with LOCAL ids (as separate canvases give)."""
  out = []
  for b in boxes:
    sel = tuple(slice(c, c + n) for c, n in zip(b.corner, b.size))
    sub = truth[sel]
    ids = [i for i in np.unique(sub) if i]
    rng = np.random.RandomState(b.index)
    rng.shuffle(ids)
    local = np.zeros_like(sub)
    for k, i in enumerate(ids):
      local[sub == i] = k + 1
    out.append((b, local))
  return out


def _partition_equal(a, b):
  """Same segmentation up to a renaming of ids."""
  pa, pb, _ = labels_oracle.pair_counts(a, b)
  return (len(set(zip(pa.tolist(), pb.tolist()))) == len(set(pa.tolist())) ==
          len(set(pb.tolist())))


def test_reconcile_single_process_restores_objects():
  shape = (48, 56, 64)
  truth = _objects_volume(shape, 5)
  boxes = ffn_dist.tile_volume(shape, (32, 36, 40), (12, 12, 12))
  results = _sub_results(truth, boxes)
  merged, offsets, edges, roots = ffn_dist.reconcile_segmentations(
      results, shape, 0, 1, device='cpu', min_overlap_voxels=8,
      min_overlap_fraction=0.0, ops=EmulatedLabelOps())
  assert _partition_equal(merged, truth)
  want, want_edges, want_roots = labels_oracle.reconcile(results, shape, 8)
  assert np.array_equal(merged, want)
  assert np.array_equal(edges, want_edges) and roots == want_roots
  # without reconciliation the cut objects stay split
  plain, _ = ffn_dist.merge_segmentations(results, shape, 0, 1)
  assert not _partition_equal(plain, truth)


def test_reconcile_defaults_do_not_merge_touching_objects():
  """Two DISTINCT objects that merely touch inside the overlap zone of a cut
  (one labelling bleeds a thin sliver into the other) are not joined by the
  default criterion; one object cut in two still is."""
  shape = (40, 40, 72)
  truth = np.zeros(shape, np.int32)
  truth[8:32, 8:32, 4:30] = 1   # object A, left of the cut (x = 36)
  truth[8:32, 8:32, 30:68] = 2  # object B, touching A inside the overlap zone
  truth[2:6, 2:38, 4:68] = 3    # object C crosses the cut: must be re-joined
  boxes = ffn_dist.tile_volume(shape, (40, 40, 44), (0, 0, 16))
  assert len(boxes) == 2
  results = _sub_results(truth, boxes)
  # the left canvas over-segments: A leaks one voxel layer into B
  b0, s0 = results[0]
  leak = (truth[:, :, :44] == 2)
  leak[:, :, 31:] = False
  a_local = int(s0[20, 20, 10])
  s0 = s0.copy()
  s0[leak] = a_local
  results[0] = (b0, s0)
  merged, _, edges, _ = ffn_dist.reconcile_segmentations(
      results, shape, 0, 1, device='cpu', ops=EmulatedLabelOps())
  ids_a = set(np.unique(merged[truth == 1]).tolist()) - {0}
  ids_b = set(np.unique(merged[:, :, 40:][truth[:, :, 40:] == 2]).tolist()) - {0}
  assert ids_a.isdisjoint(ids_b), (ids_a, ids_b, edges)
  assert len(set(np.unique(merged[truth == 3]).tolist()) - {0}) == 1
  # the permissive criterion of round 1 (1 voxel, no fraction) does chain them
  loose, _, _, _ = ffn_dist.reconcile_segmentations(
      results, shape, 0, 1, device='cpu', min_overlap_voxels=1,
      min_overlap_fraction=0.0, ops=EmulatedLabelOps())
  la = set(np.unique(loose[truth == 1]).tolist()) - {0}
  lb = set(np.unique(loose[:, :, 40:][truth[:, :, 40:] == 2]).tolist()) - {0}
  assert not la.isdisjoint(lb)


def _reconcile_worker(rank, world, port, tmpdir):
  import torch.distributed as dist
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  shape = (48, 56, 64)
  truth = _objects_volume(shape, 5)
  boxes = ffn_dist.tile_volume(shape, (32, 36, 40), (12, 12, 12))
  mine = ffn_dist.assign_round_robin(boxes, rank, world)
  results = _sub_results(truth, mine)
  merged, _, edges, _ = ffn_dist.reconcile_segmentations(
      results, shape, rank, world, device='cpu', min_overlap_voxels=8,
      min_overlap_fraction=0.0, ops=EmulatedLabelOps(seed=rank))
  np.save(os.path.join(tmpdir, 'rec_%d.npy' % rank), merged)
  np.save(os.path.join(tmpdir, 'edges_%d.npy' % rank), edges)
  dist.barrier()
  dist.destroy_process_group()


def test_reconcile_world2_gloo(tmp_path):
  import torch.multiprocessing as mp
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  world = 2
  mp.spawn(_reconcile_worker, args=(world, port, str(tmp_path)), nprocs=world,
           join=True)
  rec = [np.load(tmp_path / ('rec_%d.npy' % r)) for r in range(world)]
  edges = [np.load(tmp_path / ('edges_%d.npy' % r)) for r in range(world)]
  assert np.array_equal(rec[0], rec[1])
  assert np.array_equal(edges[0], edges[1]) and len(edges[0]) > 0
  shape = (48, 56, 64)
  truth = _objects_volume(shape, 5)
  assert _partition_equal(rec[0], truth)
  # same result as the single-process specification fed in (rank, box) order
  boxes = ffn_dist.tile_volume(shape, (32, 36, 40), (12, 12, 12))
  ordered = []
  for r in range(world):
    ordered += _sub_results(truth, ffn_dist.assign_round_robin(boxes, r, world))
  want, _, _ = labels_oracle.reconcile(ordered, shape, 8)
  assert np.array_equal(rec[0], want)
