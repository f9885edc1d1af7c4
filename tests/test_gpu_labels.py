"""GPU parity of the label kernels (include/ffn_labels.h) through the C-ABI:
against the reference-minted fixtures, against the oracle on random volumes,
and through size-independent properties at 250^3."""

import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SPLIT_CASES = ['plain', 'zeros_min50', 'ties', 'big_ids', 'max_uint32']
CC_CASES = ['conn1', 'conn2_min30', 'conn3_min5', 'no_zero']


@pytest.fixture(scope='module')
def gold():
  return np.load(os.path.join(GOLDEN, 'ref_labels.npz'))


@pytest.fixture(scope='module')
def ops():
  from ffn_amd import labels
  return labels.default_ops(0)


def _blocky(shape, nids, seed, zero_frac=0.2, dtype=np.uint64, block=6):
  rng = np.random.RandomState(seed)
  coarse = rng.randint(0, nids, [(s + block - 1) // block for s in shape])
  vol = np.kron(coarse, np.ones((block,) * 3, np.int64))[
      :shape[0], :shape[1], :shape[2]]
  vol = vol + 1
  vol[rng.rand(*shape) < zero_frac] = 0
  return np.ascontiguousarray(vol.astype(dtype))


def _sorted_pairs(pa, pb, cnt):
  order = np.lexsort((pa, pb))
  return pa[order], pb[order], cnt[order]


@pytest.mark.parametrize('dtype', [np.uint64, np.int32, np.uint32, np.uint8])
def test_pair_counts_and_apply_match_oracle(ops, dtype):
  from oracle import labels_oracle
  shape = (40, 52, 70)
  a = _blocky(shape, 40, 1, dtype=dtype)
  b = _blocky(shape, 25, 2, dtype=dtype, block=9)
  pa, pb, cnt, slots = ops.pair_counts(a, b)
  want = labels_oracle.pair_counts(a, b)
  got = _sorted_pairs(pa, pb, cnt)
  for g, w in zip(got, want):
    assert np.array_equal(g, w)
  assert int(cnt.sum()) == a.size
  # relabel every pair with a code of (a, b): recovers both volumes
  out = ops.apply_pair_labels(slots, pa * np.uint64(100) + pb)
  assert out.dtype.itemsize == max(a.dtype.itemsize, 4)
  assert np.array_equal(out.astype(np.uint64),
                        a.astype(np.uint64) * 100 + b.astype(np.uint64))
  # single-volume histogram
  pa1, pb1, cnt1, _ = ops.pair_counts(a)
  ids, counts = np.unique(a, return_counts=True)
  order = np.argsort(pa1)
  assert np.array_equal(pa1[order], ids.astype(np.uint64))
  assert np.array_equal(cnt1[order], counts.astype(np.uint64))
  assert not pb1.any()


def test_pair_counts_many_unique_pairs_grows_table(ops):
  """More unique pairs than the per-block LDS table and the first cap."""
  rng = np.random.RandomState(0)
  n = 3_000_000
  a = rng.randint(0, 1 << 31, n).astype(np.uint64)
  b = rng.randint(0, 1 << 31, n).astype(np.uint64)
  pa, pb, cnt, slots = ops.pair_counts(a, b)
  keys = np.unique(a | (b << np.uint64(32)))
  assert pa.size == keys.size and int(cnt.sum()) == n
  got = np.sort(pa | (pb << np.uint64(32)))
  assert np.array_equal(got, keys)
  out = ops.apply_pair_labels(slots, pb)
  assert np.array_equal(out, b)


def test_remap_matches_oracle(ops):
  from oracle import labels_oracle
  rng = np.random.RandomState(4)
  for dtype in (np.uint64, np.int32):
    vol = _blocky((33, 47, 65), 300, 5, dtype=dtype)
    keys = rng.choice(np.arange(1, 350), 120, replace=False).astype(np.uint64)
    vals = rng.randint(0, 10**6, keys.size).astype(np.uint64)
    for keep in (True, False):
      got = ops.remap(vol, keys, vals, keep_missing=keep)
      assert got.dtype == vol.dtype
      assert np.array_equal(got, labels_oracle.remap(vol, keys, vals, keep))
    assert np.array_equal(ops.remap(vol, [], [], True), vol)
  big = np.array([2**40 + 3, 7, 0, 2**40 + 3, 2**63], np.uint64)
  got = ops.remap(big, [2**40 + 3, 2**63], [5, 2**50], True)
  assert got.tolist() == [5, 7, 0, 5, 2**50]


@pytest.mark.parametrize('name', CC_CASES)
def test_connected_components_match_reference(ops, gold, name):
  seg = gold['cc_%s_in' % name]
  conn = int(gold['cc_%s_connectivity' % name])
  out, first, sizes, fz = ops.connected_components(seg, conn, stats=True)
  want = gold['cc_%s_plain' % name]
  assert np.array_equal(out, want)
  flat = want.ravel()
  assert np.array_equal(sizes, np.bincount(flat.astype(np.int64))[1:])
  assert all(flat[int(f)] == k + 1 for k, f in enumerate(first))
  zeros = np.nonzero(seg.ravel() == 0)[0]
  assert fz == (int(zeros[0]) if zeros.size else -1)
  assert np.array_equal(ops.connected_components(seg.astype(np.int32), conn),
                        want.astype(np.int32))


@pytest.mark.parametrize('conn', [1, 2, 3])
def test_connected_components_match_oracle_random(ops, conn):
  from oracle import labels_oracle
  for seed, shape in ((1, (37, 41, 67)), (2, (5, 130, 3)), (3, (1, 1, 200))):
    vol = _blocky(shape, 3, seed, zero_frac=0.35, block=3)
    out, first, sizes, fz = ops.connected_components(vol, conn, stats=True)
    want = labels_oracle.connected_components(vol, conn)
    assert np.array_equal(out, want[0])
    assert np.array_equal(first, want[1])
    assert np.array_equal(sizes, want[2])
    assert fz == want[3]


@pytest.mark.parametrize('name', SPLIT_CASES)
def test_split_by_intersection_matches_reference(gold, name):
  from ffn_amd.inference import segmentation
  a = gold['split_%s_a' % name].copy()
  segmentation.split_segmentation_by_intersection(
      a, gold['split_%s_b' % name], int(gold['split_%s_min_size' % name]))
  assert np.array_equal(a, gold['split_%s_out' % name])


@pytest.mark.parametrize('name', CC_CASES)
def test_clean_up_and_count_matches_reference(gold, name):
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


def test_clear_dust_matches_reference(gold):
  from ffn_amd.inference import segmentation
  got = segmentation.clear_dust(gold['dust_in'].copy(), 150)
  assert np.array_equal(got, gold['dust_out'])


def test_full_size_properties_250(ops):
  """BASELINE size (250^3): properties instead of the (slow) oracle."""
  from ffn_amd import synthetic
  from ffn_amd.inference import segmentation
  shape = (250, 250, 250)
  vol = synthetic.cells_volume(shape, seed=1234)
  # a label volume with structure: thresholded phantom -> components
  fg = (vol > 110).astype(np.uint64)
  cc = ops.connected_components(fg, 1)
  ms_cc, bytes_cc = ops.last_timing()
  n_comp = int(cc.max())
  assert n_comp > 10
  # (1) idempotent: components of a component labelling are itself
  assert np.array_equal(ops.connected_components(cc, 1), cc)
  # (2) 26-connectivity can only merge 6-connected components
  cc26 = ops.connected_components(fg, 3)
  pa, pb, cnt, _ = ops.pair_counts(cc, cc26)
  ms_pc, bytes_pc = ops.last_timing()
  assert int(cnt.sum()) == fg.size
  nz = pa != 0
  assert np.unique(pa[nz]).size == nz.sum()  # each 6-cc sits in one 26-cc
  assert int(cc26.max()) <= n_comp
  # (3) histogram == bincount; sizes of all components sum to the foreground
  sizes = np.bincount(cc.ravel().astype(np.int64))
  ids, _, counts, _ = ops.pair_counts(cc)
  order = np.argsort(ids)
  assert np.array_equal(counts[order], sizes[sizes > 0].astype(np.uint64))
  # (4) split-by-intersection with itself is the identity
  a = cc.copy()
  segmentation.split_segmentation_by_intersection(a, cc, 0)
  assert np.array_equal(a, cc)
  # (5) remap by a permutation and back is the identity
  perm = np.random.RandomState(0).permutation(n_comp).astype(np.uint64) + 1
  keys = np.arange(1, n_comp + 1, dtype=np.uint64)
  fwd = ops.remap(cc, keys, perm)
  ms_rm, bytes_rm = ops.last_timing()
  assert np.array_equal(ops.remap(fwd, perm, keys), cc)
  print('\\n250^3 label kernels: cc %.2f ms (%.0f GB/s)  pair_counts %.2f ms '
        '(%.0f GB/s)  remap %.2f ms (%.0f GB/s)' % (
            ms_cc, bytes_cc / ms_cc / 1e6, ms_pc, bytes_pc / ms_pc / 1e6,
            ms_rm, bytes_rm / ms_rm / 1e6))


def test_reconcile_on_gpu_matches_specification():
  from ffn_amd import distributed as ffn_dist
  from oracle import labels_oracle
  from tests import test_labels as tl
  shape = (48, 56, 64)
  truth = tl._objects_volume(shape, 5)
  boxes = ffn_dist.tile_volume(shape, (32, 36, 40), (12, 12, 12))
  results = tl._sub_results(truth, boxes)
  want, want_edges, want_roots = labels_oracle.reconcile(results, shape, 8, 0.1)
  # (a) everything in HBM: sub-box labels, assembled volume, margin histogram,
  # relabel -- torch owns the memory, libffn_hip.so's kernels the arithmetic
  merged, _, edges, roots = ffn_dist.reconcile_segmentations(
      results, shape, 0, 1, device='cuda:0', min_overlap_voxels=8,
      min_overlap_fraction=0.1)
  assert np.array_equal(merged, want)
  assert np.array_equal(edges, want_edges) and roots == want_roots
  assert tl._partition_equal(merged, truth)
  # the assembled volume can stay on the device
  dev, _, _, _ = ffn_dist.reconcile_segmentations(
      results, shape, 0, 1, device='cuda:0', min_overlap_voxels=8,
      min_overlap_fraction=0.1, keep_on_device=True)
  assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), want)
  # (b) host arrays, GPU label kernels through host buffers
  from ffn_amd import labels
  merged_b, _, edges_b, roots_b = ffn_dist.reconcile_segmentations(
      results, shape, 0, 1, min_overlap_voxels=8, min_overlap_fraction=0.1,
      ops=labels.default_ops(0))
  assert np.array_equal(merged_b, want) and np.array_equal(edges_b, want_edges)
  assert roots_b == want_roots
  # (c) the device-free path (numpy only)
  merged_c, _, edges_c, _ = ffn_dist.reconcile_segmentations(
      results, shape, 0, 1, min_overlap_voxels=8, min_overlap_fraction=0.1)
  assert np.array_equal(merged_c, want) and np.array_equal(edges_c, want_edges)


def test_device_assembly_primitives(ops):
  """place_core / margin_pairs / remap on device pointers == the numpy forms."""
  import torch
  from ffn_amd import distributed as ffn_dist
  rng = np.random.RandomState(3)
  shape = (40, 44, 52)
  boxes = ffn_dist.tile_volume(shape, (28, 30, 36), (10, 10, 12))
  host = ffn_dist._HostAssembly()
  dev = ffn_dist._DeviceAssembly('cuda:0', ops)
  out_h, out_d = host.zeros(shape), dev.zeros(shape)
  segs = []
  for k, box in enumerate(boxes):
    seg = rng.randint(-1, 5, box.size).astype(np.int32)
    segs.append(seg)
    host.place_core(out_h, box, host.labels(seg), 10 * k)
    dev.place_core(out_d, box, dev.labels(seg), 10 * k)
  assert np.array_equal(out_d.cpu().numpy(), out_h)
  for k, box in enumerate(boxes[:3]):
    ph = host.margin_pairs(box, host.labels(segs[k]), 10 * k, out_h)
    pd = dev.margin_pairs(box, dev.labels(segs[k]), 10 * k, out_d)
    th = sorted(zip(*[np.asarray(v).tolist() for v in ph]))
    td = sorted(zip(*[np.asarray(v).tolist() for v in pd]))
    assert th == td
  keys = np.array([3, 11, 24], np.uint64)
  vals = np.array([1, 1, 7], np.uint64)
  assert np.array_equal(dev.remap(out_d, keys, vals).cpu().numpy(),
                        host.remap(out_h, keys, vals))


def test_labels_abi_rejects_bad_arguments(ops):
  from ffn_amd import _lib
  lib = _lib.load()
  h = ops._h
  a = np.zeros(8, np.uint64)
  out = np.zeros(8, np.uint64)
  n = ctypes.c_size_t(0)
  buf = np.zeros(16, np.uint64)
  slot = np.zeros(16, np.uint32)
  assert lib.ffn_labels_pair_counts(h, a.ctypes.data, None, 2, 8, 16,
                                    buf.ctypes.data, buf.ctypes.data,
                                    buf.ctypes.data, slot.ctypes.data,
                                    ctypes.byref(n)) < 0
  assert b'elem_bytes' in lib.ffn_last_error()
  big = np.array([2**33, 1, 2, 3, 4, 5, 6, 7], np.uint64)
  assert lib.ffn_labels_pair_counts(h, big.ctypes.data, a.ctypes.data, 8, 8,
                                    16, buf.ctypes.data, buf.ctypes.data,
                                    buf.ctypes.data, slot.ctypes.data,
                                    ctypes.byref(n)) < 0
  assert b'remap' in lib.ffn_last_error()
  shape = (ctypes.c_int64 * 3)(2, 2, 2)
  nc = ctypes.c_uint64(0)
  assert lib.ffn_labels_connected_components(
      h, a.ctypes.data, 8, shape, 4, out.ctypes.data, ctypes.byref(nc), 0,
      None, None, None) < 0
  # apply without a resident table
  assert lib.ffn_labels_apply_pair_labels(h, 0, None, None,
                                          out.ctypes.data) < 0
  # empty inputs are fine
  e = np.zeros(0, np.uint64)
  pa, pb, cnt, slots = ops.pair_counts(e)
  assert pa.size == 0
  assert ops.remap(e, [1], [2]).size == 0
  assert ops.connected_components(np.zeros((0, 4, 4), np.uint64)).size == 0
