"""GPU parity tests: the HIP path (through the C-ABI) vs the CPU oracle.

Tolerance for float logits: |delta| <= 1e-4 absolute per FoV step (f32 path;
BASELINE.md section 4; observed ~1e-5).  Integer / index work (face argmax,
segment ids, queue trajectory, commit counts) must be bit-exact.
"""

import functools
import json
import ctypes
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

DEFAULT_VARIANT = 9
TOL = 1e-4


@pytest.fixture(scope='module')
def engine(fib25_model):
  from ffn_amd import engine as hip_engine
  eng = hip_engine.HipEngine.from_model(fib25_model, max_batch=4, device_id=0)
  yield eng
  eng.close()


def _fov_inputs(rng, n=1):
  from oracle import ffn_oracle
  img = ((rng.randint(0, 256, (n, 33, 33, 33)).astype(np.float32)) - 128) / 33
  seed = np.full((n, 33, 33, 33), ffn_oracle.f32_logit(0.05), np.float32)
  seed[:, 16, 16, 16] = ffn_oracle.f32_logit(0.95)
  seed[:, 10:20, 12:22, 8:30] += rng.normal(0, 1.5, (n, 10, 10, 22)).astype(
      np.float32)
  return img, seed


@pytest.mark.parametrize('variant,fuse_head,n', [
    (0, 1, 1), (2, 1, 1), (2, 0, 1), (6, 1, 1), (7, 1, 1), (8, 1, 1), (9, 1, 1),
    (9, 1, 2)])
def test_predict_matches_oracle(engine, fib25_blob, variant, fuse_head, n):
  """conv_variant 0 simple f32 MFMA (any FoV), 2 compact chunks (exact f32 MFMA,
  the oracle's summation order), 6 the split-product conv on 32x32x16 MFMAs
  with the taps split over the waves, producer-split planes staged by LDS-DMA
  (conv32d), 7 = 6 with 96-voxel chunks for two workgroups per CU, 8 = the
  M-split form with the weights in an LDS ring (conv32m), 9 (the default) =
  conv32mt for a step of ONE FoV and conv32m for a step of several; variant 2
  with the 1x1x1 head fused into the last conv or as its own launch."""
  from oracle import ffn_oracle
  engine.set_option('conv_variant', variant)
  engine.set_option('fuse_head', fuse_head)
  rng = np.random.RandomState(42)
  img, seed = _fov_inputs(rng, n)
  got = engine.predict(seed, img)
  want = ffn_oracle.forward(img, seed, fib25_blob, 12)
  assert got.shape == want.shape
  err = np.abs(got - want).max()
  print('variant %d fuse_head %d n %d: max |err| %.3g' % (
      variant, fuse_head, n, err))
  assert err <= TOL
  engine.restore_default_variant()
  engine.set_option('fuse_head', 1)


def test_predict_batch_and_ragged(engine, fib25_blob):
  from oracle import ffn_oracle
  rng = np.random.RandomState(7)
  img, seed = _fov_inputs(rng, 4)
  want = ffn_oracle.forward(img, seed, fib25_blob, 12)
  got4 = engine.predict(seed, img)
  assert np.abs(got4 - want).max() <= TOL
  # a ragged batch (3 of 4 slots) gives the same rows, bit for bit
  got3 = engine.predict(seed[:3], img[:3])
  assert np.array_equal(got3, got4[:3])
  # the default kernel choice runs ONE FoV as conv32mt (K-split tail: the sums
  # of the FoV's last voxels in another order), several as conv32m
  assert engine.get_option('conv_variant') == 9
  got1 = engine.predict(seed[2:3], img[2:3])
  assert np.abs(got1[0] - got4[2]).max() <= 2e-5
  assert np.abs(got1 - want[2:3]).max() <= TOL
  # one arithmetic whatever the batch: tail_batched (conv32mt for every step)
  # or conv_variant 8 (conv32m for every step)
  engine.set_option('tail_batched', 1)
  assert np.array_equal(engine.predict(seed, img)[2], got1[0])
  assert np.array_equal(engine.predict(seed[1:3], img[1:3])[1], got1[0])
  engine.set_option('tail_batched', 0)
  engine.set_option('conv_variant', 8)
  assert np.array_equal(engine.predict(seed[2:3], img[2:3])[0], got4[2])
  engine.restore_default_variant()


def test_predict_is_deterministic_and_variants_agree(engine):
  rng = np.random.RandomState(3)
  img, seed = _fov_inputs(rng, 2)
  a = engine.predict(seed, img)
  b = engine.predict(seed, img)
  assert np.array_equal(a, b)
  by_variant = {}
  for variant in (0, 2, 6, 7, 8, 9):
    engine.set_option('conv_variant', variant)
    c = engine.predict(seed, img)
    assert np.abs(a - c).max() <= 2e-5, variant
    by_variant[variant] = c
  # variant 9 runs several FoVs as conv32m ...
  assert np.array_equal(by_variant[9], by_variant[8])
  # ... and ONE as conv32mt: conv32m's arithmetic for the first 256 chunks of 128
  # voxels, conv32d's for the tail, in 32-voxel workgroups; with tail_batched a
  # batch splits the same tail off in 96-voxel workgroups -- the same bits
  engine.set_option('conv_variant', 9)
  engine.set_option('tail_batched', 1)
  both = engine.predict(seed, img)
  engine.set_option('tail_batched', 0)
  assert np.abs(a - both).max() <= 2e-5
  assert not np.array_equal(both, by_variant[8])
  for k in range(2):
    one = engine.predict(seed[k:k + 1], img[k:k + 1])
    assert np.array_equal(one[0], both[k])
  # conv32d's arithmetic and summation order do not depend on the chunk size
  assert np.array_equal(by_variant[6], by_variant[7])
  engine.restore_default_variant()


@pytest.mark.parametrize('fov_xyz,deltas_xyz,depth', [
    ([33, 33, 33], [8, 8, 8], 2), ([33, 33, 33], [8, 8, 8], 3),
    ([33, 33, 33], [8, 8, 8], 12), ([41, 41, 21], [10, 10, 5], 5),
    ([35, 33, 31], [8, 8, 7], 3), ([49, 49, 25], [12, 12, 6], 3)])
def test_resident_stack_same_bits_as_per_layer_launches(fov_xyz, deltas_xyz, depth):
  """Engine option flow (ffn_conv_resident.h, conv32ps): a single-FoV step of conv32mt
  runs its 2 depth - 1 convs as ONE resident launch whose workgroups hand rows
  to each other through per-producer sequence words (flow 2, the default), as
  one launch per conv with the same hand-off compiled in (1), or as plain
  dependent launches (0).  Every conv's arithmetic is the same instruction
  sequence: the logits must be equal bit for bit, run after run, whatever the
  workgroups' relative timing; no poll may ever give up."""
  from ffn_amd import _lib
  from ffn_amd import engine as hip_engine
  from ffn_amd.training.models import convstack_3d
  from oracle import ffn_oracle
  m = convstack_3d.ConvStack3DFFNModel(fov_size=fov_xyz, deltas=deltas_xyz,
                                       depth=depth)
  m.set_variables(ffn_oracle.random_weights(depth, seed=40 + depth, stddev=0.05))
  eng = hip_engine.HipEngine.from_model(m, max_batch=1)
  zyx = fov_xyz[::-1]
  if eng.get_option('conv_variant') != 9:
    # conv32mt does not take this FoV (<= 256 chunks): flow must refuse
    with pytest.raises(_lib.FFNHipError):
      eng.set_option('flow', 2)
    assert eng.get_option('flow') == 0
    eng.close()
    return
  assert eng.get_option('flow') == 2
  rng = np.random.RandomState(7)
  for trial in range(4):
    img = rng.normal(0, 1, [1] + zyx).astype(np.float32)
    seed = rng.normal(0, 1.5, [1] + zyx).astype(np.float32)
    got = {}
    for flow in (0, 2, 1, 2, 2):
      eng.set_option('flow', flow)
      out = eng.predict(seed, img)
      if flow in got:
        assert np.array_equal(out, got[flow]), (trial, flow)
      got[flow] = out
    assert np.array_equal(got[0], got[2]), trial
    assert np.array_equal(got[0], got[1]), trial
  # many back-to-back resident stacks (no host round trip between them)
  eng.set_option('flow', 2)
  eng.forward_resident(1, 200)
  eng.synchronize()
  assert np.array_equal(eng.predict(seed, img), got[0])
  assert eng.get_option('stat_flow_timeouts') == 0
  eng.close()


@pytest.mark.parametrize('fov_xyz,deltas_xyz', [([25, 25, 25], [6, 6, 6]),
                                                ([29, 21, 17], [7, 5, 4]),
                                                ([49, 49, 25], [12, 12, 6]),
                                                ([35, 33, 31], [8, 8, 7])])
def test_other_fov_sizes_every_supported_variant(fov_xyz, deltas_xyz):
  """Geometry generality: chunking, staging extents, magic divisions and the
  per-FoV gating of each kernel (a variant that does not fit a FoV must refuse,
  not miscompute) on FoVs other than 33^3 / 21x41x41; random weights, depth 3."""
  from ffn_amd import _lib
  from ffn_amd import engine as hip_engine
  from ffn_amd.training.models import convstack_3d
  from oracle import ffn_oracle
  variables = ffn_oracle.random_weights(3, seed=21, stddev=0.07)
  m = convstack_3d.ConvStack3DFFNModel(fov_size=fov_xyz, deltas=deltas_xyz,
                                       depth=3)
  m.set_variables(variables)
  eng = hip_engine.HipEngine.from_model(m, max_batch=3)
  zyx = fov_xyz[::-1]
  rng = np.random.RandomState(9)
  img = rng.normal(0, 1, [3] + zyx).astype(np.float32)
  seed = rng.normal(0, 2, [3] + zyx).astype(np.float32)
  want = ffn_oracle.forward(img, seed, ffn_oracle.weights_blob(variables, 3), 3)
  ran = []
  default = eng.get_option('conv_variant')
  for variant in (default, 0, 2, 6, 7, 8, 9):
    try:
      eng.set_option('conv_variant', variant)
    except _lib.FFNHipError:
      continue  # this kernel does not take this FoV
    for n in (1, 3):
      got = eng.predict(seed[:n], img[:n])
      assert np.abs(got - want[:n]).max() <= TOL, (variant, n)
    ran.append(variant)
  print('fov %s: default %d, variants run %s' % (fov_xyz, default, ran))
  assert 0 in ran and len(ran) >= 3
  eng.close()


def test_c5_model_full_depth(fib25_model):
  """BASELINE configs[4] model at its FULL depth: 18 residual modules, FoV zyx
  (21, 41, 41), deltas (5, 10, 10), random weights -- every conv variant that
  supports the geometry against the oracle."""
  from ffn_amd import engine as hip_engine
  from ffn_amd.training.models import convstack_3d
  from oracle import ffn_oracle
  variables = ffn_oracle.random_weights(18, seed=18, stddev=0.03)
  m = convstack_3d.ConvStack3DFFNModel(fov_size=[41, 41, 21],
                                       deltas=[10, 10, 5], depth=18)
  m.set_variables(variables)
  eng = hip_engine.HipEngine.from_model(m, max_batch=2)
  rng = np.random.RandomState(4)
  img = rng.normal(0, 1, (2, 21, 41, 41)).astype(np.float32)
  seed = rng.normal(0, 2, (2, 21, 41, 41)).astype(np.float32)
  blob = ffn_oracle.weights_blob(variables, 18)
  want = ffn_oracle.forward(img, seed, blob, 18)
  default = eng.get_option('conv_variant')
  assert default == 9 and eng.get_option('flow') == 2  # permuted layout, resident stack
  for variant in sorted({2, 6, default}):
    eng.set_option('conv_variant', variant)
    got = eng.predict(seed, img)
    err = np.abs(got - want).max()
    print('c5 depth 18 variant %d: max |err| %.3g' % (variant, err))
    assert err <= TOL, (variant, err)
    # n = 1: the default then runs conv32mt (its K-split tail; as ONE resident
    # launch of 35 convs under flow 2, as 35 launches under flow 0) on the
    # permuted layout, n = 2 above ran conv32m
    flows = (2, 1, 0) if variant == default else (eng.get_option('flow'),)
    ones = []
    for flow in flows:
      eng.set_option('flow', flow)
      for k in range(2):
        one = eng.predict(seed[k:k + 1], img[k:k + 1])
        err1 = np.abs(one[0] - want[k]).max()
        assert err1 <= TOL, (variant, flow, k, err1)
        ones.append(one)
    if variant == default:
      assert np.array_equal(ones[0], ones[2]) and np.array_equal(ones[0], ones[4])
      assert np.array_equal(ones[1], ones[3]) and np.array_equal(ones[1], ones[5])
      assert eng.get_option('stat_flow_timeouts') == 0
    eng.set_option('flow', 2)
  eng.close()


def test_predict_nan_seed_propagates_like_reference(engine):
  # The stateless contract feeds the seed as given (the Canvas substitutes NaN
  # before calling predict); a NaN must not be silently replaced.
  rng = np.random.RandomState(5)
  img, seed = _fov_inputs(rng, 1)
  seed[0, 0, 0, 0] = np.nan
  out = engine.predict(seed, img)
  assert np.isnan(out[0, 0, 0, 0])


def test_layerwise_random_weights_other_depth(fib25_model):
  """depth 3 / random weights: exercises conv0_b and one residual module."""
  from ffn_amd import engine as hip_engine
  from ffn_amd.training.models import convstack_3d
  from oracle import ffn_oracle
  variables = ffn_oracle.random_weights(3, seed=11, stddev=0.08)
  m = convstack_3d.ConvStack3DFFNModel(fov_size=[33, 33, 33],
                                       deltas=[8, 8, 8], depth=3)
  m.set_variables(variables)
  eng = hip_engine.HipEngine.from_model(m, max_batch=1)
  rng = np.random.RandomState(1)
  img, seed = _fov_inputs(rng, 1)
  got = eng.predict(seed, img)
  want = ffn_oracle.forward(img, seed, ffn_oracle.weights_blob(variables, 3), 3)
  assert np.abs(got - want).max() <= TOL
  eng.close()


def test_anisotropic_fov(fib25_model):
  """C5 geometry: fov zyx (21, 41, 41), deltas (5, 10, 10), random weights."""
  from ffn_amd import engine as hip_engine
  from ffn_amd.training.models import convstack_3d
  from oracle import ffn_oracle
  variables = ffn_oracle.random_weights(2, seed=5, stddev=0.08)
  m = convstack_3d.ConvStack3DFFNModel(fov_size=[41, 41, 21],
                                       deltas=[10, 10, 5], depth=2)
  m.set_variables(variables)
  eng = hip_engine.HipEngine.from_model(m, max_batch=1)
  rng = np.random.RandomState(2)
  img = rng.normal(0, 1, (1, 21, 41, 41)).astype(np.float32)
  seed = rng.normal(0, 1, (1, 21, 41, 41)).astype(np.float32)
  blob = ffn_oracle.weights_blob(variables, 2)
  for variant in (0, 2, 6):
    eng.set_option('conv_variant', variant)
    got = eng.predict(seed, img)
    want = ffn_oracle.forward(img, seed, blob, 2)
    assert np.abs(got - want).max() <= TOL, variant
  eng.close()


def test_canvas_step_matches_oracle(engine, fib25_blob):
  """gather + conv + disco + paste + faces + point reads on the device."""
  from ffn_amd import _lib
  from ffn_amd import synthetic
  from oracle import ffn_oracle
  vol = synthetic.normalize(synthetic.cells_volume((64, 60, 72), seed=9))
  canvas = engine.create_canvas(vol)
  oc = ffn_oracle.OracleCanvas(vol, fib25_blob, 12, (33, 33, 33), (8, 8, 8),
                               ffn_oracle.Options())
  start = (30, 30, 36)
  canvas.init_seed(start, oc.init_activation)
  oc.seed[start] = oc.init_activation
  params = _lib.StepParams(oc.pad_value, oc.move_threshold,
                           oc.disco_seed_threshold,
                           float(np.float32(ffn_oracle.logit(0.8))))
  positions = [start, (30, 30, 44), (38, 30, 36), (30, 22, 36), (30, 30, 44)]
  cands = [(30, 30, 44), (38, 30, 36), (20, 20, 20), (47, 43, 55)]
  deleted_ties = []
  for pos in positions:
    req = _lib.StepRequest()
    req.pos[:] = pos
    req.start_pos[:] = start
    req.num_candidates = len(cands)
    for k, c in enumerate(cands):
      req.candidates[k][:] = c
    res = engine.step1(canvas, req, params)
    logits = oc.update_at(pos)
    scores, idx = ffn_oracle.face_maxima((8, 8, 8), logits)
    assert np.allclose(list(res.face_score), scores, atol=TOL)
    assert list(res.face_index) == [int(i) for i in idx]
    assert abs(res.start_logit - oc.seed[start]) <= TOL
    # exact, up to the voxels whose logit is within TOL of the threshold (0):
    # those the float tolerance itself cannot place (counted by the oracle)
    assert abs(int(res.num_deleted) - oc.last_deleted) <= oc.last_deleted_ties
    deleted_ties.append(oc.last_deleted_ties)
    for k, c in enumerate(cands):
      a, b = res.cand_seed[k], oc.seed[c]
      assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= TOL
      assert res.cand_seg[k] == 0
    got = canvas.read_seed()
    assert np.array_equal(np.isnan(got), np.isnan(oc.seed))
    assert np.nanmax(np.abs(got - oc.seed)) <= TOL
  print('history_deleted: exact on %d of %d steps (ties within TOL of 0: %s)' % (
      sum(1 for t in deleted_ties if t == 0), len(deleted_ties), deleted_ties))
  canvas.close()


def test_canvas_utilities_bit_exact(engine):
  """point / box / commit kernels are integer work: exact."""
  from ffn_amd import synthetic
  rng = np.random.RandomState(0)
  shape = (40, 37, 45)
  canvas = engine.create_canvas(rng.normal(0, 1, shape).astype(np.float32))
  seed = rng.normal(0, 2, shape).astype(np.float32)
  seed[rng.rand(*shape) < 0.3] = np.nan
  seg = rng.randint(-1, 6, shape).astype(np.int32)
  canvas.write_seed((0, 0, 0), shape, seed)
  canvas.write_segmentation((0, 0, 0), shape, seg)
  assert np.array_equal(canvas.read_seed(), seed, equal_nan=True)
  assert np.array_equal(canvas.read_segmentation(), seg)
  lo, hi = (3, 5, 7), (33, 30, 41)
  sel = tuple(slice(l, h) for l, h in zip(lo, hi))
  assert np.array_equal(canvas.read_seed(lo, hi), seed[sel], equal_nan=True)
  assert np.array_equal(canvas.read_segmentation(lo, hi), seg[sel])
  # commit count
  thr = 0.4054652154
  raw, actual, ids, counts = canvas.commit_count(lo, hi, thr, 5)
  mask = seed[sel] >= np.float32(thr)
  assert raw == int(mask.sum())
  assert actual == int((mask & (seg[sel] <= 0)).sum())
  uid, ucnt = np.unique(seg[sel][mask], return_counts=True)
  keep = uid > 0
  assert list(ids) == list(uid[keep]) and list(counts) == list(ucnt[keep])
  canvas.commit_assign(lo, hi, thr, 77)
  want = seg.copy()
  want[sel][mask & (seg[sel] <= 0)] = 77
  assert np.array_equal(canvas.read_segmentation(), want)
  # any_segmented / points
  assert canvas.any_segmented((0, 0, 0), (3, 3, 3)) == bool(
      np.any(want[:3, :3, :3] > 0))
  pts = rng.randint(0, 37, (50, 3))
  s, g = canvas.read_points(pts)
  assert np.array_equal(s, seed[tuple(pts.T)], equal_nan=True)
  assert np.array_equal(g, want[tuple(pts.T)])
  canvas.write_seg_points(pts[:5], [-1] * 5)
  assert all(canvas.read_point(p)[1] == -1 for p in pts[:5])
  # init_seed: everything NaN except the seed voxel
  canvas.init_seed((20, 20, 20), 2.9444387)
  out = canvas.read_seed()
  assert np.isnan(out).sum() == out.size - 1
  assert out[20, 20, 20] == np.float32(2.9444387)
  canvas.close()


def test_segment_turn_equals_the_single_questions(engine):
  """ffn_canvas_segment_turn (commit count -> assign or -1 marker -> the next
  seeds tested in order, too-close ones marked -> init_seed at the first that
  passes; reference inference.py:573-660) against the numpy statement of the same
  sequence (tests/emulated_device.py), on random canvases: integer work, exact."""
  from tests.emulated_device import EmulatedHandle
  rng = np.random.RandomState(5)
  shape = (40, 37, 45)
  for case in range(12):
    image = np.zeros(shape, np.float32)
    seed = rng.normal(0, 2, shape).astype(np.float32)
    seed[rng.rand(*shape) < 0.3] = np.nan
    # sparse ids, so that some candidates pass / are too close / are inside one
    seg = np.zeros(shape, np.int32)
    seg[rng.rand(*shape) < (0.002, 0.02, 0.3)[case % 3]] = rng.randint(1, 6)
    seg[rng.rand(*shape) < 0.01] = -1
    canvas = engine.create_canvas(image)
    emu = EmulatedHandle(image)
    if case % 2:
      # the canvas tracks the region that steps and writes touched and init_seed
      # clears only that: a small one here (box fill), the whole volume otherwise
      blo, bhi = (4, 6, 8), (30, 31, 40)
      bsel = tuple(slice(l, h) for l, h in zip(blo, bhi))
      canvas.init_seed((1, 1, 1), 0.5)
      emu.init_seed((1, 1, 1), 0.5)
      canvas.write_seed(blo, bhi, seed[bsel])
      emu.seed[bsel] = seed[bsel]
    else:
      canvas.write_seed((0, 0, 0), shape, seed)
      emu.seed[...] = seed
    canvas.write_segmentation((0, 0, 0), shape, seg)
    emu.seg[...] = seg
    lo = [int(rng.randint(0, 10)) for _ in range(3)]
    hi = [int(rng.randint(25, s + 1)) for s in shape]
    thr = 0.4054652154
    min_size = (0, 10 ** 9)[case % 2] if case % 4 != 3 else int(
        EmulatedHandle.commit_count(emu, lo, hi, thr, 5)[1])
    commit = None if case % 5 == 4 else (lo, hi, thr, min_size, 77, 5)
    mark_pos = tuple(int(rng.randint(0, s)) for s in shape)
    mark = (None, (mark_pos, 1), (mark_pos, 2))[case % 3]
    n = (0, 1, 40, 200)[case % 4]
    cands = np.stack([rng.randint(0, s, n) for s in shape], axis=1).astype(np.int32)
    mbd = [(1, 1, 1), (0, 0, 0), (2, 1, 3)][case % 3]
    init = None if case % 6 == 5 else 2.9444387
    got = canvas.segment_turn(commit, mark, cands, mbd, init)
    want = emu.segment_turn(commit, mark, cands, mbd, init)
    assert got[0] == want[0] and got[1] == want[1], case
    assert list(got[2]) == list(want[2]) and list(got[3]) == list(want[3]), case
    assert got[4] == want[4] and got[5] == want[5], case
    assert np.array_equal(got[6], want[6]), case
    k = n if want[5] < 0 else want[5] + 1  # candidates that were looked at
    assert np.array_equal(got[7][:k], want[7][:k], equal_nan=True), case
    assert np.array_equal(got[8][:k], want[8][:k]), case
    assert np.array_equal(canvas.read_segmentation(), emu.seg), case
    assert np.array_equal(canvas.read_seed(), emu.seed, equal_nan=True), case
    # the canvas goes on as after init_seed: one more (plain) init elsewhere
    canvas.init_seed((20, 20, 20), 1.0)
    out = canvas.read_seed()
    assert np.isnan(out).sum() == out.size - 1 and out[20, 20, 20] == 1.0
    canvas.close()


@pytest.mark.parametrize('name', ['cells56', 'cells72'])
def test_device_canvas_reproduces_reference_run(fib25_model, name):
  """Full Canvas.segment_all on the GPU vs the fixture minted by the
  reference's own Python (tools/make_golden.py): identical FoV trajectory,
  segment ids and counters; seed logits within tolerance."""
  from ffn_amd import synthetic
  from ffn_amd.inference import executor
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  from ffn_amd.inference import seed as seed_lib
  import bench
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_%s.npz' % name))
  request = bench.make_request()
  counters = inference_utils.Counters()
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), fib25_model,
                                  fib25_model.info, None, counters, 1)
  image = synthetic.normalize(g['volume'])
  steps = []

  class Rec(inference.DeviceCanvas):

    def update_at(self, pos):
      steps.append(tuple(pos))
      return super().update_at(pos)

  canvas = Rec(fib25_model.info, exe.get_client(counters, direct=True), image,
               request.inference_options, counters=counters,
               movement_policy_fn=movement.get_policy_fn(request,
                                                         fib25_model.info))
  canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                   coords=g['seeds']))
  assert np.array_equal(np.array(steps).reshape(-1, 3), g['steps'])
  assert np.array_equal(np.asarray(canvas.segmentation), g['segmentation'])
  got_seed = np.asarray(canvas.seed)
  assert np.array_equal(np.isnan(got_seed), np.isnan(g['seed_logits']))
  assert np.nanmax(np.abs(got_seed - g['seed_logits'])) <= TOL
  ref_counters = json.loads(str(g['counters']))
  for key in ('update_at-calls', 'voxels-segmented', 'voxels-overlapping',
              'skip_invalid_pos', 'skip_threshold', 'seed_got_too_weak',
              'segment_at-loop-calls', 'seed-policy-calls'):
    if key in ref_counters:
      assert counters[key].value == ref_counters[key], key
  origins = json.loads(str(g['origins']))
  assert {int(k): [list(v.start_zyx), v.iters]
          for k, v in canvas.origins.items()} == {
              int(k): v for k, v in origins.items()}
  canvas.close()


def test_reference_style_host_canvas_through_threaded_executor(fib25_model):
  """The literal plug-in boundary: host Canvas + client/server threads +
  stateless ffn_predict (what an unmodified reference Canvas would use)."""
  from ffn_amd import synthetic
  from ffn_amd.inference import executor
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  from ffn_amd.inference import seed as seed_lib
  import bench
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells56.npz'))
  request = bench.make_request()
  counters = inference_utils.Counters()
  iface = executor.ExecutorInterface()
  exe = executor.HipBatchExecutor(iface, fib25_model, fib25_model.info, None,
                                  counters, 1)
  exe.start_server()
  client = executor.ThreadingExecutorClient(counters, iface)
  canvas = inference.Canvas(fib25_model.info, client,
                            synthetic.normalize(g['volume']),
                            request.inference_options, counters=counters,
                            movement_policy_fn=movement.get_policy_fn(
                                request, fib25_model.info))
  canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                   coords=g['seeds']))
  exe.stop_server()
  assert np.array_equal(np.asarray(canvas.segmentation), g['segmentation'])
  assert counters['update_at-calls'].value == len(g['steps'])


class _Halt(Exception):
  """Stops a segment after a fixed number of steps (not StopIteration: the
  canvas loops are generators, PEP 479)."""


def test_full_size_properties_250(fib25_model):
  """BASELINE size (250^3): size-independent properties instead of the oracle:
  (1) re-running the same segment gives the identical trajectory/mask
  (determinism), (2) the paste only touches the FoV box, (3) the device commit
  counts equal a host recount of the downloaded arrays."""
  from ffn_amd import synthetic
  from ffn_amd.inference import executor
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  import bench
  request = bench.make_request()
  counters = inference_utils.Counters()
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), fib25_model,
                                  fib25_model.info, None, counters, 1)
  vol = synthetic.cells_volume((250, 250, 250), seed=1234)
  canvas = inference.DeviceCanvas(
      fib25_model.info, exe.get_client(counters, direct=True),
      synthetic.normalize(vol), request.inference_options, counters=counters,
      movement_policy_fn=movement.get_policy_fn(request, fib25_model.info))
  start = (120, 120, 120)
  runs = []
  for _ in range(2):
    steps = []
    orig = canvas.update_at

    def rec(pos, _o=orig, _s=steps):
      _s.append(tuple(pos))
      if len(_s) > 60:
        raise _Halt()
      return _o(pos)

    canvas.update_at = rec
    try:
      canvas.segment_at(start)
    except _Halt:
      pass
    canvas.update_at = orig
    runs.append((list(steps), canvas._handle.read_seed()))
  assert runs[0][0] == runs[1][0]
  assert np.array_equal(runs[0][1], runs[1][1], equal_nan=True)
  seed = runs[1][1]
  touched = ~np.isnan(seed)
  pts = np.array(runs[1][0][:-1] if len(runs[1][0]) > 60 else runs[1][0])
  lo = np.maximum(pts.min(0) - 16, 0)
  hi = pts.max(0) + 17
  outside = touched.copy()
  outside[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = False
  assert not outside.any()
  thr = canvas.options.segment_threshold
  raw, actual, ids, counts = canvas._handle.commit_count(
      [int(v) for v in lo], [int(min(h, 250)) for h in hi], thr, 0)
  assert raw == int(np.sum(seed >= np.float32(thr)))
  assert actual == raw and len(ids) == 0
  canvas.close()


def test_batched_device_canvases_through_threaded_executor(fib25_model):
  """Config C3 shape: several DeviceCanvas client threads, one server thread
  batching their FoV steps into single ffn_canvas_step(n, ...) calls.  Every
  canvas must reproduce the reference trajectory independently of batching."""
  import threading
  from ffn_amd import synthetic
  from ffn_amd.inference import executor
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  from ffn_amd.inference import seed as seed_lib
  import bench
  names = ['cells72', 'cells56', 'cells72', 'cells56', 'cells56']
  gold = {n: np.load(os.path.join(GOLDEN, 'ref_canvas_%s.npz' % n))
          for n in set(names)}
  request = bench.make_request()
  counters = inference_utils.Counters()
  iface = executor.ExecutorInterface()
  exe = executor.HipBatchExecutor(iface, fib25_model, fib25_model.info, None,
                                  counters, batch_size=4,
                                  expected_clients=len(names))
  exe.start_server()
  out = {}
  errors = []

  def work(k, name):
    try:
      g = gold[name]
      sub = counters.get_sub_counters()
      canvas = inference.DeviceCanvas(
          fib25_model.info, exe.get_client(sub), synthetic.normalize(g['volume']),
          request.inference_options, counters=sub,
          movement_policy_fn=movement.get_policy_fn(request, fib25_model.info))
      canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                       coords=g['seeds']))
      out[k] = (np.asarray(canvas.segmentation),
                sub['update_at-calls'].value)
      canvas.close()
    except Exception as e:  # pylint:disable=broad-except
      errors.append(repr(e))

  threads = [threading.Thread(target=work, args=(k, n), daemon=True)
             for k, n in enumerate(names)]
  for t in threads:
    t.start()
  for t in threads:
    t.join(timeout=600)
  exe.stop_server()
  assert not errors, errors
  for k, n in enumerate(names):
    assert np.array_equal(out[k][0], gold[n]['segmentation']), (k, n)
    assert out[k][1] == len(gold[n]['steps'])
  # steps were really batched (fewer engine calls than FoV steps)
  total_steps = sum(len(gold[n]['steps']) for n in names)
  assert counters['executor-inference-calls'].value < total_steps


def _runner_request_text(g, tmp_path, out_dir):
  """The sample config's request (configs/inference_training_sample2.pbtxt) on
  an `npy:` volume with the fixture's fixed seeds, parsed from text format."""
  import json as _json
  from ffn_amd.inference import request as req_lib
  vol_path = str(tmp_path / 'vol.npy')
  np.save(vol_path, g['volume'])
  weights = os.path.join(GOLDEN, 'fib25_weights.npz')
  seeds = _json.dumps({'coords': g['seeds'].tolist()}).replace('"', '\\"')
  text = '''
    image { npy: "%s" }
    image_mean: 128
    image_stddev: 33
    checkpoint_interval: 1800
    seed_policy: "PolicyFixed"
    seed_policy_args: "%s"
    model_checkpoint_path: "%s"
    model_name: "convstack_3d.ConvStack3DFFNModel"
    model_args: "{\\"depth\\": 12, \\"fov_size\\": [33, 33, 33], \\"deltas\\": [8, 8, 8]}"
    segmentation_output_dir: "%s"
    inference_options {
      init_activation: 0.95
      pad_value: 0.05
      move_threshold: 0.9
      min_boundary_dist { x: 1 y: 1 z: 1}
      segment_threshold: 0.6
      min_segment_size: 1000
    }''' % (vol_path, seeds, weights, out_dir)
  return req_lib.request_from_text(text)


def test_runner_end_to_end_writes_reference_format(fib25_model, tmp_path):
  """run_inference.py flow: InferenceRequest (text format) -> Runner.start ->
  Runner.run -> seg-*.npz with the reference's keys; the segmentation equals
  the reference-minted fixture; a second run() skips the finished subvolume."""
  import json as _json
  from ffn_amd.inference import runner as runner_lib
  from ffn_amd.inference import storage
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells56.npz'))
  out_dir = str(tmp_path / 'out')
  request = _runner_request_text(g, tmp_path, out_dir)
  runner = runner_lib.Runner()
  runner.start(request)
  canvas = runner.run((0, 0, 0), tuple(g['volume'].shape))
  assert canvas is not None
  path = storage.segmentation_path(out_dir, (0, 0, 0))
  assert path.endswith(os.path.join('0', '0', 'seg-0_0_0.npz'))
  with np.load(path, allow_pickle=True) as d:
    assert sorted(d.files) == ['counters', 'origins', 'overlaps', 'request',
                               'segmentation']
    want = g['segmentation'].copy()
    want[want < 0] = 0
    assert np.array_equal(d['segmentation'], want)
    assert d['segmentation'].dtype == np.uint8
    counters = _json.loads(str(d['counters']))
    assert counters['update_at-calls'] == len(g['steps'])
    origins = d['origins'].item()
    ref_origins = _json.loads(str(g['origins']))
    assert {int(k): [list(v.start_zyx), v.iters]
            for k, v in origins.items()} == {int(k): v
                                             for k, v in ref_origins.items()}
  assert not os.path.exists(storage.checkpoint_path(out_dir, (0, 0, 0)))
  assert runner.run((0, 0, 0), tuple(g['volume'].shape)) is None  # already done
  seg, org = storage.load_segmentation(out_dir, (0, 0, 0), split_cc=False)
  assert np.array_equal(seg, want)
  runner.stop_executor()


def test_abi_rejects_bad_arguments(engine):
  """Error behaviour of the C-ABI: negative return code + message, no crash."""
  import ctypes
  from ffn_amd import _lib
  lib = _lib.load()
  h = ctypes.c_void_p()
  # even fov / wrong feature count / deltas larger than the fov
  assert lib.ffn_engine_create(0, _lib.i3((32, 33, 33)), _lib.i3((8, 8, 8)), 12,
                               32, 1, ctypes.byref(h)) == -1
  assert lib.ffn_engine_create(0, _lib.i3((33, 33, 33)), _lib.i3((8, 8, 8)), 12,
                               16, 1, ctypes.byref(h)) == -1
  assert lib.ffn_engine_create(0, _lib.i3((33, 33, 33)), _lib.i3((20, 8, 8)),
                               12, 32, 1, ctypes.byref(h)) == -1
  assert b'deltas' in lib.ffn_last_error()
  assert lib.ffn_engine_create(99, _lib.i3((33, 33, 33)), _lib.i3((8, 8, 8)),
                               12, 32, 1, ctypes.byref(h)) == -1
  with pytest.raises(_lib.FFNHipError):
    engine.set_weights(np.zeros(10, np.float32))
  with pytest.raises(_lib.FFNHipError):  # batch larger than max_batch
    engine.predict(np.zeros((9, 33, 33, 33), np.float32),
                   np.zeros((9, 33, 33, 33), np.float32))
  # FoV leaving the canvas / canvas smaller than the FoV
  small = engine.create_canvas(np.zeros((20, 40, 40), np.float32))
  req = _lib.StepRequest()
  req.pos[:] = (10, 20, 20)
  req.start_pos[:] = (10, 20, 20)
  with pytest.raises(_lib.FFNHipError):
    engine.step1(small, req, _lib.StepParams(-2.9, 2.2, 0.0))
  with pytest.raises(_lib.FFNHipError):
    small.read_seed((0, 0, 0), (21, 40, 40))
  small.close()


def test_empty_and_exhausted_workloads(fib25_model):
  """Edge cases the reference handles: no seeds at all, every seed too close to
  the border, canvas smaller than the FoV -> empty segmentation, zero steps."""
  from ffn_amd.inference import executor
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import seed as seed_lib
  import bench
  request = bench.make_request()
  for shape, coords in [((40, 40, 40), []),
                        ((40, 40, 40), [(2, 2, 2), (39, 20, 20)]),
                        ((20, 34, 34), [(10, 17, 17)])]:
    counters = inference_utils.Counters()
    exe = executor.HipBatchExecutor(executor.ExecutorInterface(), fib25_model,
                                    fib25_model.info, None, counters, 1)
    canvas = inference.DeviceCanvas(
        fib25_model.info, exe.get_client(counters, direct=True),
        np.zeros(shape, np.float32), request.inference_options,
        counters=counters)
    canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                     coords=coords))
    assert counters['update_at-calls'].value == 0
    assert not np.asarray(canvas.segmentation).any()
    assert canvas.origins == {}
    canvas.close()


def test_sharded_volume_end_to_end(fib25_model, tmp_path):
  """Configs C3/C4 on one GPU: the bounding box is tiled into overlapping
  sub-boxes that advance CONCURRENTLY through batched engine calls
  (Runner.run_many), seeds come from the GPU PolicyPeaks, and the sub-box
  results are assembled + reconciled into one label volume.

  Checked: (1) every sub-box equals a standalone Runner.run of the same
  sub-box (batching changes nothing); (2) the assembly equals the oracle's
  single-process specification; (3) reconciliation merges ids across cuts."""
  from ffn_amd import distributed as ffn_dist
  from ffn_amd import synthetic
  from ffn_amd.inference import request as req_lib
  from ffn_amd.inference import runner as runner_lib
  from ffn_amd.inference import storage
  from oracle import labels_oracle
  shape = (72, 80, 112)
  vol = synthetic.cells_volume(shape, seed=77, membrane_dilate=2)
  vol_path = str(tmp_path / 'vol.npy')
  np.save(vol_path, vol)
  weights = os.path.join(GOLDEN, 'fib25_weights.npz')

  def make_request(out_dir):
    return req_lib.request_from_text('''
      image { npy: "%s" }
      image_mean: 128
      image_stddev: 33
      seed_policy: "PolicyPeaks"
      model_checkpoint_path: "%s"
      model_name: "convstack_3d.ConvStack3DFFNModel"
      model_args: "{\\"depth\\": 12, \\"fov_size\\": [33, 33, 33], \\"deltas\\": [8, 8, 8]}"
      segmentation_output_dir: "%s"
      inference_options {
        init_activation: 0.95
        pad_value: 0.05
        move_threshold: 0.9
        min_boundary_dist { x: 1 y: 1 z: 1}
        segment_threshold: 0.6
        min_segment_size: 500
      }''' % (vol_path, weights, out_dir))

  sub, ov = (72, 80, 72), (33, 33, 33)
  runner = runner_lib.Runner()
  runner.start(make_request(str(tmp_path / 'sharded')), batch_size=4,
               direct=True)
  merged, info = ffn_dist.segment_volume(
      runner, (0, 0, 0), shape, sub, ov, batch_size=4, min_overlap_voxels=32,
      min_overlap_fraction=0.2)
  runner.stop_executor()
  boxes = info['boxes']
  assert len(boxes) == 3 and merged.shape == shape
  assert merged.dtype == np.int32 and merged.max() > 0

  # (1) standalone runs of each sub-box: same segmentation, same files
  solo = runner_lib.Runner()
  solo.start(make_request(str(tmp_path / 'solo')))
  for box, seg in info['local_results']:
    canvas = solo.run(box.corner, box.size)
    want = np.array(np.asarray(canvas.segmentation))
    want[want < 0] = 0
    assert np.array_equal(seg, want)
    a, _ = storage.load_segmentation(str(tmp_path / 'sharded'), box.corner,
                                     split_cc=False)
    assert np.array_equal(a, want)
  solo.stop_executor()

  # (2) assembly == specification; (3) something was merged across the cut
  want, want_edges, want_roots = labels_oracle.reconcile(
      info['local_results'], shape, 32, 0.2)
  assert np.array_equal(merged, want)
  assert np.array_equal(info['edges'], want_edges)
  assert info['roots'] == want_roots
  plain, _ = ffn_dist.merge_segmentations(info['local_results'], shape, 0, 1)
  assert len(info['edges']) > 0
  assert len(np.unique(merged)) < len(np.unique(plain))
  # (4) the same assembly with everything resident on the device
  on_dev, _, edges_dev, _ = ffn_dist.reconcile_segmentations(
      info['local_results'], shape, 0, 1, device='cuda:0', min_overlap_voxels=32,
      min_overlap_fraction=0.2)
  assert np.array_equal(on_dev, want) and np.array_equal(edges_dev, want_edges)


def test_anisotropic_canvas_step_matches_oracle():
  """C5 geometry on the canvas path: FoV zyx (21, 41, 41), deltas (5, 10, 10),
  depth 3: gather, paste, disco and the six NON-square face argmaxes
  (21x21 / 11x21 / 11x21) against the oracle, step by step."""
  from ffn_amd import _lib
  from ffn_amd import engine as hip_engine
  from ffn_amd import synthetic
  from ffn_amd.training.models import convstack_3d
  from oracle import ffn_oracle
  fov, deltas, depth = (21, 41, 41), (5, 10, 10), 3
  variables = ffn_oracle.random_weights(depth, seed=8, stddev=0.06)
  m = convstack_3d.ConvStack3DFFNModel(fov_size=list(fov[::-1]),
                                       deltas=list(deltas[::-1]), depth=depth)
  m.set_variables(variables)
  eng = hip_engine.HipEngine.from_model(m, max_batch=1)
  blob = ffn_oracle.weights_blob(variables, depth)
  vol = synthetic.normalize(synthetic.cells_volume((40, 90, 96), seed=3))
  canvas = eng.create_canvas(vol)
  oc = ffn_oracle.OracleCanvas(vol, blob, depth, fov, deltas,
                               ffn_oracle.Options())
  start = (20, 45, 48)
  canvas.init_seed(start, oc.init_activation)
  oc.seed[start] = oc.init_activation
  params = _lib.StepParams(oc.pad_value, oc.move_threshold,
                           oc.disco_seed_threshold)
  positions = [start, (20, 45, 58), (25, 45, 48), (20, 35, 48), (15, 50, 53),
               (20, 45, 58)]
  for pos in positions:
    req = _lib.StepRequest()
    req.pos[:] = pos
    req.start_pos[:] = start
    req.num_candidates = 0
    res = eng.step1(canvas, req, params)
    logits = oc.update_at(pos)
    scores, idx = ffn_oracle.face_maxima(deltas, logits)
    assert np.allclose(list(res.face_score), scores, atol=TOL)
    assert list(res.face_index) == [int(i) for i in idx]
    got = canvas.read_seed()
    assert np.array_equal(np.isnan(got), np.isnan(oc.seed))
    assert np.nanmax(np.abs(got - oc.seed)) <= TOL
  canvas.close()
  eng.close()


def test_step_submit_wait_contract(engine, fib25_blob):
  """Two steps in flight: results identical to blocking steps, and the error
  behaviour of the split call (third submit, canvas in two steps, bad ticket)."""
  import ctypes
  from ffn_amd import _lib
  from ffn_amd import synthetic
  vols = [synthetic.normalize(synthetic.cells_volume((48, 48, 48), seed=40 + k))
          for k in range(4)]
  params = _lib.StepParams(-2.9444389343, 2.1972243786, 0.0)
  start = (24, 24, 24)

  def make():
    cs = [engine.create_canvas(v) for v in vols]
    for c in cs:
      c.init_seed(start, 2.9444386959)
    return cs

  def req(pos):
    r = _lib.StepRequest()
    r.pos[:] = pos
    r.start_pos[:] = start
    r.num_candidates = 0
    return r

  def snapshot(res, n):
    return [(tuple(res[k].face_score), tuple(res[k].face_index),
             res[k].start_logit, res[k].num_above_move) for k in range(n)]

  # blocking reference: canvases {0,1} then {2,3}, two rounds
  a = make()
  want = []
  for pos in (start, (24, 24, 30)):
    want.append(snapshot(engine.step(a[:2], [req(pos)] * 2, params), 2))
    want.append(snapshot(engine.step(a[2:], [req(pos)] * 2, params), 2))
  seeds_want = [c.read_seed() for c in a]
  for c in a:
    c.close()
  # the same with both groups in flight
  b = make()
  got = []
  for pos in (start, (24, 24, 30)):
    t0 = engine.step_submit(b[:2], [req(pos)] * 2, params)
    t1 = engine.step_submit(b[2:], [req(pos)] * 2, params)
    lib = _lib.load()
    # a third step cannot be queued; neither can a canvas already in flight
    tk = ctypes.c_uint32(0)
    arr = (ctypes.c_void_p * 1)()
    arr[0] = b[0]._h
    r1 = req(pos)
    assert lib.ffn_canvas_step_submit(engine._h, 1, arr,
                                      ctypes.byref(r1),
                                      ctypes.byref(params),
                                      ctypes.byref(tk)) < 0
    assert b'in flight' in lib.ffn_last_error()
    got.append(snapshot(engine.step_wait(t0), 2))
    got.append(snapshot(engine.step_wait(t1), 2))
  assert got == want
  for c, s in zip(b, seeds_want):
    assert np.array_equal(c.read_seed(), s, equal_nan=True)
  # one slot busy: a canvas of that step cannot join the next one
  t0 = engine.step_submit(b[:1], [req(start)], params)
  with pytest.raises(_lib.FFNHipError, match='in flight'):
    engine.step_submit(b[:1], [req(start)], params)
  engine.step_wait(t0)
  res = (_lib.StepResult * 1)()
  assert _lib.load().ffn_canvas_step_wait(engine._h, 123456789, res) < 0
  for c in b:
    c.close()


def test_large_canvas_offsets_beyond_2gib(engine, fib25_blob):
  """A canvas whose arrays exceed 2 GiB (539 M voxels): a step in the far
  corner must address the right voxels (size_t index math in every kernel)."""
  from ffn_amd import _lib
  from oracle import ffn_oracle
  shape = (1100, 700, 700)
  rng = np.random.RandomState(5)
  image = np.zeros(shape, np.float32)
  pos = (1100 - 20, 700 - 22, 700 - 19)
  lo = [p - 16 for p in pos]
  sel = tuple(slice(l, l + 33) for l in lo)
  image[sel] = rng.normal(0, 1, (33, 33, 33)).astype(np.float32)
  canvas = engine.create_canvas(image)
  del image
  canvas.init_seed(pos, 2.9444386959)
  params = _lib.StepParams(-2.9444389343, 2.1972243786, 0.0)
  req = _lib.StepRequest()
  req.pos[:] = pos
  req.start_pos[:] = pos
  req.num_candidates = 1
  req.candidates[0][:] = (pos[0] - 8, pos[1], pos[2])
  res = engine.step1(canvas, req, params)
  rng = np.random.RandomState(5)
  fov = rng.normal(0, 1, (33, 33, 33)).astype(np.float32)
  seed = np.full((33, 33, 33), np.float32(-2.9444389343), np.float32)
  seed[16, 16, 16] = np.float32(2.9444386959)
  want = ffn_oracle.forward(fov[None], seed[None], fib25_blob, 12)[0]
  got = canvas.read_seed(lo, [l + 33 for l in lo])
  # disco bias: positions whose old seed was NaN keep the new logits
  assert np.abs(got - want).max() <= TOL
  assert abs(res.start_logit - want[16, 16, 16]) <= TOL
  assert abs(res.cand_seed[0] - want[8, 16, 16]) <= TOL
  # nothing else was touched; a far-away voxel is still NaN, the point API agrees
  assert np.isnan(canvas.read_seed((0, 0, 0), (4, 4, 4))).all()
  sv, gv = canvas.read_point((lo[0] - 1, lo[1], lo[2]))
  assert np.isnan(sv) and gv == 0
  # second seed elsewhere: the dirty-box clear wipes the first FoV
  canvas.init_seed((40, 40, 40), 1.0)
  assert np.isnan(canvas.read_seed(lo, [l + 33 for l in lo])).all()
  assert canvas.read_point((40, 40, 40))[0] == 1.0
  canvas.close()


@pytest.mark.parametrize('fast', [9, 8, 6])
def test_fp16_range_fallback(fib25_model, fib25_blob, fast):
  """conv_variants 6 .. 9 keep operands in fp16: a value beyond 65504 must void
  the run (nothing pasted) and repeat it with the exact-f32 kernel -- silently for
  ffn_predict, through FFN_ERR_RANGE + retry for canvas steps."""
  from ffn_amd import _lib
  from ffn_amd import engine as hip_engine
  from oracle import ffn_oracle
  eng = hip_engine.HipEngine.from_model(fib25_model, max_batch=1)
  assert eng.get_option('conv_variant') == DEFAULT_VARIANT
  eng.set_option('conv_variant', fast)
  rng = np.random.RandomState(12)
  img, seed = _fov_inputs(rng, 1)
  ok = eng.predict(seed, img)
  assert eng.get_option('conv_variant') == fast  # ordinary data stays on fp16 pairs
  big = (img * 3e5).astype(np.float32)  # conv0_a outputs far beyond 65504
  got = eng.predict(seed, big)
  assert eng.get_option('conv_variant') == 2 == eng.get_option('exact_variant')
  want = ffn_oracle.forward(big, seed, fib25_blob, 12)
  assert np.isfinite(got).all()
  assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
  assert np.array_equal(eng.predict(seed, img), eng.predict(seed, img))
  assert np.abs(eng.predict(seed, img) - ok).max() <= 2e-5
  # canvas step: the voided step must leave the canvas untouched, then repeat
  eng.set_option('conv_variant', fast)
  vol = np.zeros((40, 40, 40), np.float32)
  vol[4:37, 4:37, 4:37] = big[0]
  canvas = eng.create_canvas(vol)
  start = (20, 20, 20)
  canvas.init_seed(start, 2.9444386959)
  params = _lib.StepParams(-2.9444389343, 2.1972243786, 0.0)
  req = _lib.StepRequest()
  req.pos[:] = start
  req.start_pos[:] = start
  req.num_candidates = 0
  lib = _lib.load()
  arr = (ctypes.c_void_p * 1)()
  arr[0] = canvas._h
  res = (_lib.StepResult * 1)()
  rc = lib.ffn_canvas_step(eng._h, 1, arr, ctypes.byref(req),
                           ctypes.byref(params), res)
  assert rc == _lib.ERR_RANGE and res[0].range_error == 1
  seed_now = canvas.read_seed()
  assert np.isnan(seed_now).sum() == seed_now.size - 1  # nothing was pasted
  eng.set_option('conv_variant', fast)
  r = eng.step1(canvas, req, params)  # Python handle: retries with bf16x3
  assert eng.range_fallbacks == 1 and eng.get_option('conv_variant') == 2
  assert r.range_error == 0 and np.isfinite(r.start_logit)
  assert np.isfinite(canvas.read_seed((4, 4, 4), (37, 37, 37))).all()
  canvas.close()
  eng.close()


def test_native_segment_loop_matches_python_loop(fib25_model):
  """ffn_canvas_segment_at (the FoV loop of a segment inside the library) ==
  the Python loop over ffn_canvas_step: same segmentation, seed logits,
  histories, counters -- and both equal the reference-minted run."""
  from ffn_amd import synthetic
  from ffn_amd.inference import executor
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  from ffn_amd.inference import seed as seed_lib
  import bench
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))
  request = bench.make_request()
  image = synthetic.normalize(g['volume'])

  class PythonLoop(inference.DeviceCanvas):
    NATIVE_LOOP = False

  runs = {}
  for cls in (inference.DeviceCanvas, PythonLoop):
    counters = inference_utils.Counters()
    exe = executor.HipBatchExecutor(executor.ExecutorInterface(), fib25_model,
                                    fib25_model.info, None, counters, 1)
    canvas = cls(fib25_model.info, exe.get_client(counters, direct=True), image,
                 request.inference_options, counters=counters,
                 keep_history=True,
                 movement_policy_fn=movement.get_policy_fn(request,
                                                           fib25_model.info))
    assert canvas._native_loop_ok() == (cls is inference.DeviceCanvas)
    n = canvas.segment_at((16, 16, 32))
    first = dict(n=n, history=list(canvas.history),
                 deleted=list(canvas.history_deleted),
                 mn=canvas._min_pos.tolist(), mx=canvas._max_pos.tolist(),
                 seed=np.array(canvas._handle.read_seed()))
    canvas.segment_all(seed_policy=functools.partial(seed_lib.PolicyFixed,
                                                     coords=g['seeds']))
    runs[cls.__name__] = dict(
        first=first, seg=np.array(np.asarray(canvas.segmentation)),
        seed=np.array(canvas._handle.read_seed()),
        rejects=canvas.gate_rejects,
        counters={k: counters[k].value for k in (
            'update_at-calls', 'skip_threshold', 'skip_invalid_pos',
            'seed_got_too_weak', 'segment_at-loop-calls', 'voxels-segmented',
            'movement_policy-calls')})
    canvas.close()
  a, b = runs['DeviceCanvas'], runs['PythonLoop']
  assert a['first']['n'] == b['first']['n'] > 7
  for key in ('history', 'deleted', 'mn', 'mx'):
    assert a['first'][key] == b['first'][key], key
  assert np.array_equal(a['first']['seed'], b['first']['seed'], equal_nan=True)
  assert a['counters'] == b['counters'] and a['rejects'] == b['rejects']
  assert np.array_equal(a['seg'], b['seg'])
  assert np.array_equal(a['seed'], b['seed'], equal_nan=True)
  # the extra segment_at in front changes nothing the golden run pins but ids
  assert np.array_equal(a['seg'] > 0, g['segmentation'] > 0)


def test_native_segment_loop_budget_and_resume(fib25_model):
  """max_steps + resume over the real device: the same trajectory as one call."""
  from ffn_amd import synthetic
  from ffn_amd.inference import executor
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  import bench
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells72.npz'))
  request = bench.make_request()
  image = synthetic.normalize(g['volume'])
  out = []
  for budget in (0, 4):
    counters = inference_utils.Counters()
    exe = executor.HipBatchExecutor(executor.ExecutorInterface(), fib25_model,
                                    fib25_model.info, None, counters, 1)
    canvas = inference.DeviceCanvas(
        fib25_model.info, exe.get_client(counters, direct=True), image,
        request.inference_options, counters=counters, keep_history=True,
        movement_policy_fn=movement.get_policy_fn(request, fib25_model.info))
    n = canvas._segment_at_native((32, 16, 16), max_steps=budget)
    calls = 1
    while canvas._native_active:
      assert budget and n % budget == 0
      n += canvas._segment_at_native((32, 16, 16), max_steps=budget, resume=True)
      calls += 1
    out.append((n, list(canvas.history), list(canvas.history_deleted),
                np.array(canvas._handle.read_seed()), calls))
    canvas.close()
  assert out[0][0] == out[1][0] > 20 and out[1][4] > 5
  assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
  assert np.array_equal(out[0][3], out[1][3], equal_nan=True)


@pytest.mark.gpu
def test_c5_sharded_anisotropic_volume_reconciled(tmp_path):
  """BASELINE configs[4] as stated, at test size: the depth-18 model with FoV
  zyx (21, 41, 41) / deltas (5, 10, 10) on ONE anisotropic volume cut into
  overlapping ANISOTROPIC sub-boxes (overlap = FoV = 21 x 41 x 41), the sub-boxes
  advanced together through batched engine calls, their labels assembled and
  reconciled on the device.  The reference ships no checkpoint of this shape:
  the network is `synthetic.flood_fill_weights` (floods bright 26-connected
  regions), so segments end at the phantom's membranes and COMMIT.

  Checked: (0) the kernels against the C oracle on real canvas states; (1) every sub-box equals a standalone single-canvas run of the same box
  (conv32m batched == conv32mt / resident stack single); (2) the assembly
  equals the numpy specification; (3) objects were committed and merged
  across the cuts."""
  from ffn_amd import distributed as ffn_dist
  from ffn_amd import engine as hip_engine
  from ffn_amd import synthetic
  from ffn_amd.inference import request as req_lib
  from ffn_amd.inference import runner as runner_lib
  from ffn_amd.training.models import convstack_3d
  from oracle import ffn_oracle
  from oracle import labels_oracle
  depth, fov_xyz, deltas_xyz = 18, [41, 41, 21], [10, 10, 5]
  shape = (56, 150, 150)
  vol = synthetic.cells_volume(shape, seed=91, membrane_dilate=2)
  vol_path = str(tmp_path / 'vol.npy')
  np.save(vol_path, vol)
  variables = synthetic.flood_fill_weights(depth)
  weights = str(tmp_path / 'weights.npz')
  np.savez(weights, **variables)

  # (0) predict == oracle, exactly, on two FoVs of the volume with a seed blob
  m = convstack_3d.ConvStack3DFFNModel(fov_size=fov_xyz, deltas=deltas_xyz,
                                       depth=depth)
  m.set_variables(variables)
  eng = hip_engine.HipEngine.from_model(m, max_batch=2)
  img = synthetic.normalize(vol)
  blob = ffn_oracle.weights_blob(variables, depth)
  fovs, seeds = [], []
  for (z, y, x) in ((10, 30, 40), (30, 90, 70)):
    fovs.append(img[z:z + 21, y:y + 41, x:x + 41])
    sd = np.full((21, 41, 41), ffn_oracle.f32_logit(0.05), np.float32)
    sd[8:13, 18:23, 18:23] = ffn_oracle.f32_logit(0.95)
    seeds.append(sd)
  fovs, seeds = np.stack(fovs), np.stack(seeds)
  want = np.stack([ffn_oracle.forward(fovs[k], seeds[k], blob, depth)
                   for k in range(2)])
  grown = (want - seeds) > 3.0
  assert 1000 < grown[0].sum() < 0.9 * grown[0].size  # it floods, inside its cell
  for variant, flow in ((2, 0), (6, 0), (9, 2), (9, 0)):
    eng.set_option('conv_variant', variant)
    eng.set_option('flow', flow)
    assert np.abs(eng.predict(seeds, fovs) - want).max() <= TOL, (variant, flow)
    for k in range(2):
      one = eng.predict(seeds[k:k + 1], fovs[k:k + 1])[0]
      assert np.abs(one - want[k]).max() <= TOL, (variant, flow, k)
  eng.close()

  def make_request(out_dir):
    return req_lib.request_from_text('''
      image { npy: "%s" }
      image_mean: 128
      image_stddev: 33
      seed_policy: "PolicyPeaks"
      model_checkpoint_path: "%s"
      model_name: "convstack_3d.ConvStack3DFFNModel"
      model_args: "{\\"depth\\": 18, \\"fov_size\\": [41, 41, 21], \\"deltas\\": [10, 10, 5]}"
      segmentation_output_dir: "%s"
      inference_options {
        init_activation: 0.95
        pad_value: 0.05
        move_threshold: 0.9
        min_boundary_dist { x: 1 y: 1 z: 1}
        segment_threshold: 0.6
        min_segment_size: 500
      }''' % (vol_path, weights, out_dir))

  sub, ov = (40, 100, 100), (21, 41, 41)
  runner = runner_lib.Runner()
  runner.start(make_request(str(tmp_path / 'sharded')), batch_size=4, direct=True)
  merged, info = ffn_dist.segment_volume(
      runner, (0, 0, 0), shape, sub, ov, batch_size=4, min_overlap_voxels=32,
      min_overlap_fraction=0.2)
  runner.stop_executor()
  boxes = info['boxes']
  assert len(boxes) == 8 and merged.shape == shape  # 2 x 2 x 2 anisotropic boxes
  assert all(b.size == sub for b in boxes)
  voxels = int((merged > 0).sum())
  assert voxels > 0.3 * merged.size, voxels  # the cells were committed

  # (1) standalone single-canvas runs (batch 1: conv32mt as the resident stack)
  solo = runner_lib.Runner()
  solo.start(make_request(str(tmp_path / 'solo')))
  assert solo.executor.engine.get_option('flow') == 2
  for box, seg in info['local_results']:
    canvas = solo.run(box.corner, box.size)
    want_seg = np.array(np.asarray(canvas.segmentation))
    want_seg[want_seg < 0] = 0
    assert np.array_equal(seg, want_seg), box.index
    assert len(np.unique(seg)) > 2, box.index  # several objects per sub-box
  assert solo.executor.engine.get_option('stat_flow_timeouts') == 0
  solo.stop_executor()

  # (2) assembly == specification; (3) objects merged across the cuts
  want, want_edges, want_roots = labels_oracle.reconcile(
      info['local_results'], shape, 32, 0.2)
  assert np.array_equal(merged, want)
  assert np.array_equal(info['edges'], want_edges)
  assert info['roots'] == want_roots
  assert len(want_edges) > 0
  print('c5 sharded %s: %d sub-boxes of %s, %d voxels labelled (%.0f %%), %d '
        'objects, %d merge edges' % (shape, len(boxes), sub, voxels,
                                     100.0 * voxels / merged.size,
                                     len(np.unique(merged)) - 1, len(want_edges)))
