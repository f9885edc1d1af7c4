#!/usr/bin/env python3
"""Headline benchmark: FoV-steps/sec of the flood-filling inference loop.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full FoV step of the hot path on a device-resident canvas:
gather -> conv0_a -> 23 x (3x3x3, 32->32) convs on the matrix pipe -> logit head ->
disco -> paste-back -> 6-face argmax, plus the move-queue bookkeeping that picks
the next position (Canvas.segment_all -> segment_at -> update_at; the segment
loop runs inside the library).

Workload (BASELINE.json configs[1]): single seed stream on ONE synthetic 250^3
uint8 volume per GPU, depth 12, FoV 33^3, deltas 8, FIB-25 weights
(tests/golden/fib25_weights.npz), options of configs/inference_training_sample2.
With N > 1 every rank owns an independent 250^3 volume (weak scaling, no
data-path collective; barrier-bracketed regions, MAX over ranks).

Rank 0 prints ONE JSON line:
  value         update_at-calls / wall clock of one COMPLETE segment_all pass over
                every rank's volume (`full_volume`; the metric's definition)
  steady_state  exactly K steps after W warm-up steps, inside running segments
                (the contract's timed region)
  roofline      the resident conv stack (conv32ps): algorithmic flops / HIP-event
                time of every timed step's launch against the fp16 MFMA peak / 3;
                HBM traffic and matrix-pipe occupancy from the committed PMC
                capture; the shader clock the box held
  cpu_baseline  (N == 1) the oracle port on this host's usable cores, a bounded
                sample of the same workload; its first steps replayed on the GPU
                (parity_*)
  batched       (N == 1) configs[2]: --mode sharded on a 512^3 volume, 32 canvases
"""

import argparse
import functools
import json
import os
import sys
import time

# Kernel arguments in device memory (read by the HIP runtime when it starts):
# the command processor then fetches them from HBM instead of over PCIe -- on a
# quiet, well-placed host it changes nothing measurable
# (profiles/r03_ab_hipgraph_conv_chain.txt), on a badly placed one every launch
# of a dependent chain pays for the fetch
# (profiles/r03_ab_dev_kernarg_slow_launch_box.txt).
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
if os.path.join(ROOT, 'tools') not in sys.path:
  sys.path.insert(1, os.path.join(ROOT, 'tools'))
# (run as a script this file is `__main__`; tools/bench_modes/* import it as `bench`)
sys.modules.setdefault('bench', sys.modules[__name__])

FOV = (33, 33, 33)
DELTAS = (8, 8, 8)
DEPTH = 12
FEATURES = 32
VOXELS = FOV[0] * FOV[1] * FOV[2]
# Algorithmic FLOPs of ONE conv32 launch at batch 1: dense SAME
# count 2 * 27 taps * 32 cin * 32 cout per voxel x 35,937 voxels.
CONV32_FLOPS = 2.0 * 27 * 32 * 32 * VOXELS
STEP_FLOPS = 2.0 * (2 * 27 * 32 + 23 * 27 * 32 * 32 + 32) * VOXELS
CONFIG = 'c1'
VOLUME_ZYX = (250, 250, 250)


def configure(args):
  """--config c1 (default): BASELINE configs[1].  c5: the anisotropic hi-res
  model of configs[4] -- depth 18, FoV zyx (21, 41, 41), deltas (5, 10, 10),
  68.4 GFLOP per FoV step -- with random weights (no checkpoint of that shape
  ships with the reference) on one (64, 192, 192) tile of its volume."""
  global FOV, DELTAS, DEPTH, VOXELS, CONV32_FLOPS, STEP_FLOPS, CONFIG, VOLUME_ZYX
  CONFIG = args.config
  if args.config == 'c5':
    FOV, DELTAS, DEPTH = (21, 41, 41), (5, 10, 10), 18
    # (small enough for the flood fill of the random-weights model to finish --
    # and commit -- segments inside a default run)
    VOLUME_ZYX = (64, 192, 192) if args.volume == 250 else (
        max(args.volume // 3, 64), args.volume, args.volume)
  else:
    VOLUME_ZYX = (args.volume,) * 3
  if args.volume_zyx:  # e.g. configs[4]'s stated canvas: 256 2048 2048
    VOLUME_ZYX = tuple(args.volume_zyx)
  VOXELS = FOV[0] * FOV[1] * FOV[2]
  CONV32_FLOPS = 2.0 * 27 * 32 * 32 * VOXELS
  STEP_FLOPS = 2.0 * (2 * 27 * 32 + (2 * DEPTH - 1) * 27 * 32 * 32 + 32) * VOXELS


def model_variables():
  """TF-named weight arrays: the reference's FIB-25 checkpoint (c1), or -- c5,
  for which the reference ships no checkpoint -- the constructed flood-fill
  network of `synthetic.flood_fill_weights`: the same architecture and cost per
  step, and floods that stay inside the cells of the phantom, so that segments
  end, commit and can be reconciled across sub-boxes."""
  if CONFIG == 'c1':
    with np.load(os.path.join(ROOT, 'tests', 'golden', 'fib25_weights.npz')) as d:
      return {k: d[k] for k in d.files}
  from ffn_amd import synthetic
  return synthetic.flood_fill_weights(DEPTH, FEATURES)


def dense_random_blob():
  """Engine weight blob of the same architecture with DENSE seeded random
  weights (normal, std 0.02).  The constructed c5 network is mostly zeros: the
  kernels do the same work on zeros, but the chip draws less power and holds a
  higher clock, so its kernel time flatters the roofline; the c5 kernel rates
  are therefore (also) taken with these weights loaded."""
  from ffn_amd.training.models import convstack_3d
  rng = np.random.RandomState(18)
  v = {}
  for name in convstack_3d.conv_scopes(DEPTH):
    cin = 2 if name == 'conv0_a' else FEATURES
    cout = 1 if name == 'conv_lom' else FEATURES
    k = 1 if name == 'conv_lom' else 3
    v['seed_update/%s/weights' % name] = rng.normal(
        0, 0.02, (k, k, k, cin, cout)).astype(np.float32)
    v['seed_update/%s/biases' % name] = np.zeros((cout,), np.float32)
  m = convstack_3d.ConvStack3DFFNModel(
      fov_size=list(FOV[::-1]), deltas=list(DELTAS[::-1]), batch_size=1,
      depth=DEPTH, features=FEATURES)
  m.set_variables(v)
  return m.weights_blob()


def bench_volume(shape, seed):
  """The cells phantom of a config (c5: thicker membranes -- its network follows
  26-connected bright voxels)."""
  from ffn_amd import synthetic
  if CONFIG == 'c5':
    return synthetic.cells_volume(shape, seed=seed, membrane_dilate=2)
  return synthetic.cells_volume(shape, seed=seed)


PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, v_mfma_f32_16x16x4_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense bf16 MFMA
# the split-product kernels (conv_variant >= 6): every f32 product = 3 fp16
# products on the 16-bit MFMA, so their ceiling in ALGORITHMIC (f32) flops is
# 1/3 of the dense 16-bit peak
SPLIT_PRODUCTS = 3


class _Done(Exception):
  pass


def _dist_env():
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  return rank, local_rank, world


def rank_devices(args, local_rank):
  """-> (index of the GPU this rank computes on, device its collectives run on).
  --ranks-share-gpus: rank r computes on GPU r % device_count (several ranks on one GPU:
  the launcher, the deal and the rank bookkeeping of an N > 1 job exercised on a one-GPU
  box -- not a scaling point); --collective-backend gloo: collectives on host tensors."""
  import torch
  dev_index = local_rank % torch.cuda.device_count() if args.ranks_share_gpus else local_rank
  coll = torch.device('cuda', dev_index) if args.collective_backend == 'nccl' else None
  return dev_index, coll


def init_group(args, dev_index):
  import torch
  import torch.distributed as dist
  if args.collective_backend == 'nccl':
    dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
  else:
    dist.init_process_group('gloo')


class Comm:
  """The ranks of the job as far as the bench talks to them: a barrier and
  small reductions over `torch.distributed` (backend nccl = RCCL on the GPUs;
  gloo in tests/test_bench_ranks.py, where tests/emulated_device.py stands in for
  the engine).  world == 1: no process group is needed or touched."""

  def __init__(self, rank, world, device=None):
    self.rank, self.world, self.device = rank, world, device

  def _tensor(self, values):
    import torch
    return torch.tensor([float(v) for v in values], dtype=torch.float64,
                        device=self.device if self.device is not None else 'cpu')

  def barrier(self):
    if self.world > 1:
      import torch.distributed as dist
      dist.barrier()

  def _reduce(self, values, op):
    if self.world == 1:
      return [float(v) for v in values]
    import torch.distributed as dist
    t = self._tensor(values)
    dist.all_reduce(t, op=op)
    return [float(v) for v in t.tolist()]

  def all_sum(self, values):
    import torch.distributed as dist
    return self._reduce(values, dist.ReduceOp.SUM)

  def all_max(self, values):
    import torch.distributed as dist
    return self._reduce(values, dist.ReduceOp.MAX)

  def all_gather(self, values):
    """-> one list of floats per rank."""
    if self.world == 1:
      return [[float(v) for v in values]]
    import torch
    import torch.distributed as dist
    t = self._tensor(values)
    out = [torch.zeros_like(t) for _ in range(self.world)]
    dist.all_gather(out, t)
    return [[float(v) for v in g.tolist()] for g in out]


def full_volume_totals(comm, steps, voxels, t_local, t_all, objects, volume_zyx):
  """The complete pass of every rank as ONE job: steps and voxels summed over
  the ranks, the clock of the slowest (t_all was taken behind a barrier)."""
  steps_all, voxels_all = comm.all_sum([steps, voxels])
  return {
      'what': 'one complete segment_all pass over each rank\'s %s volume (%d '
              'rank(s), summed): every grid seed, segment commits included; wall '
              'clock between barriers' % ('x'.join(str(v) for v in volume_zyx),
                                          comm.world),
      'steps': int(steps_all),
      'seconds': round(t_all, 4),
      'fov_steps_per_s': round(steps_all / t_all, 1),
      'voxels_segmented': int(voxels_all),
      'voxels_segmented_per_s': round(voxels_all / t_all, 1),
      'objects': objects,
      'rank0': {'steps': int(steps), 'voxels_segmented': int(voxels)},
      'rank0_seconds': round(t_local, 4),
  }


def stream_totals(comm, local):
  """Per-rank numbers of the stream mode -> the job's: the timed region ends
  with the slowest rank (MAX), steps and voxels add up (SUM)."""
  out = dict(local)
  out['elapsed'] = comm.all_max([local['elapsed']])[0]
  out['voxels_run'], out['steps_run'], out['voxels'] = comm.all_sum(
      [local['voxels_run'], local['steps_run'], local['voxels']])
  out['seconds_run'] = comm.all_max([local['seconds_run']])[0]
  # resident launches that timed out (several ranks on ONE GPU cannot all be resident)
  out['flow_voids'], out['flow_auto_off'] = comm.all_sum(
      [local.get('flow_voids', 0), local.get('flow_auto_off', 0)])
  return out


def make_request():
  from ffn_amd.inference import request as req_lib
  r = req_lib.InferenceRequest()
  r.image_mean = 128
  r.image_stddev = 33
  r.model_name = 'convstack_3d.ConvStack3DFFNModel'
  r.model_args = json.dumps({'depth': DEPTH, 'fov_size': list(FOV[::-1]),
                             'deltas': list(DELTAS[::-1])})
  o = r.inference_options
  o.init_activation = 0.95
  o.pad_value = 0.05
  o.move_threshold = 0.9
  o.segment_threshold = 0.6
  o.min_segment_size = 1000
  o.min_boundary_dist.x = 1
  o.min_boundary_dist.y = 1
  o.min_boundary_dist.z = 1
  return r


def _engine_options(args):
  """--engine-option name=value ... -> [(name, int)]: ffn_engine_set_option
  switches applied to every engine of the run (A/B runs; recorded in the JSON)."""
  out = []
  for item in args.engine_option or []:
    name, _, value = item.partition('=')
    out.append((name, int(value)))
  return out


def load_model():
  from ffn_amd.training.models import convstack_3d
  model = convstack_3d.ConvStack3DFFNModel(
      fov_size=list(FOV[::-1]), deltas=list(DELTAS[::-1]), batch_size=1,
      depth=DEPTH, features=FEATURES)
  model.set_variables(model_variables())
  return model


def full_fixture(workload_seed, forward='onednn'):
  """The reference-minted run of the WHOLE 250^3 phantom of this seed (tools/
  make_golden.py --only cells250 --forward onednn|f64c --num-seeds 0 [--volume-seed S])."""
  tag = '' if workload_seed == 1234 else '_s%d' % workload_seed
  return os.path.join(ROOT, 'tests', 'golden',
                      'ref_canvas_cells250_%s_full%s.npz' % (forward, tag))


def run_agreement(seg, seen, fixture_path, forward_name):
  """A finished GPU pass (labels `seg`, FoV positions `seen`) against a run the
  reference's own Canvas made of the same volume: the headline is the FOREGROUND IoU
  (labelled in both / labelled in either: what the segmentation covers), the id-for-id
  figure stands beside it (void as soon as the two runs number their objects apart)."""
  fixture = np.load(fixture_path)
  agree = segmentation_agreement(seg, fixture['segmentation'])
  ref_steps = [tuple(int(v) for v in p) for p in fixture['steps']]
  n = min(len(seen), len(ref_steps))
  first_bad = next((k for k in range(n) if seen[k] != ref_steps[k]), None)
  return {
      'fixture': '%s (the reference\'s Canvas behind %s)' % (
          os.path.relpath(fixture_path, ROOT), forward_name),
      'iou': agree['iou_foreground'],
      'iou_what': 'iou = iou_foreground: labelled in both / labelled in either; '
                  'iou_id_for_id: same id in both / labelled in either (ids must agree: '
                  'void once the runs count their objects apart); iou_best_match: per '
                  'reference object, size-weighted',
      'iou_foreground': agree['iou_foreground'],
      'iou_id_for_id': agree['iou_labelled'],
      'iou_best_match': agree['iou_best_match'],
      'objects': agree['objects'],
      'objects_matched_at_0999': agree['objects_matched_at_0999'],
      'reference_steps': len(ref_steps),
      'reference_objects': len(json.loads(str(fixture['origins']))),
      'first_position_mismatch': first_bad,
      'positions_compared': n,
  }


SHADER_CLOCK_SPREAD = {}  # batch -> min / median / max of sample_shader_clock's samples


class BoardSampler:
  """Socket power and shader clock of the GPU this process runs on, read from the
  driver's hwmon files (power1_input: microwatts of the package power tracker, freq1_input:
  sclk in Hz; what rocm-smi --showpower --showclocks prints) by a thread every few
  milliseconds WHILE a kernel loop runs: the figures that tell a power-capped clock from
  an idle one.  Read-only; everything is None where the files are not there."""

  def __init__(self, device_index=0, period_s=0.004):
    import glob
    import threading
    self.period_s = period_s
    self.hwmon = None
    self.cap_w = None
    self._stop = threading.Event()
    self._thread = None
    self.power_w, self.sclk_mhz = [], []
    try:
      import torch
      pr = torch.cuda.get_device_properties(device_index)
      tail = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
      for d in glob.glob('/sys/class/drm/card*/device'):
        if os.path.realpath(d).endswith(tail):
          hw = glob.glob(os.path.join(d, 'hwmon', 'hwmon*'))
          if hw and os.path.exists(os.path.join(hw[0], 'power1_input')):
            self.hwmon = hw[0]
      if self.hwmon:
        self.cap_w = int(open(os.path.join(self.hwmon, 'power1_cap')).read()) / 1e6
    except Exception:  # pylint:disable=broad-except
      self.hwmon = None

  def _read(self, name):
    with open(os.path.join(self.hwmon, name)) as f:
      return int(f.read())

  def _loop(self):
    while not self._stop.is_set():
      try:
        self.power_w.append(self._read('power1_input') / 1e6)
        self.sclk_mhz.append(self._read('freq1_input') / 1e6)
      except Exception:  # pylint:disable=broad-except
        pass
      self._stop.wait(self.period_s)

  def __enter__(self):
    import threading
    if self.hwmon:
      self._thread = threading.Thread(target=self._loop, daemon=True)
      self._thread.start()
    return self

  def __exit__(self, *exc):
    self._stop.set()
    if self._thread:
      self._thread.join()

  def summary(self, what):
    if not self.power_w:
      return None
    # (the first samples still show the idle state: the tracker averages over ~ms)
    pw = np.array(self.power_w[len(self.power_w) // 4:] or self.power_w)
    ck = np.array(self.sclk_mhz[len(self.sclk_mhz) // 4:] or self.sclk_mhz)
    return {'what': what, 'samples': int(len(pw)),
            'board_w': {'median': round(float(np.median(pw)), 1),
                        'max': round(float(pw.max()), 1)},
            'power_cap_w': self.cap_w,
            'sclk_mhz': {'min': round(float(ck.min()), 0),
                         'median': round(float(np.median(ck)), 0),
                         'max': round(float(ck.max()), 0)},
            'source': 'hwmon power1_input (package power tracker) / freq1_input (sclk) of this '
                      'GPU, polled every %d ms while the loop ran' % int(self.period_s * 1e3)}


def board_under_stack(eng, batch, seconds=1.2):
  """BoardSampler over ~`seconds` of resident conv stacks at this batch (timing only)."""
  try:
    eng.forward_resident(batch, 3)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.forward_resident(batch, 5)
    eng.synchronize()
    per = max((time.perf_counter() - t0) / 5, 1e-5)
    with BoardSampler() as bs:
      t_end = time.perf_counter() + seconds
      while time.perf_counter() < t_end:
        eng.forward_resident(batch, max(1, int(0.05 / per)))
        eng.synchronize()
    return bs.summary('%.1f s of back-to-back resident conv stacks, %d FoV(s) per launch'
                      % (seconds, batch))
  except Exception:  # pylint:disable=broad-except
    return None


def sample_shader_clock(eng, batch=1):
  """The shader clock the chip holds under this engine's conv stack, in GHz: the
  in-kernel cycle counter against the 100 MHz wall clock over one conv body of
  workgroup 0 (engine option debug_clock 1; a burst of resident stacks on whatever
  FoVs the engine last saw -- timing only).  The dense MFMA peak is quoted at the
  2.4 GHz boost clock (256 CUs x 4 SIMDs x 1,024 flop/clk); boxes of one pool hold
  1.85 - 2.1 GHz under a single FoV and ~1.45 GHz under batched steps, which is
  most of why identical code measures +- 6 % from box to box.  None on failure."""
  try:
    eng.set_option('debug_layer', 3)
    eng.set_option('debug_clock', 1)
    ghz = []
    for _ in range(7):  # (one conv body is ~6 us against a 10-ns clock: several samples)
      eng.forward_resident(batch, 6)
      eng.synchronize()
      c = eng.debug_clocks().astype(np.float64)
      ghz += [(c[w, 3] - c[w, 0]) / ((c[w, 5] - c[w, 4]) * 10.0)
              for w in range(4) if c[w, 5] > c[w, 4] and c[w, 3] > c[w, 0]]
    if not ghz:
      return None
    SHADER_CLOCK_SPREAD[batch] = {'samples': len(ghz), 'min': round(float(np.min(ghz)), 3),
                                  'median': round(float(np.median(ghz)), 3),
                                  'max': round(float(np.max(ghz)), 3)}
    return round(float(np.median(ghz)), 3)
  except Exception:  # pylint:disable=broad-except
    return None
  finally:
    try:
      eng.set_option('debug_clock', 0)
    except Exception:  # pylint:disable=broad-except
      pass


def segmentation_agreement(seg, want):
  """Three views of how two label volumes agree.  `iou_labelled`: voxels carrying
  the SAME id in both / voxels labelled in either -- the strictest, and void as
  soon as the two runs number their objects apart (one object more or less shifts
  every later id).  `iou_foreground`: labelled in both / labelled in either.
  `iou_best_match`: every reference object against the object of `seg` that covers
  most of it, weighted by the reference object's size (id-agnostic)."""
  seg = np.asarray(seg)
  want = np.asarray(want).astype(np.int32)
  both = (seg > 0) & (want > 0)
  union = int(np.sum((seg > 0) | (want > 0)))
  inter = int(np.sum(both & (seg == want)))
  size_w = np.bincount(want[want > 0].ravel())
  size_g = np.bincount(seg[seg > 0].ravel())
  keys, cnt = np.unique(want[both].astype(np.int64) * (1 << 32) +
                        seg[both].astype(np.int64), return_counts=True)
  best = {}
  for kk, c in zip(keys.tolist(), cnt.tolist()):
    w, gid = kk >> 32, kk & 0xffffffff
    iou = c / float(size_w[w] + size_g[gid] - c)
    if iou > best.get(w, 0.0):
      best[w] = iou
  ref_ids = np.nonzero(size_w)[0]
  matched = sum(best.get(int(w), 0.0) * size_w[w] for w in ref_ids) / max(
      float(size_w.sum()), 1.0)
  return {
      'iou_labelled': round(inter / max(union, 1), 6),
      'iou_foreground': round(int(both.sum()) / max(union, 1), 6),
      'iou_best_match': round(float(matched), 6),
      'objects': int(np.count_nonzero(size_g)),
      'reference_objects': int(len(ref_ids)),
      'objects_matched_at_0999': int(sum(1 for w in ref_ids
                                         if best.get(int(w), 0.0) >= 0.999)),
  }


def recorded_pass(make, policy):
  """One more complete pass (untimed) with the FoV positions kept."""
  from ffn_amd.inference import inference_utils
  canvas = make(inference_utils.Counters(), keep_history=True)
  seen = []
  inner = canvas._segment_at_native

  def recording(start_pos, *a, **kw):
    n = inner(start_pos, *a, **kw)
    if n:
      seen.extend(tuple(int(v) for v in p) for p in canvas.history[-n:])
    return n

  canvas._segment_at_native = recording
  canvas.segment_all(seed_policy=policy)
  again = np.array(np.asarray(canvas.segmentation))
  canvas.close()
  return seen, again


def full_volume_pass(args, comm, model, exe, request, image, barrier):
  """BASELINE.json's metric on a COMPLETE pass: `Canvas.segment_all` over every
  grid seed of this rank's volume (reference inference.py:538-683; 24 k FoV
  steps on the 250^3 phantom) -- seed set-up, validity tests, segment commits
  and all -- timed between barriers; then (rank 0, configs[1] only, untimed)
  the same pass once more with the FoV positions recorded, against the run the
  reference's own Canvas made of this volume behind the torch-CPU / oneDNN f32
  forward (tests/golden/ref_canvas_cells250_onednn_full.npz, tools/make_golden.py
  --only cells250 --forward onednn --num-seeds 0 --tag _full)."""
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  from ffn_amd.inference import seed as seed_lib
  eng = exe.engine
  fixture = None
  fixture_path = full_fixture(args.workload_seed)
  if (CONFIG == 'c1' and args.workload == 'cells' and
      tuple(VOLUME_ZYX) == (250, 250, 250) and os.path.exists(fixture_path)):
    fixture = np.load(fixture_path)
  if fixture is not None:  # the grid of the fixture = every valid seed position
    policy = functools.partial(seed_lib.PolicyFixed, coords=fixture['seeds'])
  else:
    policy = functools.partial(seed_lib.PolicyGrid3d, step=16,
                               offsets=(0, 8, 4, 12, 2, 10, 14))

  def make(counters, **kw):
    return inference.DeviceCanvas(
        model.info, exe.get_client(counters, direct=True), image,
        request.inference_options, counters=counters,
        movement_policy_fn=movement.get_policy_fn(request, model.info), **kw)

  mode = eng.get_option('profile_every')
  eng.set_profiling(0)
  counters = inference_utils.Counters()
  canvas = make(counters)
  eng.synchronize()
  barrier()
  t0 = time.perf_counter()
  canvas.segment_all(seed_policy=policy)
  eng.synchronize()
  t_local = time.perf_counter() - t0
  barrier()
  t_all = time.perf_counter() - t0
  steps = counters['update_at-calls'].value
  voxels = counters['voxels-segmented'].value
  out = full_volume_totals(comm, steps, voxels, t_local, t_all, len(canvas.origins),
                           VOLUME_ZYX)
  out['seeds_tried'] = int(len(fixture['seeds'])) if fixture is not None else None
  seg = np.array(np.asarray(canvas.segmentation))
  canvas.close()
  if fixture is not None and comm.rank == 0:
    seen, again = recorded_pass(make, policy)
    out['vs_reference_run'] = run_agreement(seg, seen, fixture_path,
                                            'the torch-CPU / oneDNN f32 forward')
    out['vs_reference_run']['repeat_pass_identical'] = bool(np.array_equal(again, seg))
    f64_path = full_fixture(args.workload_seed, 'f64')
    if os.path.exists(f64_path):
      out['vs_f64_run'] = run_agreement(seg, seen, f64_path, 'a double-precision forward')
  # the OTHER whole volume the suite holds a reference-minted run of (one more build of
  # a 250^3 phantom and two passes, ~15 s): rank 0, N = 1, untimed for `value`
  other = 4321 if args.workload_seed == 1234 else 1234
  if (fixture is not None and comm.rank == 0 and comm.world == 1 and
      not args.no_second_volume and os.path.exists(full_fixture(other))):
    from ffn_amd import synthetic
    image2 = synthetic.normalize(bench_volume(tuple(VOLUME_ZYX), other))
    fx2 = np.load(full_fixture(other))
    policy2 = functools.partial(seed_lib.PolicyFixed, coords=fx2['seeds'])
    make2 = lambda counters, **kw: inference.DeviceCanvas(  # noqa: E731
        model.info, exe.get_client(counters, direct=True), image2,
        request.inference_options, counters=counters,
        movement_policy_fn=movement.get_policy_fn(request, model.info), **kw)
    c2 = inference_utils.Counters()
    canvas = make2(c2)
    eng.synchronize()
    t0 = time.perf_counter()
    canvas.segment_all(seed_policy=policy2)
    eng.synchronize()
    t2 = time.perf_counter() - t0
    seg2 = np.array(np.asarray(canvas.segmentation))
    canvas.close()
    seen2, _ = recorded_pass(make2, policy2)
    rec = {'workload_seed': other, 'fov_steps': int(c2['update_at-calls'].value),
           'seconds': round(t2, 4),
           'fov_steps_per_s': round(c2['update_at-calls'].value / t2, 1),
           'vs_reference_run': run_agreement(seg2, seen2, full_fixture(other),
                                             'the torch-CPU / oneDNN f32 forward')}
    if os.path.exists(full_fixture(other, 'f64')):
      rec['vs_f64_run'] = run_agreement(seg2, seen2, full_fixture(other, 'f64'),
                                        'a double-precision forward')
    out['second_volume'] = rec
  eng.set_option('profile_every', mode)
  eng.set_profiling(args.profile_mode)
  return out


def run_gpu(args, rank, local_rank, world):
  import torch
  import torch.distributed as dist
  from ffn_amd import synthetic
  from ffn_amd.inference import executor
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  from ffn_amd.inference import seed as seed_lib

  if not torch.cuda.is_available():
    raise RuntimeError('bench.py needs an MI355X: no CPU fallback exists')
  dev_index, coll_device = rank_devices(args, local_rank)
  torch.cuda.set_device(dev_index)
  if world > 1:
    init_group(args, dev_index)
  comm = Comm(rank, world, coll_device)

  def barrier():
    comm.barrier()
    torch.cuda.synchronize()

  model = load_model()
  request = make_request()
  counters = inference_utils.Counters()
  exe = executor.HipBatchExecutor(executor.ExecutorInterface(), model,
                                  model.info, None, counters, 1,
                                  device_id=dev_index)
  eng = exe.engine
  if args.conv_variant is not None:
    eng.set_option('conv_variant', args.conv_variant)
  if args.sync_mode is not None:
    eng.set_option('sync_mode', args.sync_mode)
  for name, value in _engine_options(args):
    eng.set_option(name, value)

  shape = VOLUME_ZYX
  if args.workload == 'cells':
    vol = bench_volume(shape, args.workload_seed + rank)
  else:
    vol = synthetic.noise_volume(shape, seed=rank)
  image = synthetic.normalize(vol)

  total = args.warmup + args.steps
  state = {'n': 0, 't0': None, 't1': None, 'vox0': 0, 'vox1': 0,
           't_run0': None}

  class BenchCanvas(inference.DeviceCanvas):

    def update_at(self, pos):
      # untimed spin-up before the W warmup steps: host cores and GPU clocks
      # leave their idle states (run-to-run spread of `value` 3 % -> < 1 %)
      if state['t_run0'] is None:
        state['t_run0'] = time.perf_counter()
      if state['n'] == 0 and args.prewarm_seconds > 0:
        if state.get('pre_until') is None:
          state['pre_until'] = time.perf_counter() + args.prewarm_seconds
        if (time.perf_counter() < state['pre_until'] and
            state.get('prewarm_steps', 0) < args.prewarm_max_steps):
          state['prewarm_steps'] = state.get('prewarm_steps', 0) + 1
          return super().update_at(pos)
      if state['n'] == args.warmup:
        eng.synchronize()
        barrier()
        eng.get_profile(reset=True)
        state['vox0'] = self.counters['voxels-segmented'].value
        state['t0'] = time.perf_counter()
      pred = super().update_at(pos)
      state['n'] += 1
      if state['n'] == total:
        eng.synchronize()
        state['t1_local'] = time.perf_counter()
        barrier()
        state['t1'] = time.perf_counter()
        state['vox1'] = self.counters['voxels-segmented'].value
        raise _Done()
      return pred

  class NativeBenchCanvas(inference.DeviceCanvas):
    """The default: every segment's FoV loop runs inside the library
    (ffn_canvas_segment_at); the loop is called with step budgets that end
    exactly on the prewarm / warmup / timed boundaries and resumed."""

    def _segment_at_native(self, start_pos, max_steps=0, resume=False):
      del max_steps, resume
      done, first = 0, True
      if state['t_run0'] is None:
        state['t_run0'] = time.perf_counter()
      while True:
        if state['n'] == 0 and args.prewarm_seconds > 0 and not state.get(
            'prewarm_over'):
          if state.get('pre_until') is None:
            state['pre_until'] = time.perf_counter() + args.prewarm_seconds
          left = args.prewarm_max_steps - state.get('prewarm_steps', 0)
          if time.perf_counter() < state['pre_until'] and left > 0:
            n = super()._segment_at_native(start_pos, max_steps=min(left, 64),
                                           resume=not first)
            state['prewarm_steps'] = state.get('prewarm_steps', 0) + n
            done, first = done + n, False
            if not self._native_active:
              return done
            continue
          state['prewarm_over'] = True
        if state['n'] == args.warmup and state['t0'] is None:
          eng.synchronize()
          barrier()
          eng.get_profile(reset=True)
          state['vox0'] = self.counters['voxels-segmented'].value
          state['t0'] = time.perf_counter()
        boundary = args.warmup if state['n'] < args.warmup else total
        n = super()._segment_at_native(start_pos,
                                       max_steps=boundary - state['n'],
                                       resume=not first)
        state['n'] += n
        done, first = done + n, False
        if state['n'] == total:
          eng.synchronize()
          state['t1_local'] = time.perf_counter()
          barrier()
          state['t1'] = time.perf_counter()
          state['vox1'] = self.counters['voxels-segmented'].value
          raise _Done()
        if not self._native_active:
          return done

  canvas_cls = BenchCanvas if args.host_loop == 'python' else NativeBenchCanvas

  def new_canvas():
    return canvas_cls(model.info, exe.get_client(counters, direct=True), image,
                       request.inference_options, counters=counters,
                       movement_policy_fn=movement.get_policy_fn(
                           request, model.info))

  canvas = new_canvas()
  # HIP-event pairs around every conv32 launch of 1 FoV step in
  # `profile_every` (sampling keeps the event overhead out of `value`).
  # (a short timed region -- the driver's 20 steps -- is sampled whole)
  profile_every = 1 if args.steps <= 50 else args.profile_every
  state['profile_every'] = profile_every
  eng.set_option('profile_every', profile_every)
  eng.set_profiling(args.profile_mode)
  policy = functools.partial(seed_lib.PolicyGrid3d, step=16,
                             offsets=(0, 8, 4, 12, 2, 10, 14))
  # The 250^3 phantom holds ~80 k FoV steps; should K ask for more (or a small
  # --volume be used) the same volume is segmented again on a fresh canvas --
  # its set-up then sits inside the timed region and is reported.
  passes = 0
  try:
    while True:
      done_before = state['n'] + state.get('prewarm_steps', 0)
      canvas.segment_all(seed_policy=policy)
      passes += 1
      if state['n'] + state.get('prewarm_steps', 0) == done_before:
        raise RuntimeError('workload yields no FoV steps')
      canvas.close()
      canvas = new_canvas()
  except _Done:
    pass
  state['volume_passes_completed'] = passes

  conv_ms, conv_launches = eng.get_profile()
  prof_samples_ms = eng.get_profile_samples()
  flow = eng.get_option('flow')
  elapsed = state['t1'] - state['t0']
  elapsed_local = state['t1_local'] - state['t0']
  # Final segmentation merge (the ONLY collective of the path): the N per-rank
  # volumes are treated as N sub-boxes stacked along z of one virtual volume;
  # id offsets by all_gather, union by all_reduce(MAX) over RCCL.  Untimed
  # extra, outside the FoV-step region; reported as merge_ms.
  merge_ms = None
  merged_ids = None
  try:
    from ffn_amd import distributed as ffn_dist
    seg = np.array(canvas._handle.read_segmentation())
    seg[seg < 0] = 0
    full = (world * shape[0], shape[1], shape[2])
    boxes = ffn_dist.tile_volume(full, shape, (0, 0, 0))
    barrier()
    tm = time.perf_counter()
    merged, _ = ffn_dist.merge_segmentations(
        [(boxes[rank], seg)], full, rank, world,
        device=coll_device)
    barrier()
    merge_ms = (time.perf_counter() - tm) * 1e3
    merged_ids = int(merged.max())
    del merged
  except Exception as e:  # pylint:disable=broad-except
    print('merge skipped: %r' % (e,), file=sys.stderr)
  canvas._flush_hot()
  shader_ghz = sample_shader_clock(eng, 1)
  board = board_under_stack(eng, 1) if rank == 0 else None
  pace = {'ticks_10ns': eng.get_option('flow_pace_now'),
          'tuner_free_running_us': eng.get_option('flow_pace_free_ns') / 1e3,
          'tuner_at_its_beat_us': eng.get_option('flow_pace_best_ns') / 1e3,
          'what': 'the resident stack is PACED: conv l of a workgroup does not start before '
                  't0 + l x beat + beat x (its first voxel / V); the beat is measured when the '
                  'weights are set (a ladder of beats on noise inputs, the best one + 0.1 us '
                  'if it beats the free-running stack by > 1.5 %, else 0 = free-running); '
                  'timing only, bit-identical results (DESIGN.md section 3.3)'}
  cvals = {k: c.value for k, c in counters}
  cvals['gate_rejects'] = canvas.gate_rejects
  full_volume = None
  if not args.no_full_volume:
    full_volume = full_volume_pass(args, comm, model, exe, request, image, barrier)
  result = {
      'full_volume': full_volume,
      'shader_clock_ghz': shader_ghz,
      'board': board,
      'pace': pace,
      'turn_around': {
          'gpu_us': round(eng.get_option('stat_turn_gpu_ns') / 1e3, 2),
          'host_us': round(eng.get_option('stat_turn_host_ns') / 1e3, 2),
          'launch_call_us': round(eng.get_option('stat_launch_host_ns') / 1e3, 2),
          'steps': eng.get_option('stat_turn_count'),
          'segment_turn': {
              'calls': eng.get_option('stat_segturn_calls'),
              'queue_us': round(eng.get_option('stat_segturn_queue_ns') / 1e3, 1),
              'wait_us': round(eng.get_option('stat_segturn_wait_ns') / 1e3, 1),
              'what': 'ffn_canvas_segment_turn (commit, the next seeds tested, init_seed: one '
                      'call between two segments), host view: queueing its device sequence, '
                      'then waiting for its record',
          },
          'what': 'between two single-FoV steps, mean over the run: gpu_us = from the faces '
                  'block publishing a step\'s record (in-kernel wall clock) to the first '
                  'instruction of the NEXT resident launch; host_us = from the host seeing '
                  'that record to the next hipLaunchKernelGGL(conv32ps) having returned; '
                  'launch_call_us = inside that call alone.  Without stack_ahead gpu_us - '
                  'host_us = the record\'s trip over PCIe + doorbell -> first wave; with it '
                  '(the default) the next stack was queued before the record left and gpu_us '
                  'is what remains of the fused launch behind its faces block + one launch '
                  'boundary: the host\'s figures then describe work that runs UNDER the stack.',
      },
      'flow_voids': eng.get_option('stat_flow_voids'),
      'flow_auto_off': eng.get_option('flow_auto_off'),
      'merge_ms': merge_ms,
      'merged_ids': merged_ids,
      'counters': cvals,
      'conv_variant': (args.conv_variant if args.conv_variant is not None
                       else eng.get_option('conv_variant')),
      'prewarm_steps': state.get('prewarm_steps', 0),
      'speculation': {
          'conv0a_launched_ahead': eng.get_option('stat_spec_launched'),
          'steps_that_used_one': eng.get_option('stat_spec_hits'),
          'mismatches_repeated': eng.get_option('stat_spec_mismatch'),
          'launched_but_step_elsewhere_hint_list_full': eng.get_option('stat_spec_miss_full'),
          'launched_but_step_elsewhere_hint_list_short': eng.get_option('stat_spec_miss_short'),
          'stack_ahead': eng.get_option('stack_ahead'),
          'stacks_queued_ahead_and_used': eng.get_option('stat_ahead_used'),
          'stacks_queued_ahead_not_used': eng.get_option('stat_ahead_wasted'),
          'of_them_ended_after_the_first_conv': eng.get_option('stat_ahead_aborted'),
          'note': 'the whole run of this rank; single-FoV steps of the '
                  'library segment loop queue the next conv0_a behind the paste '
                  '(engine option speculate) and the resident stack of that step behind '
                  'it (stack_ahead): the host\'s turn-around leaves the critical path; a '
                  'stack whose conv0_a found no valid position ends after its first conv '
                  '(DESIGN.md section 4)',
      },
      'volume_passes_completed': state['volume_passes_completed'],
      'elapsed': elapsed,
      'elapsed_local': elapsed_local,
      'conv_ms': conv_ms,
      'conv_launches': conv_launches,
      'prof_samples_ms': prof_samples_ms,
      'flow': flow,
      'profile_every': state['profile_every'],
      'voxels': state['vox1'] - state['vox0'],
      # voxels leg: everything this rank did from its first FoV step (prewarm +
      # warmup + timed steps, segment set-up and commits included) on its own
      # wall clock -- the K timed steps alone rarely contain a segment commit
      'voxels_run': state['vox1'],
      'steps_run': state['n'] + state.get('prewarm_steps', 0),
      'seconds_run': state['t1_local'] - state['t_run0'],
      'segments': len(canvas.origins),
      'canvas': canvas,
      'exe': exe,
      'model': model,
      'request': request,
      'image': image,
  }
  result = stream_totals(comm, result)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()
  return result


def batched_leg(args):
  """BASELINE configs[2] next to the headline, time-boxed: this script once more
  with --mode sharded on a 512^3 volume -- 32 canvases open (two groups of 16
  FoVs per engine call; --batched-max-steps > 0 drops every canvas after that
  many FoV steps) -- in a process of its own (its engine holds 16 FoVs of activations;
  the headline's holds one).  Reported: the kernel-only rate of the batched
  conv stack with its roofline fraction, and the end-to-end FoV-steps/s of the
  bounded run.  Never `value`."""
  import subprocess
  cmd = [sys.executable, os.path.abspath(__file__), '--mode', 'sharded',
         '--sharded-volume', '512', '--sharded-sub', '153', '--sharded-batch', '16',
         '--sharded-groups', '2', '--sharded-max-steps',
         str(args.batched_max_steps), '--no-cpu-baseline']
  # (the assembled 512^3 labels are checked against the numpy specification
  # unless the run is bounded: dropped canvases leave nothing to compare)
  t0 = time.perf_counter()
  try:
    run = subprocess.run(cmd, capture_output=True, text=True,
                         timeout=args.batched_timeout)
    line = run.stdout.strip().splitlines()[-1]
    d = json.loads(line)
  except Exception as e:  # pylint:disable=broad-except
    return {'error': repr(e), 'seconds': round(time.perf_counter() - t0, 1)}
  return {
      'what': ('bench.py --mode sharded on one 512^3 volume in %d sub-boxes, 32 '
               'canvases open; %s' % (
                   d['config']['sub_boxes'],
                   'the whole volume' if not args.batched_max_steps else
                   'bounded: a sub-box is dropped after %d FoV steps'
                   % args.batched_max_steps)),
      'workload': d['config']['workload'],
      'value': d['value'],
      'unit': d['unit'],
      'steps': d['steps'],
      'seconds': d['segmentation_seconds'],
      'roofline': dict(d['batched_kernel'], board=d['roofline'].get('board')),
      'check_vs_specification': d['assembly']['check_vs_specification'],
      'voxels_segmented_per_s': d['voxels_segmented_per_s'],
      'assembly_ms': d['assembly']['merge_plus_reconcile_ms'],
      'engine_calls': d['engine_calls'],
      'host_loop': d['host_loop'],
      'setup_seconds': d['setup_seconds'],
      'wall_seconds_of_this_leg': round(time.perf_counter() - t0, 1),
  }


def c5_leg(args):
  """BASELINE configs[4]'s model (depth 18, FoV zyx 21 x 41 x 41, deltas 5 x 10 x 10) next
  to the headline, bounded: this script once more with --mode sharded --config c5 on a
  96 x 512 x 512 volume (32 anisotropic sub-boxes, 16 canvases open) in a process of its
  own.  The run uses the constructed flood-fill network (no checkpoint of that shape
  exists) and is checked against the numpy specification; the kernel `roofline` is
  timed on dense random weights (68.44 GFLOP per FoV step against 2,500 / 3 TFLOP/s).
  Never `value`."""
  import subprocess
  cmd = [sys.executable, os.path.abspath(__file__), '--mode', 'sharded', '--config', 'c5',
         '--sharded-volume-zyx', '96', '512', '512', '--sharded-sub-zyx', '64', '192', '192',
         '--sharded-batch', '8', '--sharded-groups', '2', '--no-cpu-baseline']
  t0 = time.perf_counter()
  try:
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=args.c5_timeout)
    d = json.loads(run.stdout.strip().splitlines()[-1])
  except Exception as e:  # pylint:disable=broad-except
    return {'error': repr(e), 'seconds': round(time.perf_counter() - t0, 1)}
  return {
      'what': 'bench.py --mode sharded --config c5 on one 96x512x512 volume in %d '
              'sub-boxes of 64x192x192, 16 canvases open' % d['config']['sub_boxes'],
      'workload': d['config']['workload'],
      'conv_variant': d['config']['conv_variant'],
      'value': d['value'],
      'unit': d['unit'],
      'steps': d['steps'],
      'seconds': d['segmentation_seconds'],
      'roofline': d['roofline'],
      'whole_run': d['batched_kernel']['whole_run'],
      'check_vs_specification': d['assembly']['check_vs_specification'],
      'voxels_segmented_per_s': d['voxels_segmented_per_s'],
      'assembly_ms': d['assembly']['merge_plus_reconcile_ms'],
      'wall_seconds_of_this_leg': round(time.perf_counter() - t0, 1),
  }


def _self_launch(args):
  """`python bench.py --gpus N` outside torch.distributed.run: become
  `python -m torch.distributed.run --nproc-per-node N bench.py ...` (one rank
  per GPU over RCCL); rank 0 of that job prints the one JSON line."""
  import socket
  with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
         '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  sys.stdout.flush()
  os.execve(sys.executable, cmd, env)


def stream_line(args, world, res):
  """The one JSON line of the stream mode from the job's totals (`stream_totals`,
  `full_volume_totals`) and rank 0's kernel timing: pure bookkeeping, no device
  (tests/test_bench_ranks.py runs it under gloo with world 2 and 4)."""
  steps_per_s = world * args.steps / res['elapsed']
  n_convs = 2 * DEPTH - 1
  avg_conv_ms = res['conv_ms'] / max(res['conv_launches'], 1)
  # one sample = one HIP-event pair (on the engine's stream) around the convs of
  # a step: the ONE launch of the resident stack (engine option flow 2), or the
  # chain of 2 depth - 1 dependent launches (then with their gaps), or -- mode 1
  # -- a single conv launch
  samples_ms = np.asarray(res['prof_samples_ms'], np.float64)
  per_sample_convs = n_convs if args.profile_mode == 2 else 1
  resident = res.get('flow') == 2 and res.get('conv_variant') == 9
  if len(samples_ms):
    med_ms, min_ms, mean_ms = (float(np.median(samples_ms)), float(samples_ms.min()),
                               float(samples_ms.mean()))
  else:
    med_ms = min_ms = mean_ms = avg_conv_ms * per_sample_convs
  # achieved = algorithmic flops of the sampled launches / their MEAN duration
  # (what rocprofv3 --stats reports as the kernel's average; median and minimum
  # next to it)
  achieved = (per_sample_convs * CONV32_FLOPS / (mean_ms * 1e-3) / 1e12
              if mean_ms else 0.0)
  # HBM traffic of the conv kernel cannot be counted from inside this process
  # (PMC counters need rocprofv3): it is taken from the committed PMC profile of
  # this same command, when present.
  traffic = None
  traffic_source = None
  traffic_stale = None
  pmc_commit = None
  pmc_busy_cycles = None
  try:
    resident = res.get('flow') == 2 and res.get('conv_variant') == 9
    tname = (('r06_conv32ps_pmc.json' if resident else
              'conv32_pmc_traffic.json') if CONFIG == 'c1' else
             'r03_conv32mt_c5_pmc_traffic.json')
    if not os.path.exists(os.path.join(ROOT, 'profiles', tname)) and resident:
      tname = 'r05_conv32ps_pmc.json'
    with open(os.path.join(ROOT, 'profiles', tname)) as f:
      tj = json.load(f)
    if (tj.get('conv_variant', 9) == res.get('conv_variant', 9) and
        not (CONFIG != 'c1' and resident)):  # (c5's capture is of the per-conv launches)
      traffic = tj['traffic_bytes_per_launch']
      traffic_source = ('profiles/%s: separate rocprofv3 --pmc FETCH_SIZE / '
                        'WRITE_SIZE passes of this command on the tree at commit %s '
                        '(NOT measured in this run)' % (tname, tj.get('commit') or
                                                        '(not recorded)'))
      pmc_commit = tj.get('commit')
      from ffn_amd import _lib as lib_mod
      pmc_sha = tj.get('csrc_sha')
      traffic_stale = pmc_sha != lib_mod.csrc_sha()
      busy = (tj.get('sq_per_launch') or {}).get('SQ_VALU_MFMA_BUSY_CYCLES')
      if busy:
        pmc_busy_cycles = float(busy)
  except (OSError, KeyError, ValueError):
    pass
  variant = res.get('conv_variant', 9)
  PEAK_F16_MFMA_TFLOPS = PEAK_BF16_MFMA_TFLOPS  # same dense rate on gfx950
  if variant >= 6:
    products = SPLIT_PRODUCTS
    mfma = 'v_mfma_f32_32x32x16_f16'
    shape = (('conv32ps: the %d convs of a step as ONE resident launch -- conv32mt\'s '
              'workgroups keep their voxels through the stack and hand rows to each '
              'other through per-producer sequence words instead of kernel '
              'boundaries; each conv = ' % n_convs if resident else '') +
             'conv32mt (3x3x3 32->32 implicit GEMM on producer-split fp16 planes '
             'staged by LDS-DMA: 256 conv32m workgroups of 128 voxels, one per '
             'CU -- one 32-position tile per wave for all 27 taps, weights '
             'through an LDS-DMA ring -- and the remaining voxels in 32-voxel '
             'tail workgroups whose taps are split over the four waves'
             if variant == 9 else
             'conv32m (3x3x3 32->32 implicit GEMM on producer-split fp16 planes '
             'staged by LDS-DMA, 4-wave workgroups of 128 voxels, one 32-position '
             'tile per wave for all 27 taps, weights through an LDS-DMA ring, two '
             'workgroups per CU' if variant == 8 else
             'conv32d (3x3x3 32->32 implicit GEMM on producer-split fp16 planes '
             'staged by LDS-DMA, 4-wave workgroups, the 27 taps split over the '
             'waves')
    kernel_name = (
        '%s; f32 operands split into fp16 hi + 2^-11-scaled fp16 residual (22 '
        'mantissa bits), %d products per f32 product on %s, f32 accumulation)' %
        (shape, products, mfma))
    peak = PEAK_F16_MFMA_TFLOPS / products
    peak_basis = ('dense 16-bit MFMA peak %.0f TFLOP/s / %d products per '
                  'algorithmic f32 product' % (PEAK_F16_MFMA_TFLOPS, products))
    executed_ratio = products
    dtype = ('f32 (fp16 hi + scaled-residual split products on the fp16 MFMA, '
             'f32 accumulate)')
  else:
    kernel_name = ('conv32 (3x3x3 32->32 implicit GEMM, '
                   'v_mfma_f32_16x16x4_f32)')
    peak = PEAK_F32_MFMA_TFLOPS
    peak_basis = 'dense f32 MFMA peak'
    executed_ratio = 1
    dtype = 'f32'
  # The metric: delta(update_at-calls) / wall(segment_all) --
  # the COMPLETE pass over the volume(s), seed set-up, validity tests and segment
  # commits included (`full_volume`).  The K timed steps of the contract, all
  # inside one segment, are the steady-state rate next to it.  Without the
  # complete pass (--no-full-volume, c5) `value` falls back to the steady state.
  fv = res.get('full_volume')
  value = fv['fov_steps_per_s'] if fv else round(steps_per_s, 2)
  out = {
      'metric': ('FoV-steps/sec (flood-filling inference loop, 250^3 volume)'
                 if CONFIG == 'c1' else
                 'FoV-steps/sec (flood-filling inference loop, configs[4] model)'),
      'value': value,
      'value_is': ('update_at-calls / wall clock of one complete segment_all pass over '
                   'every rank\'s volume (full_volume: %d FoV steps in %.3f s)'
                   % (fv['steps'], fv['seconds']) if fv else
                   'the K timed steps (steady_state): no complete pass was run'),
      'unit': 'FoV-steps/s',
      'n_gpus': world,
      'steps': args.steps,
      'warmup': args.warmup,
      'prewarm_steps_untimed': res.get('prewarm_steps', 0),
      'volume_passes_completed': res.get('volume_passes_completed', 0),
      'ms_per_step': round(1e3 * world / value, 4) if fv else round(
          1e3 * res['elapsed'] / args.steps, 4),
      'steady_state': {
          'what': 'exactly K = %d FoV steps after %d warm-up steps, all inside running '
                  'segments, between barriers, max over ranks' % (args.steps, args.warmup),
          'value': round(steps_per_s, 2),
          'ms_per_step': round(1e3 * res['elapsed'] / args.steps, 4),
          'seconds': round(res['elapsed'], 6),
      },
      'higher_is_better': True,
      'scaling': 'weak',
      'vs_baseline': None,
      'dtype': dtype,
      'data': 'synthetic',
      'config': {
          'workload': (
              'configs[1] single-seed single-GPU: depth=12 fov=33^3 '
              'deltas=8, synthetic %s %d^3 uint8 volume per GPU (seed %d), '
              'FIB-25 weights, device-resident canvas, batch 1'
              % (args.workload, args.volume, args.workload_seed) if CONFIG == 'c1' else
              'configs[4] model on one tile: depth=18 fov zyx %s deltas %s, '
              'synthetic %s %s uint8 volume per GPU, constructed flood-fill weights '
              '(synthetic.flood_fill_weights: no checkpoint of this shape exists), '
              'device-resident canvas, batch 1 (NOT the headline config)'
              % (list(FOV), list(DELTAS), args.workload, list(VOLUME_ZYX))),
          'volume': list(VOLUME_ZYX),
          'parallelism': 'independent volume per rank (no data-path collective)',
          'host_loop': ('ffn_canvas_segment_at (segment loop inside the library)'
                        if args.host_loop == 'native' else
                        'Python, one ffn_canvas_step per FoV step'),
          'engine_options': dict(_engine_options(args)),
          'env': {k: os.environ[k] for k in ('HIP_FORCE_DEV_KERNARG',)
                  if k in os.environ},
          'launch': {'collective_backend': args.collective_backend,
                     'ranks_share_gpus': bool(args.ranks_share_gpus)},
      },
      'resident_launch_fallbacks': {
          'steps_repeated_per_layer': int(res.get('flow_voids', 0)),
          'ranks_that_turned_the_resident_launch_off': int(res.get('flow_auto_off', 0)),
          'note': 'all ranks; 0 on a GPU of its own -- ranks that SHARE a GPU cannot all keep '
                  'their 356 workgroups resident: FFN_ERR_FLOW, the step repeated with '
                  'per-layer launches (same bits), three in a row turn the resident launch '
                  'off for that engine',
      },
      # the second half of BASELINE.json's metric: from the complete pass below
      # (`full_volume`) when it ran, else from the partial run (`voxels_leg`)
      'voxels_segmented_per_s': (
          res['full_volume']['voxels_segmented_per_s'] if res.get('full_volume')
          else round(res['voxels_run'] / max(res['seconds_run'], 1e-9), 1)),
      'full_volume': res.get('full_volume'),
      'voxels_leg': {
          'voxels_segmented': int(res['voxels_run']),
          'fov_steps': int(res['steps_run']),
          'seconds': round(res['seconds_run'], 4),
          'region': 'all ranks, from the first FoV step of the run to the end '
                    'of the timed region (prewarm + warmup + timed steps, '
                    'seed set-up and segment commits included); the K timed '
                    'steps alone held %d voxels' % int(res['voxels']),
      },
      'turn_around_us': res.get('turn_around'),
      'host_breakdown_us_per_step': {
          'c_abi_step_call': round(1e3 * res['counters'].get(
              'inference-time-ms', 0) / max(res['counters'].get(
                  'inference-calls', 1), 1), 1),
          'update_at': round(1e3 * res['counters'].get(
              'update_at-time-ms', 0) / max(res['counters'].get(
                  'update_at-calls', 1), 1), 1),
          'movement_policy': round(1e3 * res['counters'].get(
              'movement_policy-time-ms', 0) / max(res['counters'].get(
                  'movement_policy-calls', 1), 1), 1),
      },
      'speculation': res['speculation'],
      'queue_stats': {k: res['counters'].get(k, 0) for k in (
          'update_at-calls', 'skip_threshold', 'skip_invalid_pos',
          'seed_got_too_weak', 'segment_at-loop-calls', 'gate_rejects')},
      'segmentation_merge': {
          'ms': None if res['merge_ms'] is None else round(res['merge_ms'], 2),
          'global_ids': res['merged_ids'],
          'how': 'all_gather(id offsets) + all_reduce(MAX) of the int32 label '
                 'volume over RCCL (no-op collective at n_gpus = 1); untimed',
      },
      'step_gflop': round(STEP_FLOPS / 1e9, 3),
      'end_to_end_tflops': round(value * STEP_FLOPS / 1e12, 3),
      'roofline': {
          'bound': 'mfma',
          'kernel': kernel_name,
          'achieved': round(achieved, 3),
          'peak': round(peak, 1),
          'unit': 'TFLOP/s',
          'frac': round(achieved / peak, 4),
          'peak_basis': peak_basis,
          'executed_mfma_tflops': round(achieved * executed_ratio, 1),
          'vs_native_f32_mfma_peak': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
          'traffic': traffic,
          'traffic_source': traffic_source,
          'traffic_commit': pmc_commit,
          # True: ffn_amd/csrc has changed since that capture (its csrc_sha differs from
          # this tree's): the figure describes an EARLIER kernel
          'traffic_stale': traffic_stale,
          # matrix-pipe occupancy: SQ_VALU_MFMA_BUSY_CYCLES of one launch (summed
          # over the chip's 1,024 SIMDs, from the same committed PMC capture) over
          # the SIMD-cycles of THIS run's mean launch at the 2.1 GHz the chip holds
          # under a single FoV
          'mfma_busy': (round(pmc_busy_cycles / (1024 * mean_ms * 1e-3 * 2.1e9), 4)
                        if pmc_busy_cycles and mean_ms and resident else None),
          'mfma_busy_cycles_per_launch': pmc_busy_cycles,
          # the clock this box held under the kernel (sample_shader_clock): the peak
          # above is at the 2.4 GHz boost clock
          'shader_clock_ghz': res.get('shader_clock_ghz'),
          'shader_clock_samples': SHADER_CLOCK_SPREAD.get(1),
          'board': res.get('board'),
          'pace': res.get('pace'),
          'frac_of_peak_at_that_clock': (
              round(achieved / (peak * res['shader_clock_ghz'] / 2.4), 4)
              if res.get('shader_clock_ghz') else None),
          'launch_kernel': ('conv32ps_kernel' if resident else
                            'conv32mt_kernel' if variant == 9 else 'conv32*_kernel'),
          'launches_per_step': 1 if resident else n_convs,
          'convs_per_launch': n_convs if resident else 1,
          'flops_per_launch': (n_convs if resident else 1) * CONV32_FLOPS,
          'avg_launch_us': round(
              mean_ms * 1e3 / (1 if resident else per_sample_convs), 3),
          'median_launch_us': round(
              med_ms * 1e3 / (1 if resident else per_sample_convs), 3),
          'min_launch_us': round(
              min_ms * 1e3 / (1 if resident else per_sample_convs), 3),
          'us_per_conv': round(mean_ms * 1e3 / per_sample_convs, 3),
          'frac_at_median': round(
              per_sample_convs * CONV32_FLOPS / (med_ms * 1e-3) / 1e12 / peak, 4)
              if med_ms else None,
          'frac_at_min': round(
              per_sample_convs * CONV32_FLOPS / (min_ms * 1e-3) / 1e12 / peak, 4)
              if min_ms else None,
          'samples': int(len(samples_ms)),
          'timing': ('HIP event pair on the engine\'s stream around the conv '
                     'launch(es) of %s step of the timed region: %s'
                     % ('every' if res.get('profile_every', 1) == 1 else
                        'every %dth' % res['profile_every'],
                        'the one resident launch' if resident else
                        'the chain of %d dependent launches, gaps included, / %d'
                        % (n_convs, n_convs) if args.profile_mode == 2 else
                        'each launch by itself')),
          'whole_step': {
              'what': 'every flop of the step (conv0_a, the %d convs, the head) '
                      'over the end-to-end time per step (host turn-around, '
                      'conv0_a, faces + paste included)' % n_convs,
              'tflops': round(value / world * STEP_FLOPS / 1e12, 3),
              'frac': round(value / world * STEP_FLOPS / 1e12 / peak, 4),
          },
      },
  }
  return out


# the modes behind a leg or a flag live in tools/bench_modes/ (they see this module as B)
from bench_modes.cpu import (_cpu_run, cpu_baseline, cpu_worker, gpu_parity_leg,  # noqa: E402,F401
                             usable_cpus)
from bench_modes.sharded import (ffn_dist_min_overlap, run_sharded,  # noqa: E402,F401
                                 sampled_assembly_check, sharded_line, sharded_totals)


def build_parser():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--collective-backend', choices=['nccl', 'gloo'], default='nccl',
                  help='nccl = RCCL on the GPUs (default); gloo = collectives on host tensors')
  ap.add_argument('--ranks-share-gpus', action='store_true',
                  help='rank r computes on GPU r %% device_count (N ranks on fewer GPUs)')
  ap.add_argument('--steps', type=int, default=1500)
  ap.add_argument('--warmup', type=int, default=100)
  ap.add_argument('--prewarm-max-steps', type=int, default=2500)
  ap.add_argument('--prewarm-seconds', type=float, default=1.0,
                  help='untimed spin-up (extra FoV steps) before the warmup '
                  'steps are counted')
  ap.add_argument('--volume', type=int, default=250)
  ap.add_argument('--volume-zyx', type=int, nargs=3, default=None,
                  help='canvas size zyx (overrides --volume)')
  ap.add_argument('--mode', choices=['stream', 'sharded'], default='stream',
                  help='stream: the headline (one seed stream per GPU); sharded: '
                  'one volume tiled into sub-boxes, timed assembly (configs[3])')
  ap.add_argument('--sharded-volume', type=int, default=320)
  ap.add_argument('--sharded-volume-zyx', type=int, nargs=3, default=None,
                  help='a non-cubic volume (configs[4]: 256 2048 2048)')
  ap.add_argument('--sharded-sub', type=int, default=176)
  ap.add_argument('--sharded-sub-zyx', type=int, nargs=3, default=None,
                  help='sub-box size zyx (default: --sharded-sub cubed)')
  ap.add_argument('--sharded-batch', type=int, default=8,
                  help='FoVs per engine call (= canvases per group)')
  ap.add_argument('--sharded-groups', type=int, default=2,
                  help='canvas groups, each with its own host thread and engine '
                  'calls (canvases open = groups x batch)')
  ap.add_argument('--sharded-deal', choices=['dynamic', 'static'],
                  default='dynamic')
  ap.add_argument('--sharded-collective', choices=['all_reduce', 'broadcast'],
                  default='broadcast',
                  help='assembly of the label volume over RCCL: every sub-box core '
                  'broadcast once by its owner (default), or north_star\'s '
                  'all_reduce(MAX) of a zero-filled volume (same result, 2 N x the '
                  'bytes sent per GPU: assembly.collective_bytes)')
  ap.add_argument('--sharded-max-steps', type=int, default=0,
                  help='bound the run: a canvas is dropped after this many FoV '
                  'steps (0 = segment everything)')
  ap.add_argument('--config', choices=['c1', 'c5'], default='c1',
                  help='c1: BASELINE configs[1] (the headline); c5: the depth-18 '
                  'anisotropic model of configs[4], random weights')
  ap.add_argument('--workload', choices=['cells', 'noise'], default='cells')
  ap.add_argument('--workload-seed', type=int, default=1234,
                  help='seed of the synthetic cells phantom (rank r: seed + r); 1234 '
                  'and 4321 have a reference-minted whole-volume run to compare with')
  ap.add_argument('--conv-variant', type=int, default=None)
  ap.add_argument('--engine-option', action='append', default=[],
                  metavar='NAME=VALUE', help='ffn_engine_set_option switch '
                  '(e.g. use_graph=1); may be given several times')
  ap.add_argument('--profile-every', type=int, default=8)
  ap.add_argument('--sync-mode', type=int, default=None)
  ap.add_argument('--profile-mode', type=int, default=2,
                  help='1 = event pair per conv launch, 2 = per 23-conv chain')
  ap.add_argument('--cpu-seconds', type=float, default=24.0,
                  help='cpu_baseline: seconds of CPU work for the samples (split '
                  'over the implementations timed)')
  ap.add_argument('--cpu-steps', type=int, default=100000)
  ap.add_argument('--cpu-probe-steps', type=int, default=20)
  ap.add_argument('--cpu-parity-steps', type=int, default=60,
                  help='steps of the CPU runs the GPU replays (gpu_parity_leg)')
  ap.add_argument('--cpu-box-seconds', type=float, default=10.0)
  ap.add_argument('--no-cpu-whole-box', action='store_true')
  ap.add_argument('--cpu-worker', type=int, default=None, help=argparse.SUPPRESS)
  ap.add_argument('--cpu-workers', type=int, default=1, help=argparse.SUPPRESS)
  ap.add_argument('--cpu-threads', type=int, default=1, help=argparse.SUPPRESS)
  ap.add_argument('--cpu-impl', default='c_oracle', help=argparse.SUPPRESS)
  ap.add_argument('--cpu-image', default='', help=argparse.SUPPRESS)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-full-volume', action='store_true',
                  help='skip the complete segment_all pass (and its comparison '
                  'with the reference-minted run) behind the K timed steps')
  ap.add_argument('--assembly-check-stride', type=int, default=1,
                  help='--mode sharded: 1 = the whole assembled volume against the '
                  'numpy specification; k > 1 = every k-th sub-box (its margin\'s '
                  'merge edges and its core\'s final labels) + every id offset -- for '
                  'volumes of a billion voxels')
  ap.add_argument('--no-assembly-check', action='store_true',
                  help='--mode sharded: skip the (untimed) comparison of the '
                  'assembled volume with the numpy specification')
  ap.add_argument('--no-batched-leg', action='store_true',
                  help='skip the time-boxed batched (configs[2]) leg of the '
                  'default run')
  ap.add_argument('--no-second-volume', action='store_true',
                  help='skip the pass over the other reference-minted 250^3 volume')
  ap.add_argument('--no-c5-leg', action='store_true',
                  help='skip the configs[4] leg (depth 18, FoV 21x41x41) of the default line')
  ap.add_argument('--c5-timeout', type=float, default=240.0)
  ap.add_argument('--batched-max-steps', type=int, default=0)
  ap.add_argument('--batched-timeout', type=float, default=420.0)
  ap.add_argument('--host-loop', choices=['native', 'python'], default='native',
                  help='native: ffn_canvas_segment_at runs each segment\'s FoV '
                  'loop inside the library; python: one ffn_canvas_step call '
                  'per step from the interpreter')
  return ap


def main():
  args = build_parser().parse_args()
  configure(args)
  if args.cpu_worker is not None:
    cpu_worker(args)
    return

  rank, local_rank, world = _dist_env()
  if world != args.gpus:
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
      _self_launch(args)  # does not return
    raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.'
                     'distributed.run --nproc-per-node %d' %
                     (args.gpus, world, args.gpus))

  if args.mode == 'sharded':
    run_sharded(args, rank, local_rank, world)
    return
  res = run_gpu(args, rank, local_rank, world)
  if rank != 0:
    return

  out = stream_line(args, world, res)
  if world == 1 and not args.no_cpu_baseline:
    oracle_trace, out['cpu_baseline'] = cpu_baseline(args)
    try:
      out.update(gpu_parity_leg(res, oracle_trace))
    except Exception as e:  # pylint:disable=broad-except
      out.update({'parity_steps_checked': 0, 'parity_ok': False,
                  'parity_error': repr(e)})
  if world == 1 and CONFIG == 'c1' and not args.no_batched_leg:
    out['batched'] = batched_leg(args)
  if world == 1 and CONFIG == 'c1' and not args.no_c5_leg:
    out['c5'] = c5_leg(args)
  print(json.dumps(out))


if __name__ == '__main__':
  main()
