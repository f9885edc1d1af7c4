"""bench.py's cpu_baseline leg: the oracle's canvas loop timed on the host cores (ORACLE = the checker and
the reported CPU baseline, never the product), and the replay of its first steps on the GPU."""
import functools
import json
import os
import sys
import time

import numpy as np

import bench as B


def _cpu_run(image, blob, variables, seeds, impl, threads, budget_s, max_steps,
             keep_trace=False):
  """The oracle's canvas loop (oracle/ffn_oracle.py: the reference's
  Canvas.segment_all restated) behind one CPU conv stack -- 'c_oracle' (plain C,
  OpenMP) or 'torch_onednn' -- on `threads` threads, until `max_steps` FoV steps
  or `budget_s` seconds; the first step is warm-up and not counted.
  -> (steps per second, steps, seconds, trace)."""
  from oracle import ffn_oracle
  forward_fn = None
  if impl == 'torch_onednn':
    forward_fn = functools.partial(ffn_oracle.forward_torch, variables=variables,
                                   depth=B.DEPTH, threads=threads)
  else:
    ffn_oracle.set_threads(threads)

  class _Stop(Exception):
    pass

  oc = ffn_oracle.OracleCanvas(image, blob, B.DEPTH, B.FOV, B.DELTAS, ffn_oracle.Options())
  oc.forward_fn = forward_fn
  t0 = [None]
  inner = oc.update_at
  n = [0]

  def timed_update(pos):
    if n[0] == 1:  # first step = warm-up (thread spin-up, page faults)
      t0[0] = time.perf_counter()
    out = inner(pos)
    n[0] += 1
    if n[0] > 1 and (n[0] - 1 >= max_steps or
                     time.perf_counter() - t0[0] > budget_s):
      raise _Stop()
    return out

  oc.update_at = timed_update
  try:
    oc.segment_all(seeds)
  except _Stop:
    pass
  steps = n[0] - 1
  dt = time.perf_counter() - t0[0]
  return steps / dt, steps, dt, (list(oc.trace) if keep_trace else None)


def cpu_worker(args):
  """`bench.py --cpu-worker K`: one of the P concurrent oracle processes of the
  whole-box CPU figure (cpu_baseline).  Its own canvas over the same volume, its
  own part of the seed grid; prints {"steps", "seconds"}."""
  k, p = args.cpu_worker, args.cpu_workers
  # this worker's own cores (the thread pools of P processes would otherwise all
  # start on the same first cores)
  try:
    cores = sorted(os.sched_getaffinity(0))
    mine = cores[k * args.cpu_threads:(k + 1) * args.cpu_threads]
    if len(mine) == args.cpu_threads:
      os.sched_setaffinity(0, mine)
  except (AttributeError, OSError):
    pass
  from oracle import ffn_oracle
  variables = B.model_variables()
  blob = ffn_oracle.weights_blob(variables, B.DEPTH)
  image = np.load(args.cpu_image, mmap_mode='r')
  seeds = ffn_oracle.grid_seeds(B.VOLUME_ZYX, tuple(f // 2 for f in B.FOV))
  first = (len(seeds) * k) // p
  seeds = np.concatenate([seeds[first:], seeds[:first]])
  _, steps, dt, _ = _cpu_run(np.asarray(image), blob, variables, seeds, args.cpu_impl,
                             args.cpu_threads, args.cpu_seconds, 10 ** 9)
  print(json.dumps({'steps': steps, 'seconds': dt}))


def usable_cpus():
  """Cores this process may actually run on: the affinity mask and the cgroup's
  CPU quota, not os.cpu_count() (a container on a 256-thread host may own 16)."""
  n = os.cpu_count() or 1
  try:
    n = min(n, len(os.sched_getaffinity(0)))
  except (AttributeError, OSError):
    pass
  for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try:
      with open(path) as f:
        parts = f.read().split()
      if path.endswith('cpu.max'):
        if parts[0] != 'max':
          n = min(n, max(int(int(parts[0]) / int(parts[1])), 1))
      else:
        quota = int(parts[0])
        if quota > 0:
          with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            n = min(n, max(quota // int(f.read()), 1))
    except (OSError, ValueError, IndexError):
      pass
  return n


def cpu_baseline(args):
  """Oracle port on a bounded sample of the same workload, on this host's cores.

  Two restatements of the conv stack are timed on the first FoV steps of the
  same volume / seeds / options through the oracle's numpy canvas loop: the
  plain-C OpenMP one (best of a few thread counts) and the torch-CPU / oneDNN
  one (the stand-in for the reference's TF CPU path).
  The faster is reported as `value`."""
  from ffn_amd import synthetic
  from oracle import ffn_oracle
  variables = B.model_variables()
  blob = ffn_oracle.weights_blob(variables, B.DEPTH)
  shape = B.VOLUME_ZYX
  if args.workload == 'cells':
    vol = B.bench_volume(shape, args.workload_seed)
  else:
    vol = synthetic.noise_volume(shape, seed=0)
  image = synthetic.normalize(vol)
  seeds = ffn_oracle.grid_seeds(shape, tuple(f // 2 for f in B.FOV))
  ncpu = usable_cpus()

  results, oracle_trace = {}, {}
  impls = ['c_oracle']
  try:
    import torch  # noqa: F401
    impls.append('torch_onednn')
  except ImportError:
    pass
  probes = {}
  for impl in impls:
    # thread count: probes of >= 20 FoV steps each (4-step probes mostly time the
    # thread pool's spin-up)
    best_thr, best_rate = None, 0.0
    # (a 33^3 FoV does not feed more than a few dozen threads: 128 and 256 ran
    # at 1/5 and 1/300 of the 16-thread rate on the round's 256-core boxes)
    for thr in sorted({min(64, ncpu), min(32, ncpu), min(16, ncpu), min(8, ncpu)},
                      reverse=True):
      rate, n, _, _ = _cpu_run(image, blob, variables, seeds, impl, thr, 4.0,
                               args.cpu_probe_steps)
      probes.setdefault(impl, {})[str(thr)] = [round(rate, 2), n]
      if rate > best_rate:
        best_thr, best_rate = thr, rate
    rate, steps, dt, trace = _cpu_run(image, blob, variables, seeds, impl, best_thr,
                                      args.cpu_seconds / len(impls), args.cpu_steps,
                                      keep_trace=True)
    results[impl] = (rate, steps, dt, best_thr)
    # (FoV position, queued moves) of the run: the GPU replays it (gpu_parity_leg),
    # its first cpu_parity_steps steps
    oracle_trace[impl] = trace[:args.cpu_parity_steps]
  name = max(results, key=lambda k: results[k][0])
  rate, steps, dt, thr = results[name]
  # The whole box: floor(host cores / threads) such processes at once, each with
  # its own canvas and its own part of the seed grid -- what "this box's host
  # cores" deliver on this workload when none of them idles.
  whole = None
  procs = max(ncpu // max(thr, 1), 1)
  if procs <= 1:
    whole = {'value': round(rate, 3), 'unit': 'FoV-steps/s', 'processes': 1,
             'threads_each': int(thr), 'cores': int(thr),
             'what': 'this process may run on %d of the host\'s %d logical CPUs '
                     '(affinity mask / cgroup quota): the %d-thread sample above IS '
                     'the whole box as far as this job can use it'
                     % (ncpu, os.cpu_count() or 1, thr)}
  if procs > 1 and not args.no_cpu_whole_box:
    import subprocess
    import tempfile
    shm = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
    img_path = os.path.join(shm, 'ffn_amd_bench_cpu_image_%d.npy' % os.getpid())
    np.save(img_path, image)
    cmd = [sys.executable, os.path.abspath(B.__file__), '--cpu-workers', str(procs),
           '--cpu-threads', str(thr), '--cpu-impl', name, '--cpu-image', img_path,
           '--cpu-seconds', str(args.cpu_box_seconds), '--config', B.CONFIG,
           '--workload', args.workload, '--volume', str(args.volume)]
    t0 = time.perf_counter()
    try:
      ps = [subprocess.Popen(cmd + ['--cpu-worker', str(k)], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True) for k in range(procs)]
      outs = [json.loads(p.communicate(timeout=args.cpu_box_seconds * 6 + 120)[0]
                         .strip().splitlines()[-1]) for p in ps]
      whole = {
          'value': round(sum(o['steps'] for o in outs) / max(o['seconds'] for o in outs), 2),
          'unit': 'FoV-steps/s',
          'processes': procs, 'threads_each': int(thr), 'cores': int(procs * thr),
          'steps': int(sum(o['steps'] for o in outs)),
          'seconds': round(max(o['seconds'] for o in outs), 2),
          'what': '%d concurrent oracle processes (%s, %d threads each), each with '
                  'its own canvas over the same volume and its own part of the seed '
                  'grid; sum of their FoV steps / the longest of their clocks'
                  % (procs, name, thr),
          'wall_seconds_of_this_leg': round(time.perf_counter() - t0, 1),
      }
    except Exception as e:  # pylint:disable=broad-except
      whole = {'error': repr(e)}
    finally:
      try:
        os.remove(img_path)
      except OSError:
        pass
  return oracle_trace, {
      'value': round(rate, 3),
      'unit': 'FoV-steps/s',
      'cores': int(thr),
      'host_cores': os.cpu_count() or 1,
      'usable_cores': ncpu,
      'kind': 'port',
      'implementation': name,
      'all': {k: round(v[0], 3) for k, v in results.items()},
      'thread_probes': probes,
      'whole_box': whole,
      'sample': ('first %d FoV steps of the same %s %s workload (same seeds, '
                 'options, weights) through the oracle canvas loop with the %s '
                 'conv stack on %d threads, %.1f s; thread count = the best of '
                 '%d-step probes' % (steps, args.workload,
                                     'x'.join(str(v) for v in B.VOLUME_ZYX), name, thr,
                                     dt, args.cpu_probe_steps)),
  }


def gpu_parity_leg(res, oracle_traces, tol=1e-4):
  """The first FoV steps of the bench workload once more on the GPU -- a fresh
  device canvas, same volume / seeds / options, default kernels -- compared
  step for step with the trajectories the cpu_baseline leg just produced with
  the oracle's canvas loop: one behind the C oracle's forward (sequential f32
  sums), one behind the torch-CPU / oneDNN forward (the stand-in for the
  reference's TF CPU path).  FoV positions and queued move targets must be
  equal, move scores (the face maxima of the pasted logits) within `tol`.
  Untimed; rank 0 at N = 1 only."""
  from ffn_amd.inference import inference
  from ffn_amd.inference import inference_utils
  from ffn_amd.inference import movement
  from ffn_amd.inference import seed as seed_lib

  n_want = max((len(t) for t in oracle_traces.values()), default=0)
  if n_want == 0:
    return {'parity_steps_checked': 0, 'parity_ok': False}
  exe, model, request = res['exe'], res['model'], res['request']
  counters = inference_utils.Counters()
  canvas = inference.DeviceCanvas(
      model.info, exe.get_client(counters, direct=True), res['image'],
      request.inference_options, counters=counters,
      movement_policy_fn=movement.get_policy_fn(request, model.info))
  got = []
  thr = canvas.movement_policy.score_threshold
  deltas = canvas.movement_policy.deltas

  class _Enough(Exception):
    pass

  inner = canvas.update_at

  def recording_update(pos):  # an instance hook: the Python loop runs
    if len(got) >= n_want:
      raise _Enough()
    pred = inner(pos)
    moves = sorted(((s, tuple(int(v) for v in o))
                    for s, o, _ in pred.scored_move_offsets(deltas, thr)),
                   reverse=True)
    got.append((tuple(int(v) for v in pos), moves))
    return pred

  canvas.update_at = recording_update
  policy = functools.partial(seed_lib.PolicyGrid3d, step=16,
                             offsets=(0, 8, 4, 12, 2, 10, 14))
  try:
    canvas.segment_all(seed_policy=policy)
  except _Enough:
    pass
  canvas.close()

  def compare(trace):
    n = len(trace)
    ok = len(got) >= n
    max_err = 0.0
    first_bad = None
    for k in range(min(len(got), n)):
      (gp, gm), (op, om) = got[k], trace[k]
      same = (gp == tuple(int(v) for v in op) and len(gm) == len(om) and
              all(tuple(int(v) for v in a[1]) == b[1] for a, b in zip(om, gm)))
      if same and gm:
        err = max(abs(a[0] - b[0]) for a, b in zip(om, gm))
        max_err = max(max_err, err)
        same = err <= tol
      if not same:
        ok = False
        first_bad = k
        break
    return {'steps_checked': min(len(got), n), 'ok': bool(ok),
            'max_move_score_err': max_err, 'first_mismatch_step': first_bad}

  legs = {name: compare(trace) for name, trace in oracle_traces.items()}
  main_leg = legs['c_oracle']
  return {'parity_steps_checked': main_leg['steps_checked'],
          'parity_ok': bool(all(l['ok'] for l in legs.values())),
          'parity_max_move_score_err': max(l['max_move_score_err']
                                           for l in legs.values()),
          'parity_tolerance': tol,
          'parity_first_mismatch_step': main_leg['first_mismatch_step'],
          'parity_legs': legs,
          'parity_what': 'FoV position, queued move targets (equal) and move '
                         'scores (abs tol) of the first steps of this workload: '
                         'default GPU path vs the CPU oracle canvas loop, once '
                         'behind the C oracle forward and once behind the '
                         'torch-CPU / oneDNN forward'}
