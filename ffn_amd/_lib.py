"""ctypes binding of libffn_hip.so (the C-ABI in include/ffn_hip.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, every
entry point raises.  The oracle under oracle/ is test infrastructure and is
never imported from here.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
# (FFN_AMD_LIB: another build of the same sources, for A/B runs of compile-time
# kernel switches)
LIB_PATH = os.environ.get('FFN_AMD_LIB') or os.path.join(CSRC, 'libffn_hip.so')
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'ffn_hip.h')
HEADERS = [HEADER,
           os.path.join(os.path.dirname(_HERE), 'include', 'ffn_labels.h'),
           os.path.join(os.path.dirname(_HERE), 'include', 'ffn_seeds.h')]
SOURCES = ['ffn_hip.hip', 'ffn_labels.hip', 'ffn_seeds.hip']

MAX_CANDIDATES = 16


class FFNHipError(RuntimeError):
  """Raised for any non-zero return code of the C-ABI."""


class FFNRangeError(FFNHipError):
  """FFN_ERR_RANGE: a split-product kernel (conv_variant >= 6) met an operand
  outside the fp16 range; the step changed nothing and can be repeated with the
  exact-f32 kernel (conv_variant -1)."""


class FFNFlowError(FFNHipError):
  """FFN_ERR_FLOW: the resident conv launch (engine option flow = 2) timed out
  waiting for one of its own workgroups (a shared or partitioned GPU); the step
  changed nothing and is simply repeated -- the library runs the repeat as one
  launch per conv (same arithmetic) and turns the resident launch off by itself
  after three such steps in a row."""


ERR_RANGE = -4
ERR_FLOW = -5
#: return codes of a step that changed nothing and is to be repeated
ERR_VOIDED = (ERR_RANGE, ERR_FLOW)


class StepParams(ctypes.Structure):
  _fields_ = [('pad_value', ctypes.c_float),
              ('move_threshold', ctypes.c_float),
              ('disco_seed_threshold', ctypes.c_float),
              ('deleted_threshold', ctypes.c_float)]

  def __init__(self, pad_value=0.0, move_threshold=0.0,
               disco_seed_threshold=0.0, deleted_threshold=float('nan')):
    # deleted_threshold = NaN: the keep_history count is not requested
    super().__init__(pad_value, move_threshold, disco_seed_threshold,
                     deleted_threshold)


class StepRequest(ctypes.Structure):
  _fields_ = [('pos', ctypes.c_int32 * 3),
              ('start_pos', ctypes.c_int32 * 3),
              ('num_candidates', ctypes.c_int32),
              ('candidates', (ctypes.c_int32 * 3) * MAX_CANDIDATES)]


class StepResult(ctypes.Structure):
  _fields_ = [('face_score', ctypes.c_float * 6),
              ('face_index', ctypes.c_int32 * 6),
              ('face_seg', ctypes.c_int32 * 6),
              ('start_logit', ctypes.c_float),
              ('num_above_move', ctypes.c_uint32),
              ('disco_applied', ctypes.c_int32),
              ('cand_seed', ctypes.c_float * MAX_CANDIDATES),
              ('cand_seg', ctypes.c_int32 * MAX_CANDIDATES),
              ('num_deleted', ctypes.c_uint32),
              ('range_error', ctypes.c_int32)]


class SegmentParams(ctypes.Structure):
  """ffn_segment_params (include/ffn_hip.h)."""
  _fields_ = [('step', StepParams),
              ('score_threshold', ctypes.c_double),
              ('deltas_zyx', ctypes.c_int32 * 3),
              ('margin_zyx', ctypes.c_int32 * 3),
              ('shape_zyx', ctypes.c_int32 * 3),
              ('init_min_pos', ctypes.c_int32 * 3),
              ('init_max_pos', ctypes.c_int32 * 3),
              ('initial_start_logit', ctypes.c_float),
              ('prefetch', ctypes.c_int32),
              ('keep_history', ctypes.c_int32),
              ('max_steps', ctypes.c_int64)]


class SegmentResult(ctypes.Structure):
  """ffn_segment_result (include/ffn_hip.h)."""
  _fields_ = [('num_steps', ctypes.c_int64),
              ('skip_threshold', ctypes.c_int64),
              ('skip_invalid_pos', ctypes.c_int64),
              ('gate_rejects', ctypes.c_int64),
              ('queue_len', ctypes.c_int64),
              ('seed_got_too_weak', ctypes.c_int32),
              ('budget_exhausted', ctypes.c_int32),
              ('active', ctypes.c_int32),
              ('start_logit_known', ctypes.c_int32),
              ('start_logit', ctypes.c_float),
              ('min_pos', ctypes.c_int32 * 3),
              ('max_pos', ctypes.c_int32 * 3)]


class CommitCounts(ctypes.Structure):
  _fields_ = [('raw_segmented_voxels', ctypes.c_int64),
              ('actual_segmented_voxels', ctypes.c_int64),
              ('num_overlapped_ids', ctypes.c_int32)]


class TurnRequest(ctypes.Structure):
  """ffn_turn_request (include/ffn_hip.h)."""
  _fields_ = [('do_commit', ctypes.c_int32),
              ('lo', ctypes.c_int32 * 3), ('hi', ctypes.c_int32 * 3),
              ('segment_threshold', ctypes.c_float),
              ('min_segment_size', ctypes.c_int64),
              ('segment_id', ctypes.c_int32),
              ('max_existing_id', ctypes.c_int32),
              ('mark_mode', ctypes.c_int32),
              ('mark_pos', ctypes.c_int32 * 3),
              ('num_candidates', ctypes.c_int32),
              ('min_boundary_dist', ctypes.c_int32 * 3),
              ('do_init', ctypes.c_int32),
              ('init_value', ctypes.c_float)]


class TurnResult(ctypes.Structure):
  """ffn_turn_result (include/ffn_hip.h)."""
  _fields_ = [('counts', CommitCounts), ('committed', ctypes.c_int32),
              ('chosen', ctypes.c_int32)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_I3 = ctypes.POINTER(ctypes.c_int32)

# name -> (restype, argtypes); every symbol include/ffn_hip.h declares.
SIGNATURES = {
    'ffn_abi_version': (_I, []),
    'ffn_last_error': (ctypes.c_char_p, []),
    'ffn_engine_weight_count': (ctypes.c_size_t, [_I, _I]),
    'ffn_engine_create': (_I, [_I, _I3, _I3, _I, _I, _I,
                               ctypes.POINTER(_P)]),
    'ffn_engine_destroy': (None, [_P]),
    'ffn_engine_set_weights': (_I, [_P, _P, ctypes.c_size_t]),
    'ffn_engine_set_pred_size': (_I, [_P, _I3]),
    'ffn_engine_set_option': (_I, [_P, ctypes.c_char_p, _I]),
    'ffn_engine_get_option': (_I, [_P, ctypes.c_char_p,
                                   ctypes.POINTER(ctypes.c_int)]),
    'ffn_engine_set_profiling': (_I, [_P, _I]),
    'ffn_engine_get_profile': (_I, [_P, ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_int64), _I]),
    'ffn_engine_get_profile_samples': (_I, [_P, _P, _I, _P]),
    'ffn_engine_synchronize': (_I, [_P]),
    'ffn_engine_debug_clocks': (_I, [_P, _P]),
    'ffn_engine_debug_workgroups': (_I, [_P, _P, _I]),
    'ffn_engine_debug_flow_trace': (_I, [_P, _P, _I]),
    'ffn_predict': (_I, [_P, _I, _P, _P, _P]),
    'ffn_forward_resident': (_I, [_P, _I, _I]),
    'ffn_canvas_create': (_I, [_P, _P, _I3, ctypes.POINTER(_P)]),
    'ffn_canvas_create_u8': (_I, [_P, _P, _I3, ctypes.c_float, ctypes.c_float,
                                  ctypes.POINTER(_P)]),
    'ffn_canvas_destroy': (None, [_P]),
    'ffn_canvas_init_seed': (_I, [_P, _I3, ctypes.c_float]),
    'ffn_canvas_step': (_I, [_P, _I, ctypes.POINTER(_P),
                             ctypes.POINTER(StepRequest),
                             ctypes.POINTER(StepParams),
                             ctypes.POINTER(StepResult)]),
    'ffn_canvas_step_submit': (_I, [_P, _I, ctypes.POINTER(_P),
                                    ctypes.POINTER(StepRequest),
                                    ctypes.POINTER(StepParams),
                                    ctypes.POINTER(ctypes.c_uint32)]),
    'ffn_canvas_step_wait': (_I, [_P, ctypes.c_uint32,
                                  ctypes.POINTER(StepResult)]),
    'ffn_canvas_segment_at': (_I, [_P, _I3, ctypes.POINTER(SegmentParams), _I,
                                   ctypes.POINTER(SegmentResult)]),
    'ffn_canvas_segment_many': (_I, [_P, _I, _P, _P, _P, _P, _P, _P]),
    'ffn_canvas_segment_many_carry': (_I, [_P, _I, _P, _P, _P, _P, _P, _P,
                                           ctypes.c_int32]),
    'ffn_canvas_segment_history': (_I, [_P, ctypes.c_size_t, ctypes.c_size_t,
                                        _P, _P,
                                        ctypes.POINTER(ctypes.c_size_t)]),
    'ffn_canvas_read_points': (_I, [_P, _I, _P, _P, _P]),
    'ffn_canvas_write_seg_points': (_I, [_P, _I, _P, _P]),
    'ffn_canvas_any_segmented': (_I, [_P, _I3, _I3,
                                      ctypes.POINTER(ctypes.c_int32)]),
    'ffn_canvas_commit_count': (_I, [_P, _I3, _I3, ctypes.c_float,
                                     ctypes.c_int32,
                                     ctypes.POINTER(CommitCounts),
                                     ctypes.c_int32, _P, _P]),
    'ffn_canvas_commit_assign': (_I, [_P, _I3, _I3, ctypes.c_float,
                                      ctypes.c_int32]),
    'ffn_canvas_segment_turn': (_I, [_P, _P, _P, _P, ctypes.c_int32, _P, _P, _P, _P,
                                     _P]),
    'ffn_canvas_read_seed': (_I, [_P, _I3, _I3, _P]),
    'ffn_canvas_read_segmentation': (_I, [_P, _I3, _I3, _P]),
    'ffn_canvas_write_seed': (_I, [_P, _I3, _I3, _P]),
    'ffn_canvas_write_segmentation': (_I, [_P, _I3, _I3, _P]),
    # include/ffn_labels.h
    'ffn_labels_create': (_I, [_I, ctypes.POINTER(_P)]),
    'ffn_labels_destroy': (None, [_P]),
    'ffn_labels_pair_counts': (_I, [_P, _P, _P, _I, ctypes.c_size_t,
                                    ctypes.c_size_t, _P, _P, _P, _P,
                                    ctypes.POINTER(ctypes.c_size_t)]),
    'ffn_labels_apply_pair_labels': (_I, [_P, ctypes.c_size_t, _P, _P, _P]),
    'ffn_labels_remap': (_I, [_P, _P, _I, ctypes.c_size_t, ctypes.c_size_t,
                              _P, _P, _I, _P]),
    'ffn_labels_connected_components': (
        _I, [_P, _P, _I, ctypes.POINTER(ctypes.c_int64), _I, _P,
             ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t, _P, _P,
             ctypes.POINTER(ctypes.c_int64)]),
    'ffn_labels_last_timing': (_I, [_P, ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_double)]),
    'ffn_labels_copy_device': (_I, [_P, _P, ctypes.c_size_t, _P]),
    'ffn_labels_copy_canvas': (_I, [_P, _P, _P]),
    'ffn_labels_place_core_device': (
        _I, [_P, _P, ctypes.POINTER(ctypes.c_int64),
             ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
             ctypes.c_int32, _P, ctypes.POINTER(ctypes.c_int64),
             ctypes.POINTER(ctypes.c_int64)]),
    'ffn_labels_margin_pairs_device': (
        _I, [_P, _P, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32,
             ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), _P,
             ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
             ctypes.c_size_t, _P, _P, _P, ctypes.POINTER(ctypes.c_size_t)]),
    'ffn_labels_remap_device': (_I, [_P, _P, ctypes.c_size_t, ctypes.c_size_t,
                                     _P, _P]),
    # include/ffn_seeds.h
    'ffn_seeder_create': (_I, [_I, ctypes.POINTER(_P)]),
    'ffn_seeder_destroy': (None, [_P]),
    'ffn_seeder_set_noise': (_I, [_P, _P, ctypes.c_size_t]),
    'ffn_seeder_set_gaussian': (_I, [_P, _P, _I]),
    'ffn_seeder_peaks': (_I, [_P, _P, _P, _P, ctypes.POINTER(ctypes.c_int64),
                              ctypes.POINTER(ctypes.c_double), ctypes.c_size_t,
                              _P, ctypes.POINTER(ctypes.c_size_t),
                              ctypes.POINTER(ctypes.c_int32)]),
    'ffn_seeder_peaks_canvas': (_I, [_P, _P, ctypes.POINTER(ctypes.c_double),
                                     ctypes.c_size_t, _P,
                                     ctypes.POINTER(ctypes.c_size_t),
                                     ctypes.POINTER(ctypes.c_int32)]),
    'ffn_seeder_edt': (_I, [_P, _P, ctypes.POINTER(ctypes.c_int64),
                            ctypes.POINTER(ctypes.c_double), _P]),
    'ffn_seeder_read_stage': (_I, [_P, _I, _P]),
    'ffn_seeder_last_timing': (_I, [_P, ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_double)]),
}

_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
  """Compiles csrc/*.hip for gfx950 into csrc/libffn_hip.so (in-tree): one
  object per source under csrc/build/ (stale ones only, in parallel), then the
  link."""
  headers = HEADERS + [os.path.join(CSRC, name) for name in sorted(os.listdir(CSRC))
                       if name.endswith('.h')]
  hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
  if not os.path.exists(hipcc):
    hipcc = 'hipcc'
  objdir = os.path.join(CSRC, 'build')
  os.makedirs(objdir, exist_ok=True)
  jobs, objs = [], []
  for name in SOURCES:
    src = os.path.join(CSRC, name)
    obj = os.path.join(objdir, os.path.splitext(name)[0] + '.o')
    objs.append(obj)
    if (force or not os.path.exists(obj) or
        any(os.path.getmtime(obj) < os.path.getmtime(d) for d in [src] + headers)):
      jobs.append((name, subprocess.Popen(
          [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c',
           src, '-o', obj], cwd=CSRC)))
  failed = [name for name, p in jobs if p.wait() != 0]
  if failed:
    raise FFNHipError('hipcc failed for %s' % ', '.join(failed))
  if (jobs or not os.path.exists(LIB_PATH) or
      any(os.path.getmtime(LIB_PATH) < os.path.getmtime(o) for o in objs)):
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC',
                           '-o', LIB_PATH] + objs, cwd=CSRC)
  return LIB_PATH


def load() -> ctypes.CDLL:
  """Loads the HIP library; raises (never falls back) if it is unavailable."""
  global _lib
  with _lock:
    if _lib is not None:
      return _lib
    if not os.path.exists(LIB_PATH):
      raise FFNHipError(
          '%s not found: build it with `python -c "import __graft_entry__ as g;'
          ' g.build()"` (hipcc --offload-arch=gfx950). There is no CPU '
          'fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(lib, name)  # AttributeError if the symbol is missing
      fn.restype = res
      fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
  if rc != 0:
    msg = load().ffn_last_error()
    cls = (FFNRangeError if rc == ERR_RANGE else
           FFNFlowError if rc == ERR_FLOW else FFNHipError)
    raise cls('libffn_hip error %d: %s' %
                      (rc, msg.decode('utf-8', 'replace') if msg else '?'))


def i3(values):
  return (ctypes.c_int32 * 3)(*[int(v) for v in values])


def csrc_sha() -> str:
  """sha256 over the kernel / host sources of the library (ffn_amd/csrc/*.hip, *.h), in
  name order: what a committed profile was taken on (profiles/*_pmc.json `csrc_sha`) against
  what this tree builds (bench.py: roofline.traffic_stale)."""
  import hashlib
  h = hashlib.sha256()
  for name in sorted(os.listdir(CSRC)):
    if name.endswith(('.hip', '.h')):
      h.update(name.encode())
      with open(os.path.join(CSRC, name), 'rb') as f:
        h.update(f.read())
  return h.hexdigest()[:16]
