"""The C-ABI library loads (no GPU needed) and exports every symbol that
include/*.h declare; entry points fail loudly, never fall back."""

import ctypes
import os
import re

import pytest

from ffn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
  text = ''
  for header in sorted(os.listdir(os.path.join(ROOT, 'include'))):
    with open(os.path.join(ROOT, 'include', header)) as f:
      text += f.read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(ffn_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
  assert os.path.exists(_lib.LIB_PATH), 'build with __graft_entry__.build()'
  lib = ctypes.CDLL(_lib.LIB_PATH)
  declared = _declared_symbols()
  assert len(declared) >= 25
  for name in declared:
    assert hasattr(lib, name), 'missing export: %s' % name
  assert set(declared) == set(_lib.SIGNATURES), (
      set(declared) ^ set(_lib.SIGNATURES))


def test_abi_version_and_weight_count():
  lib = _lib.load()
  assert lib.ffn_abi_version() == 10
  assert lib.ffn_engine_weight_count(12, 32) == 638433
  assert lib.ffn_engine_weight_count(18, 32) == 27 * 2 * 32 + 32 + 35 * (
      27 * 32 * 32 + 32) + 33


def test_no_cpu_fallback_without_gpu():
  """Without a GPU the product path raises; it never routes to the oracle."""
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from ffn_amd import engine
  with pytest.raises(_lib.FFNHipError):
    engine.HipEngine((33, 33, 33), (8, 8, 8), 12)


def test_product_never_imports_the_oracle():
  import ast
  pkg = os.path.join(ROOT, 'ffn_amd')
  for dirpath, _, files in os.walk(pkg):
    for fn in files:
      if not fn.endswith('.py'):
        continue
      with open(os.path.join(dirpath, fn)) as f:
        tree = ast.parse(f.read())
      for node in ast.walk(tree):
        names = []
        if isinstance(node, ast.Import):
          names = [a.name for a in node.names]
        elif isinstance(node, ast.ImportFrom):
          names = [node.module or '']
        assert not any(n.split('.')[0] == 'oracle' for n in names), (fn, names)


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
  """include/ffn_hip.h compiled as plain C: sizeof / offsetof of the structs the
  Python side mirrors with ctypes (ffn_amd/_lib.py) are what ctypes lays out."""
  import ctypes
  import subprocess
  from ffn_amd import _lib
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  structs = {
      'ffn_turn_request': (_lib.TurnRequest, [
          'do_commit', 'lo', 'hi', 'segment_threshold', 'min_segment_size',
          'segment_id', 'max_existing_id', 'mark_mode', 'mark_pos',
          'num_candidates', 'min_boundary_dist', 'do_init', 'init_value']),
      'ffn_turn_result': (_lib.TurnResult, ['counts', 'committed', 'chosen']),
      'ffn_commit_counts': (_lib.CommitCounts, [
          'raw_segmented_voxels', 'actual_segmented_voxels', 'num_overlapped_ids']),
      'ffn_segment_result': (_lib.SegmentResult, ['num_steps']),
      'ffn_segment_params': (_lib.SegmentParams, ['step', 'score_threshold']),
      'ffn_step_result': (_lib.StepResult, ['face_score', 'start_logit']),
      'ffn_step_request': (_lib.StepRequest, ['pos', 'num_candidates']),
  }
  lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ffn_hip.h"',
           'int main(void) {']
  for name, (_, fields) in structs.items():
    lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (name, name))
    for f in fields:
      lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
  lines += ['  return 0;', '}']
  src = tmp_path / 'layout.c'
  src.write_text('\n'.join(lines))
  exe = tmp_path / 'layout'
  subprocess.check_call(['gcc', '-std=c11', '-I', os.path.join(root, 'include'),
                         '-o', str(exe), str(src)])
  out = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
  for name, (cls, fields) in structs.items():
    assert int(out[name]) == ctypes.sizeof(cls), name
    for f in fields:
      assert int(out['%s.%s' % (name, f)]) == getattr(cls, f).offset, (name, f)


def test_from_model_refuses_a_prediction_that_cannot_be_centred():
  """(seed - pred) odd on an axis is not a geometry the reference can
  run (inference.py:218,410-411); refused before any device is touched."""
  import pytest
  from ffn_amd import engine as hip_engine
  from ffn_amd.training import model as model_lib

  class _Odd:
    info = model_lib.ModelInfo(deltas=(8, 8, 8), pred_mask_size=(28, 33, 33),
                               input_seed_size=(33, 33, 33),
                               input_image_size=(33, 33, 33))
    depth, features = 12, 32

  with pytest.raises(ValueError, match='even'):
    hip_engine.HipEngine.from_model(_Odd(), max_batch=1)
