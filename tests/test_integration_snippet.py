"""INTEGRATION.md section A, run AS WRITTEN: the python block is cut out of the document,
exec'ed into the namespace of the REFERENCE's own ffn/inference/executor.py (imported
through tools/ref_shims, unmodified), and the reference's own ThreadingExecutorClient /
Canvas.segment_all run behind the HipBatchExecutor it defines -- with `libffn_hip.so`
resolved to tests/abi_stub.c (the same four C entry points backed by the C oracle), so
that the snippet's argtypes, shapes, locking and error path are exercised without a GPU.
Expected: the reference-minted run tests/golden/ref_canvas_cells56.npz, bit for bit
(that fixture came from the reference Canvas behind the same oracle forward).

Needs /root/reference (this container only); skipped elsewhere."""
import ctypes
import functools
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('FFN_REFERENCE', '/root/reference')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'ffn')),
                                reason='the reference checkout is not on this machine')


def _snippet():
  text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
  section = text[text.index('## A.'):text.index('## B.')]
  blocks = re.findall(r'```python\n(.*?)```', section, re.S)
  assert len(blocks) == 1, 'section A holds exactly one python block'
  return blocks[0]


@pytest.fixture(scope='module')
def stub(tmp_path_factory):
  out = str(tmp_path_factory.mktemp('abi_stub') / 'libffn_hip.so')
  subprocess.check_call(['gcc', '-O3', '-march=x86-64-v3', '-fopenmp', '-fPIC', '-std=c11',
                         '-shared', '-o', out, os.path.join(ROOT, 'tests', 'abi_stub.c'),
                         os.path.join(ROOT, 'oracle', 'convstack_oracle.c')])
  return out


@pytest.fixture(scope='module')
def ref(stub):
  """The reference's modules + the namespace the snippet was exec'ed into."""
  os.environ.setdefault('PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION', 'python')
  added = [REF, os.path.join(ROOT, 'tools', 'ref_shims')]
  sys.path[:0] = added
  try:
    from ffn.inference import executor as ref_executor
    from ffn.inference import inference as ref_inference
    from ffn.inference import inference_pb2
    from ffn.inference import inference_utils as ref_utils
    from ffn.inference import movement as ref_movement
    from ffn.inference import seed as ref_seed
    from ffn.training import model as ref_model
  finally:
    for p in added:
      sys.path.remove(p)
  assert ref_executor.__file__.startswith(REF)
  real_cdll = ctypes.CDLL

  def cdll(name, *a, **k):  # `libffn_hip.so` = the stub on this machine
    return real_cdll(stub if name == 'libffn_hip.so' else name, *a, **k)

  ns = dict(vars(ref_executor))  # "an addition to ffn/inference/executor.py"
  ctypes.CDLL = cdll
  try:
    exec(compile(_snippet(), 'INTEGRATION.md#A', 'exec'), ns)  # pylint:disable=exec-used
  finally:
    ctypes.CDLL = real_cdll
  mods = dict(executor=ref_executor, inference=ref_inference, pb2=inference_pb2,
              utils=ref_utils, movement=ref_movement, seed=ref_seed, model=ref_model)
  return ns, mods


def _blob():
  from oracle import ffn_oracle
  with np.load(os.path.join(GOLDEN, 'fib25_weights.npz')) as d:
    return ffn_oracle.weights_blob({k: d[k] for k in d.files}, 12)


def test_snippet_class_is_a_subclass_of_the_references_executor(ref):
  ns, mods = ref
  cls = ns['HipBatchExecutor']
  assert issubclass(cls, mods['executor'].ThreadingBatchExecutor)
  assert cls._run_executor is mods['executor'].ThreadingBatchExecutor._run_executor
  assert cls._schedule_batch is not mods['executor'].ThreadingBatchExecutor._schedule_batch


def test_reference_canvas_behind_the_snippet_reproduces_the_reference_run(ref):
  """ffn/inference/executor.py:207-340 (server loop, unmodified) + inference.py:356-384,
  460-683 (Canvas.predict / update_at / segment_all, unmodified) + the snippet."""
  from ffn_amd import synthetic
  ns, m = ref
  g = np.load(os.path.join(GOLDEN, 'ref_canvas_cells56.npz'))
  info = m['model'].ModelInfo(deltas=np.array([8, 8, 8]), pred_mask_size=np.array([33] * 3),
                              input_seed_size=np.array([33] * 3),
                              input_image_size=np.array([33] * 3))
  request = m['pb2'].InferenceRequest()
  o = request.inference_options
  o.init_activation, o.pad_value, o.move_threshold, o.segment_threshold = 0.95, 0.05, 0.9, 0.6
  o.min_segment_size = 1000
  o.min_boundary_dist.x = o.min_boundary_dist.y = o.min_boundary_dist.z = 1
  counters = m['utils'].Counters()
  iface = m['executor'].ExecutorInterface()
  exe = ns['HipBatchExecutor'](iface, info, 12, _blob(), counters, batch_size=1)
  exe.start_server()
  try:
    client = exe.get_client(counters)  # the reference's ThreadingExecutorClient
    assert type(client) is m['executor'].ThreadingExecutorClient
    canvas = m['inference'].Canvas(
        info, client, synthetic.normalize(g['volume']), o, counters=counters,
        movement_policy_fn=m['movement'].get_policy_fn(request, info))
    seeds = g['seeds']

    class FixedSeeds(m['seed'].BaseSeedPolicy):

      def init_coords(self):
        self.coords = np.array(seeds)

    canvas.segment_all(seed_policy=FixedSeeds)
  finally:
    exe.stop_server()
  assert np.array_equal(np.asarray(canvas.segmentation), g['segmentation'])
  got = np.asarray(canvas.seed)
  assert np.array_equal(np.isnan(got), np.isnan(g['seed_logits']))
  assert np.array_equal(got[~np.isnan(got)], g['seed_logits'][~np.isnan(got)])
  ref_counters = json.loads(str(g['counters']))
  assert counters['update_at-calls'].value == ref_counters['update_at-calls'] == len(g['steps'])
  origins = json.loads(str(g['origins']))
  assert {int(k): [list(int(x) for x in v.start_zyx), int(v.iters)]
          for k, v in canvas.origins.items()} == {int(k): v for k, v in origins.items()}
  lib = ctypes.CDLL(ns['_hip']._name)
  assert lib.ffn_stub_predict_calls() == len(g['steps'])  # one ffn_predict per FoV step


def test_snippet_surfaces_library_errors(ref):
  """_check raises with ffn_last_error's text (the reference's fail-fast contract,
  executor.py:187-200): a blob of the wrong size never reaches the server thread."""
  ns, m = ref
  info = m['model'].ModelInfo(deltas=np.array([8, 8, 8]), pred_mask_size=np.array([33] * 3),
                              input_seed_size=np.array([33] * 3),
                              input_image_size=np.array([33] * 3))
  with pytest.raises(RuntimeError, match='weight blob has 10 floats'):
    ns['HipBatchExecutor'](m['executor'].ExecutorInterface(), info, 12,
                           np.zeros(10, np.float32), m['utils'].Counters(), batch_size=1)
