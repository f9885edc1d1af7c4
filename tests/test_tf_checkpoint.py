"""TensorBundle reader (ffn_amd/training/tf_checkpoint.py) on a synthetic bundle
written here in the on-disk format the reference's checkpoints use
(models/fib25/model.ckpt-*: leveldb-style .index table + raw .data shard)."""

import struct

import numpy as np
import pytest

from ffn_amd.training import tf_checkpoint


def _varint(v):
  out = b''
  while True:
    b = v & 0x7F
    v >>= 7
    if v:
      out += bytes([b | 0x80])
    else:
      return out + bytes([b])


def _field(num, wt, payload):
  tag = _varint((num << 3) | wt)
  if wt == 0:
    return tag + _varint(payload)
  if wt == 2:
    return tag + _varint(len(payload)) + payload
  if wt == 5:
    return tag + payload
  raise ValueError(wt)


def _entry_proto(dtype, shape, offset, size, sliced=False):
  dims = b''.join(_field(2, 2, _field(1, 0, d)) for d in shape)
  p = _field(1, 0, dtype) + _field(2, 2, dims) + _field(4, 0, offset) + \
      _field(5, 0, size) + _field(6, 5, b'\0\0\0\0')
  if sliced:
    p += _field(7, 2, b'\x0a\x00')
  return p


def _block(items):
  body = b''
  for key, value in items:
    body += _varint(0) + _varint(len(key)) + _varint(len(value)) + key + value
  return body + struct.pack('<II', 0, 1)


def _write_bundle(prefix, tensors, extra=()):
  data = b''
  items = [(b'', _field(1, 0, 1))]  # BundleHeaderProto num_shards = 1
  entries = []
  for name, arr, enum in tensors:
    raw = arr.tobytes()
    entries.append((name.encode(), _entry_proto(enum, arr.shape, len(data),
                                                len(raw))))
    data += raw
  entries += list(extra)
  items += sorted(entries)
  block = _block(items)
  table = block + b'\0' + b'\0\0\0\0'
  index_block = _block([(b'\xff', _varint(0) + _varint(len(block)))])
  index_off = len(table)
  table += index_block + b'\0' + b'\0\0\0\0'
  meta = _block([])
  meta_off = len(table)
  table += meta + b'\0' + b'\0\0\0\0'
  footer = _varint(meta_off) + _varint(len(meta)) + _varint(index_off) + \
      _varint(len(index_block))
  footer += b'\0' * (40 - len(footer)) + struct.pack('<Q', 0xDB4775248B80FB57)
  with open(prefix + '.index', 'wb') as f:
    f.write(table + footer)
  with open(prefix + '.data-00000-of-00001', 'wb') as f:
    f.write(data)


def test_reads_f32_and_skips_unsupported_entries(tmp_path):
  prefix = str(tmp_path / 'model.ckpt-1')
  w = np.arange(2 * 3 * 4, dtype='<f4').reshape(2, 3, 4) / 7
  b = np.array([1.5, -2.5], '<f4')
  step = np.array(27465036, '<i8')
  extra = [
      # a DT_STRING entry (enum 7) as object-based TF2 checkpoints carry, a
      # bfloat16 one (14) and a sliced (partitioned) f32 variable
      (b'_CHECKPOINTABLE_OBJECT_GRAPH', _entry_proto(7, (), 0, 4)),
      (b'seed_update/half', _entry_proto(14, (2,), 0, 4)),
      (b'seed_update/partitioned', _entry_proto(1, (4,), 0, 16, sliced=True)),
  ]
  _write_bundle(prefix, [('seed_update/conv0_a/weights', w, 1),
                         ('seed_update/conv0_a/biases', b, 1),
                         ('global_step', step, 9)], extra)
  listed = tf_checkpoint.list_variables(prefix)
  assert set(listed) == {'seed_update/conv0_a/weights',
                         'seed_update/conv0_a/biases', 'global_step',
                         '_CHECKPOINTABLE_OBJECT_GRAPH', 'seed_update/half',
                         'seed_update/partitioned'}
  assert listed['_CHECKPOINTABLE_OBJECT_GRAPH']['dtype'] is None
  assert listed['seed_update/partitioned']['sliced']
  got = tf_checkpoint.load_checkpoint(prefix)
  assert set(got) == {'seed_update/conv0_a/weights',
                      'seed_update/conv0_a/biases', 'global_step'}
  assert np.array_equal(got['seed_update/conv0_a/weights'], w)
  assert np.array_equal(got['seed_update/conv0_a/biases'], b)
  assert int(got['global_step']) == 27465036
  only = tf_checkpoint.load_checkpoint(prefix,
                                       names=['seed_update/conv0_a/biases'])
  assert list(only) == ['seed_update/conv0_a/biases']
  with pytest.raises(ValueError):
    tf_checkpoint.load_checkpoint(prefix, names=['seed_update/half'])
  with pytest.raises(ValueError):
    tf_checkpoint.load_checkpoint(prefix, names=['seed_update/partitioned'])
  with pytest.raises(KeyError):
    tf_checkpoint.load_checkpoint(prefix, names=['missing'])
