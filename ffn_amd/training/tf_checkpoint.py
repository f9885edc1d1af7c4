"""Pure-Python reader for TensorFlow "TensorBundle" checkpoints.

The reference restores weights with ``tf.train.Saver().restore``
(reference ffn/inference/runner.py:98-111).  TensorFlow is neither a dependency
of this package nor available on the MI355X image, so the on-disk format is
parsed directly:

* ``<prefix>.index`` is a leveldb-style sorted string table.  A 48-byte footer
  carries two block handles (metaindex, index) and the magic number
  0xdb4775248b80fb57.  Every block is followed by a 5-byte trailer
  (compression byte + crc32c).  Inside a block, entries are prefix-compressed
  ``(shared, non_shared, value_len, key_suffix, value)`` varint records followed
  by a restart array.
* key ``""`` maps to a ``BundleHeaderProto``; every other key maps to a
  ``BundleEntryProto`` {1: dtype, 2: shape, 3: shard_id, 4: offset, 5: size,
  6: crc32c}.
* ``<prefix>.data-0000N-of-0000M`` holds raw little-endian tensor bytes.

Only what an FFN inference checkpoint needs is implemented (uncompressed
blocks, float/int tensors).
"""

from __future__ import annotations

import os
import struct
from typing import Dict, Tuple

import numpy as np

_TABLE_MAGIC = 0xDB4775248B80FB57

# tensorflow/core/framework/types.proto
_DTYPES = {
    1: np.dtype('<f4'),
    2: np.dtype('<f8'),
    3: np.dtype('<i4'),
    4: np.dtype('u1'),
    6: np.dtype('i1'),
    9: np.dtype('<i8'),
    10: np.dtype('bool'),
}


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
  result = 0
  shift = 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7F) << shift
    if not b & 0x80:
      return result, pos
    shift += 7


def _read_block(data: bytes, offset: int, size: int) -> bytes:
  block = data[offset:offset + size]
  compression = data[offset + size]
  if compression != 0:
    raise ValueError('compressed table blocks are not supported (type %d)' %
                     compression)
  return block


def _block_entries(block: bytes):
  """Yields (key, value) from one prefix-compressed table block."""
  num_restarts = struct.unpack('<I', block[-4:])[0]
  limit = len(block) - 4 - 4 * num_restarts
  pos = 0
  key = b''
  while pos < limit:
    shared, pos = _varint(block, pos)
    non_shared, pos = _varint(block, pos)
    value_len, pos = _varint(block, pos)
    key = key[:shared] + block[pos:pos + non_shared]
    pos += non_shared
    value = block[pos:pos + value_len]
    pos += value_len
    yield key, value


def _parse_proto(buf: bytes):
  """Minimal protobuf wire-format walk -> list of (field, wire_type, value)."""
  out = []
  pos = 0
  while pos < len(buf):
    tag, pos = _varint(buf, pos)
    field, wt = tag >> 3, tag & 7
    if wt == 0:
      val, pos = _varint(buf, pos)
    elif wt == 1:
      val = buf[pos:pos + 8]
      pos += 8
    elif wt == 2:
      ln, pos = _varint(buf, pos)
      val = buf[pos:pos + ln]
      pos += ln
    elif wt == 5:
      val = buf[pos:pos + 4]
      pos += 4
    else:
      raise ValueError('unsupported wire type %d' % wt)
    out.append((field, wt, val))
  return out


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
  dims = []
  for field, _, val in _parse_proto(buf):
    if field == 2:  # repeated Dim
      size = 0
      for f2, _, v2 in _parse_proto(val):
        if f2 == 1:
          size = v2
      dims.append(size)
  return tuple(dims)


def list_variables(prefix: str) -> Dict[str, dict]:
  """Returns {name: {dtype, shape, shard, offset, size}} for a checkpoint."""
  with open(prefix + '.index', 'rb') as f:
    data = f.read()
  footer = data[-48:]
  magic = struct.unpack('<Q', footer[-8:])[0]
  if magic != _TABLE_MAGIC:
    raise ValueError('%s.index: not a TensorBundle table' % prefix)
  pos = 0
  _, pos = _varint(footer, pos)  # metaindex offset
  _, pos = _varint(footer, pos)  # metaindex size
  index_off, pos = _varint(footer, pos)
  index_size, pos = _varint(footer, pos)

  entries = {}
  index_block = _read_block(data, index_off, index_size)
  for _, handle in _block_entries(index_block):
    off, p = _varint(handle, 0)
    size, p = _varint(handle, p)
    for key, value in _block_entries(_read_block(data, off, size)):
      if key == b'':
        continue  # BundleHeaderProto
      info = {'dtype': None, 'dtype_enum': 0, 'shape': (), 'shard': 0,
              'offset': 0, 'size': 0, 'sliced': False}
      for field, _, val in _parse_proto(value):
        if field == 1:
          # unknown dtypes (DT_STRING object graphs of TF2 checkpoints, half,
          # bfloat16 ...) are recorded, not rejected: only a tensor that is
          # actually requested must be readable
          info['dtype_enum'] = val
          info['dtype'] = _DTYPES.get(val)
        elif field == 7:
          info['sliced'] = True  # BundleEntryProto.slices: partitioned variable
        elif field == 2:
          info['shape'] = _parse_shape(val)
        elif field == 3:
          info['shard'] = val
        elif field == 4:
          info['offset'] = val
        elif field == 5:
          info['size'] = val
      entries[key.decode('utf-8')] = info
  return entries


def load_checkpoint(prefix: str, names=None) -> Dict[str, np.ndarray]:
  """Loads the tensors of a TensorBundle checkpoint into numpy arrays.

  names: tensors to load (KeyError / ValueError if one is missing, of an
    unsupported dtype or sliced); default: every tensor this reader can
    represent -- entries of other dtypes (e.g. the DT_STRING
    `_CHECKPOINTABLE_OBJECT_GRAPH` of object-based checkpoints) and sliced
    (partitioned) entries are skipped."""
  entries = list_variables(prefix)
  if names is not None:
    wanted = {}
    for n in names:
      e = entries[n]
      if e['dtype'] is None:
        raise ValueError('%s: unsupported tensor dtype enum %d' %
                         (n, e['dtype_enum']))
      if e['sliced']:
        raise ValueError('%s: sliced (partitioned) tensors are not supported' % n)
      wanted[n] = e
    entries = wanted
  else:
    entries = {n: e for n, e in entries.items()
               if e['dtype'] is not None and not e['sliced']}
  num_shards = 1 + max((e['shard'] for e in entries.values()), default=0)
  shards = {}
  out = {}
  for name, e in entries.items():
    shard = e['shard']
    if shard not in shards:
      # Shard count in the file name is not recorded per entry; probe.
      path = None
      for total in range(max(num_shards, 1), max(num_shards, 1) + 64):
        cand = '%s.data-%05d-of-%05d' % (prefix, shard, total)
        if os.path.exists(cand):
          path = cand
          break
      if path is None:
        raise FileNotFoundError('data shard %d of %s' % (shard, prefix))
      with open(path, 'rb') as f:
        shards[shard] = f.read()
    raw = shards[shard][e['offset']:e['offset'] + e['size']]
    arr = np.frombuffer(raw, dtype=e['dtype'])
    out[name] = arr.reshape(e['shape']).copy()
  return out
