"""Configuration messages of the inference path, without a protobuf runtime.

The reference configures inference through text-format protos
(reference ffn/inference/inference.proto:22-282 `DecoratedVolume`,
`InferenceOptions` :131-168, `InferenceRequest` :189-282;
ffn/utils/bounding_box.proto; flags in inference_flags.py:24-43).  The shipped
`inference_pb2.py` is pre-3.20 generated code that does not load under current
protobuf, so this module provides plain-Python message classes with the same
field names / defaults / `HasField` / `CopyFrom` semantics, a text-format
parser (`parse_text`) accepting the reference's .pbtxt files unchanged, and
`SerializeToString()` (text format, utf-8) for embedding the request in the
result .npz like the reference does (runner.py:468-475).
"""

from __future__ import annotations

import copy
import re
from typing import Any, Dict, Tuple

_SCALARS = (float, int, str, bool)


class _Repeated(list):
  """Repeated field: a list with protobuf's `add()` for message elements."""

  __slots__ = ('_ftype', '_owner')

  def __init__(self, ftype, items=()):
    super().__init__(items)
    self._ftype = ftype
    self._owner = None  # the message to mark present when an element is added

  def _touch(self):
    if self._owner is not None:
      self._owner._attach()

  def add(self, **kwargs):
    item = self._ftype(**kwargs)
    self.append(item)
    return item

  def append(self, item):
    super().append(item)
    self._touch()

  def extend(self, items):
    super().extend(items)
    self._touch()

  def __deepcopy__(self, memo):
    return _Repeated(self._ftype, (copy.deepcopy(v, memo) for v in self))

  def __reduce__(self):
    return (_Repeated, (self._ftype, list(self)))


class Message:
  """Minimal proto2-like message: typed fields, presence tracking."""

  # name -> (type, default, repeated)
  FIELDS: Dict[str, Tuple[Any, Any, bool]] = {}
  # oneof name -> member field names
  ONEOFS: Dict[str, Tuple[str, ...]] = {}

  def __init__(self, **kwargs):
    object.__setattr__(self, '_values', {})
    for k, v in kwargs.items():
      setattr(self, k, v)

  def __getattr__(self, name):
    fields = type(self).FIELDS
    if name not in fields:
      raise AttributeError(name)
    values = object.__getattribute__(self, '_values')
    if name in values:
      return values[name]
    ftype, default, repeated = fields[name]
    if repeated:
      values[name] = _Repeated(ftype)
      # adding to a repeated field of a not-yet-present sub-message makes the
      # sub-message present (protobuf semantics: c.image.channels.add())
      values[name]._owner = self
      return values[name]
    if isinstance(ftype, type) and issubclass(ftype, Message):
      # Reading an unset sub-message yields a default instance that becomes
      # present only when something inside it is set (proto2 semantics are
      # approximated: the instance is attached lazily on first write).
      sub = ftype()
      object.__setattr__(sub, '_parent', (self, name))
      return sub
    return default

  def __setattr__(self, name, value):
    fields = type(self).FIELDS
    if name not in fields:
      raise AttributeError('%s has no field %r' % (type(self).__name__, name))
    ftype, _, repeated = fields[name]
    if not repeated and ftype in _SCALARS:
      if ftype is float:
        value = float(value)
      elif ftype is int:
        value = int(value)
      elif ftype is bool:
        value = bool(value)
      else:
        value = str(value)
    elif repeated and not isinstance(value, _Repeated):
      value = _Repeated(ftype, value)
    self._values[name] = value
    for members in type(self).ONEOFS.values():
      if name in members:
        for other in members:
          if other != name:
            self._values.pop(other, None)
    self._attach()

  def _attach(self):
    parent = self.__dict__.get('_parent')
    if parent is not None:
      msg, name = parent
      object.__setattr__(self, '_parent', None)
      msg._values[name] = self
      msg._attach()

  def HasField(self, name):  # pylint:disable=invalid-name
    if name not in type(self).FIELDS:
      raise ValueError('unknown field %r' % name)
    return name in self._values

  def ClearField(self, name):  # pylint:disable=invalid-name
    self._values.pop(name, None)

  def WhichOneof(self, oneof):  # pylint:disable=invalid-name
    for member in type(self).ONEOFS[oneof]:
      if member in self._values:
        return member
    return None

  def CopyFrom(self, other):  # pylint:disable=invalid-name
    object.__setattr__(self, '_values', copy.deepcopy(other._values))

  def __deepcopy__(self, memo):
    new = type(self)()
    object.__setattr__(new, '_values', copy.deepcopy(self._values, memo))
    return new

  def __eq__(self, other):
    return type(self) is type(other) and self._values == other._values

  def to_text(self, indent=0) -> str:
    pad = '  ' * indent
    out = []
    for name, (ftype, _, repeated) in type(self).FIELDS.items():
      if name not in self._values:
        continue
      vals = self._values[name] if repeated else [self._values[name]]
      for v in vals:
        if isinstance(v, Message):
          out.append('%s%s {\n%s%s}\n' % (pad, name, v.to_text(indent + 1), pad))
        elif isinstance(v, str):
          esc = v.replace('\\', '\\\\').replace('"', '\\"').replace('\n', '\\n')
          out.append('%s%s: "%s"\n' % (pad, name, esc))
        elif isinstance(v, bool):
          out.append('%s%s: %s\n' % (pad, name, 'true' if v else 'false'))
        else:
          out.append('%s%s: %r\n' % (pad, name, v))
    return ''.join(out)

  def SerializeToString(self) -> bytes:  # pylint:disable=invalid-name
    return self.to_text().encode('utf-8')

  def ParseFromString(self, data) -> int:  # pylint:disable=invalid-name
    """Inverse of SerializeToString (which writes protobuf TEXT format: there
    is no protobuf wire codec in this package)."""
    object.__setattr__(self, '_values', {})
    if isinstance(data, (bytes, bytearray)):
      data = bytes(data).decode('utf-8')
    parse_text(data, self)
    return len(data)

  def __repr__(self):
    return '%s(\n%s)' % (type(self).__name__, self.to_text(1))


# ---------------------------------------------------------------------------
# Schemas
# ---------------------------------------------------------------------------


class Vector3j(Message):
  FIELDS = {'x': (int, 0, False), 'y': (int, 0, False), 'z': (int, 0, False)}


class BoundingBox(Message):
  """ffn/utils/bounding_box.proto: xyz start + size."""
  FIELDS = {'start': (Vector3j, None, False), 'size': (Vector3j, None, False)}


class DecoratedVolume(Message):
  FIELDS = {
      'volinfo': (str, '', False),
      'decorator_specs': (str, '', False),
      'hdf5': (str, '', False),
      'tensorstore': (str, '', False),
      # Extension of this implementation: a memory-mappable .npy file.
      'npy': (str, '', False),
  }
  ONEOFS = {'volume_path': ('volinfo', 'hdf5', 'tensorstore', 'npy')}

  def which_volume(self):
    return self.WhichOneof('volume_path')


class MaskChannelConfig(Message):
  """inference.proto:60-66: min_value <= channel <= max_value, or isin(values)."""
  FIELDS = {
      'channel': (int, 0, False),
      'min_value': (float, 0.0, False),
      'max_value': (float, 0.0, False),
      'values': (int, None, True),
      'invert': (bool, False, False),
  }


class ImageMaskOptions(Message):
  FIELDS = {'channels': (MaskChannelConfig, None, True)}


class VolumeMaskOptions(Message):
  FIELDS = {
      'mask': (DecoratedVolume, None, False),
      'channels': (MaskChannelConfig, None, True),
  }


class CoordinateExpressionOptions(Message):
  """inference.proto:77-86: a numpy expression over z, y, x index arrays (eval)."""
  FIELDS = {'expression': (str, '', False)}


class MaskConfig(Message):
  """inference.proto:96-103."""
  FIELDS = {
      'volume': (VolumeMaskOptions, None, False),
      'image': (ImageMaskOptions, None, False),
      'coordinate_expression': (CoordinateExpressionOptions, None, False),
      'invert': (bool, False, False),
  }
  ONEOFS = {'source': ('volume', 'image', 'coordinate_expression')}


class MaskConfigs(Message):
  FIELDS = {'masks': (MaskConfig, None, True)}


class InferenceOptions(Message):
  FIELDS = {
      'init_activation': (float, 0.0, False),
      'pad_value': (float, 0.0, False),
      'move_threshold': (float, 0.0, False),
      'disco_seed_threshold': (float, 0.0, False),
      'min_boundary_dist': (Vector3j, None, False),
      'segment_threshold': (float, 0.0, False),
      'min_segment_size': (int, 0, False),
  }

  def __setattr__(self, name, value):
    # proto `float` fields hold f32 values: round on assignment so that
    # thresholds behave exactly as in the reference (inference.py:189-195).
    if name in ('init_activation', 'pad_value', 'move_threshold',
                'disco_seed_threshold', 'segment_threshold'):
      import numpy as np  # pylint:disable=g-import-not-at-top
      value = float(np.float32(value))
    super().__setattr__(name, value)


class AlignmentOptions(Message):
  # inference.proto:171-179 (default NO_ALIGNMENT)
  UNKNOWN_ALIGNMENT = 0
  NO_ALIGNMENT = 1
  FIELDS = {'type': (int, 1, False), 'save_raw': (bool, False, False)}
  ENUMS = {'type': {'UNKNOWN_ALIGNMENT': 0, 'NO_ALIGNMENT': 1}}


class SegmentationSource(Message):
  """inference.proto:111-127."""
  FIELDS = {
      'directory': (str, '', False),
      'threshold': (float, 0.0, False),
      'split_cc': (bool, False, False),
      'min_size': (int, 0, False),
      'mask': (MaskConfigs, None, False),
  }


class ConsensusRequest(Message):
  """consensus.proto:22-37."""
  CONSENSUS_SPLIT = 2
  FIELDS = {
      'segmentation1': (SegmentationSource, None, False),
      'segmentation2': (SegmentationSource, None, False),
      'segmentation_output_dir': (str, '', False),
      'type': (int, 0, False),
      'split_min_size': (int, 0, False),
  }
  ENUMS = {'type': {'CONSENSUS_SPLIT': 2}}


class InferenceRequest(Message):
  FIELDS = {
      'image': (DecoratedVolume, None, False),
      'image_mean': (float, 0.0, False),
      'image_stddev': (float, 0.0, False),
      'reference_histogram': (str, '', False),
      'shift_mask': (DecoratedVolume, None, False),
      'shift_mask_fov': (BoundingBox, None, False),
      'shift_mask_scale': (int, 1, False),
      'shift_mask_threshold': (int, 4, False),
      'movement_policy_name': (str, '', False),
      'movement_policy_args': (str, '', False),
      'model_name': (str, '', False),
      'model_args': (str, '', False),
      'model_checkpoint_path': (str, '', False),
      'batch_size': (int, 1, False),
      'concurrent_requests': (int, 1, False),
      'inference_options': (InferenceOptions, None, False),
      'segmentation_output_dir': (str, '', False),
      'checkpoint_interval': (int, 0, False),
      'seed_policy': (str, 'PolicyPeaks', False),
      'seed_policy_args': (str, '', False),
      'alignment_options': (AlignmentOptions, None, False),
      'init_segmentation': (DecoratedVolume, None, False),
      # exclusion masks (inference.proto:198-205)
      'masks': (MaskConfig, None, True),
      'seed_masks': (MaskConfig, None, True),
  }

  def __setattr__(self, name, value):
    if name in ('image_mean', 'image_stddev'):
      import numpy as np  # pylint:disable=g-import-not-at-top
      value = float(np.float32(value))
    super().__setattr__(name, value)


class ResegmentationPoint(Message):
  """inference.proto:284-293; `id_b` unset = endpoint extension request."""
  FIELDS = {
      'id_a': (int, 0, False),
      'id_b': (int, 0, False),
      'point': (Vector3j, None, False),
  }


class ResegmentationRequest(Message):
  """inference.proto:295-341."""
  FIELDS = {
      'inference': (InferenceRequest, None, False),
      'points': (ResegmentationPoint, None, True),
      'radius': (Vector3j, None, False),
      'output_directory': (str, '', False),
      'subdir_digits': (int, 0, False),
      'max_retry_iters': (int, 1, False),
      'exclusion_radius': (Vector3j, None, False),
      'init_exclusion_radius': (Vector3j, None, False),
      'segment_recovery_fraction': (float, 0.0, False),
      'terminate_early': (bool, False, False),
      'analysis_radius': (Vector3j, None, False),
  }



# ---------------------------------------------------------------------------
# Text-format parser
# ---------------------------------------------------------------------------

_TOKEN = re.compile(
    r'\s*(?:(#[^\n]*)|([A-Za-z_][A-Za-z0-9_\.]*)|("(?:\\.|[^"\\])*"|'
    r"'(?:\\.|[^'\\])*')|([-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|inf|nan)f?)|([{}<>:;,\[\]]))")


def _tokenize(text):
  pos = 0
  out = []
  n = len(text)
  while pos < n:
    m = _TOKEN.match(text, pos)
    if not m:
      if text[pos:].strip() == '':
        break
      raise ValueError('text-format parse error near %r' % text[pos:pos + 30])
    pos = m.end()
    if m.group(1) is not None:
      continue
    if m.group(2) is not None:
      out.append(('id', m.group(2)))
    elif m.group(3) is not None:
      out.append(('str', _unescape(m.group(3)[1:-1])))
    elif m.group(4) is not None:
      out.append(('num', m.group(4)))
    else:
      out.append(('sym', m.group(5)))
  return out


def _unescape(s):
  return (s.encode('latin-1', 'backslashreplace').decode('unicode_escape')
          if '\\' in s else s)


def _parse_fields(msg: Message, toks, i, closer):
  fields = type(msg).FIELDS
  while i < len(toks):
    kind, val = toks[i]
    if kind == 'sym' and val == closer:
      return i + 1
    if kind == 'sym' and val in ';,':
      i += 1
      continue
    if kind != 'id':
      raise ValueError('expected field name, got %r' % (val,))
    name = val
    if name not in fields:
      raise ValueError('%s has no field %r' % (type(msg).__name__, name))
    ftype, _, repeated = fields[name]
    i += 1
    if i < len(toks) and toks[i] == ('sym', ':'):
      i += 1
    if isinstance(ftype, type) and issubclass(ftype, Message):
      if toks[i] not in (('sym', '{'), ('sym', '<')):
        raise ValueError('expected { after %s' % name)
      sub = ftype()
      i = _parse_fields(sub, toks, i + 1, '}' if toks[i][1] == '{' else '>')
      if repeated:
        getattr(msg, name).append(sub)
      else:
        msg._values[name] = sub
    else:
      kind, val = toks[i]
      i += 1
      if ftype is str:
        if kind != 'str':
          raise ValueError('field %s expects a string' % name)
        while i < len(toks) and toks[i][0] == 'str':  # adjacent literals concat
          val += toks[i][1]
          i += 1
        value = val
      elif ftype is bool:
        value = str(val).lower() in ('true', '1', 't')
      elif ftype is int:
        value = int(val) if kind == 'num' else _enum(type(msg), name, val)
      else:
        value = float(val.rstrip('f'))
      if repeated:
        getattr(msg, name).append(value)
      else:
        setattr(msg, name, value)
  if closer is not None:
    raise ValueError('unterminated message')
  return i


def _enum(msg_cls, name, val):
  enums = getattr(msg_cls, 'ENUMS', {}).get(name, {})
  if val in enums:
    return enums[val]
  raise ValueError('unsupported enum value %r for %s.%s' %
                   (val, msg_cls.__name__, name))


def parse_text(text: str, msg: Message) -> Message:
  """Parses protobuf text format into `msg` (merging) and returns it."""
  toks = _tokenize(text)
  _parse_fields(msg, toks, 0, None)
  return msg


def request_from_text(request_text: str, options_text: str = ''):
  """Equivalent of inference_flags.request_from_flags (inference_flags.py:38-43)."""
  request = InferenceRequest()
  if request_text:
    parse_text(request_text, request)
  if options_text:
    options = InferenceOptions()
    parse_text(options_text, options)
    request.inference_options = options
  return request
