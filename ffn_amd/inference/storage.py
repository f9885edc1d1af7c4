"""Storage helpers of the inference path (mirror of reference ffn/inference/storage.py).

`NumpyArray` (:55-71), `decorated_volume` (:74-112), `atomic_file` (:116-134),
probability (de)quantisation (:137-151), `save_subvolume` (:154-171), path
helpers (:174-233), `clip_subvolume_to_bounds`, `load_segmentation`.
Output files are byte-compatible with the reference (`seg-*.npz` with keys
segmentation / origins / request / counters / overlaps; `.prob` with `qprob`).

Volume sources: the reference reads HDF5 / TensorStore / VolumeStore.  This
image has neither h5py nor tensorstore, so `decorated_volume` additionally
accepts `npy: "path.npy"` (memory-mapped) and uses h5py only if importable.
"""

from __future__ import annotations

import collections
from contextlib import contextmanager
import glob
import json
import os
import re
import tempfile

import numpy as np

from . import segmentation

OriginInfo = collections.namedtuple('OriginInfo',
                                    ['start_zyx', 'iters', 'walltime_sec'])


class NumpyArray(np.ndarray):
  """ndarray with a `clear` method (dense in-memory canvas storage)."""

  def __new__(cls, default_value=0, **kwargs):
    ret = super().__new__(cls, **kwargs)
    ret.default_value = default_value
    return ret

  def __init__(self, *args, **kwargs):
    del args, kwargs
    self.clear()

  def __array_finalize__(self, obj):
    self.default_value = getattr(obj, 'default_value', 0)

  def clear(self):
    self[...] = self.default_value


def decorated_volume(settings, **kwargs):
  """DecoratedVolume -> array-like with __getitem__, shape, ndim in (3, 4)."""
  del kwargs
  which = settings.which_volume()
  if which == 'npy':
    volume = np.load(settings.npy, mmap_mode='r')
  elif which == 'hdf5':
    path = settings.hdf5.split(':')
    if len(path) != 2:
      raise ValueError('hdf5 volume_path should be specified as file_path:'
                       'hdf5_internal_dataset_path.  Got: ' + settings.hdf5)
    try:
      import h5py  # pylint:disable=g-import-not-at-top
    except ImportError as e:
      raise NotImplementedError(
          'h5py is not available in this environment; convert the volume to '
          '.npy and use `image { npy: "..." }`') from e
    volume = h5py.File(path[0], 'r')[path[1]]
  elif which == 'tensorstore':
    raise NotImplementedError('tensorstore is not available.')
  elif which == 'volinfo':
    raise NotImplementedError('VolumeStore operations not available.')
  else:
    raise ValueError('A volume_path must be set.')
  if volume.ndim not in (3, 4):
    raise ValueError('Volume must be 3d or 4d.')
  return volume


@contextmanager
def atomic_file(path, mode='w+b'):
  """Atomically saves data to `path` (write to temp, then rename)."""
  d = os.path.dirname(os.path.abspath(path))
  os.makedirs(d, exist_ok=True)
  with tempfile.NamedTemporaryFile(mode=mode, dir=d, delete=False) as tmp:
    try:
      yield tmp
      tmp.flush()
    except BaseException:
      tmp.close()
      os.unlink(tmp.name)
      raise
  os.replace(tmp.name, path)


def quantize_probability(prob: np.ndarray) -> np.ndarray:
  """Quantises a probability map into bytes; 0 = NaN (never predicted)."""
  ret = np.digitize(prob, np.linspace(0.0, 1.0, 255))
  ret[np.isnan(prob)] = 0
  return ret.astype(np.uint8)


def dequantize_probability(prob: np.ndarray) -> np.ndarray:
  dq = 1.0 / 255
  ret = ((prob - 0.5) * dq).astype(np.float32)
  ret[prob == 0] = np.nan
  return ret


def save_subvolume(labels, origins, output_path, **misc_items):
  """Saves an FFN subvolume as .npz (ids reduced to the minimal uint type)."""
  seg = segmentation.reduce_id_bits(labels)
  os.makedirs(os.path.dirname(os.path.abspath(output_path)), exist_ok=True)
  with atomic_file(output_path) as fd:
    np.savez_compressed(fd, segmentation=seg, origins=origins, **misc_items)


def legacy_subvolume_path(output_dir, corner, suffix):
  return os.path.join(
      output_dir, 'seg-%s.%s' % ('_'.join([str(x) for x in corner[::-1]]),
                                 suffix))


def subvolume_path(output_dir, corner, suffix):
  """<dir>/<x>/<y>/seg-<x>_<y>_<z>.<suffix> for a (z, y, x) corner."""
  return os.path.join(
      output_dir, str(corner[2]), str(corner[1]),
      'seg-%s.%s' % ('_'.join([str(x) for x in corner[::-1]]), suffix))


def get_corner_from_path(path):
  match = re.search(r'(\d+)_(\d+)_(\d+).npz', os.path.basename(path))
  if match is None:
    raise ValueError('Unrecognized path: %s' % path)
  coord = tuple([int(x) for x in match.groups()])
  return coord[::-1]


def get_existing_corners(segmentation_dir):
  corners = []
  for path in glob.glob(os.path.join(segmentation_dir, 'seg-*_*_*.npz')):
    corners.append(get_corner_from_path(path))
  for path in glob.glob(os.path.join(segmentation_dir, '*/*/seg-*_*_*.npz')):
    corners.append(get_corner_from_path(path))
  return corners


def checkpoint_path(output_dir, corner):
  return subvolume_path(output_dir, corner, 'cpoint')


def segmentation_path(output_dir, corner):
  return subvolume_path(output_dir, corner, 'npz')


def object_prob_path(output_dir, corner):
  return subvolume_path(output_dir, corner, 'prob')


def legacy_segmentation_path(output_dir, corner):
  return legacy_subvolume_path(output_dir, corner, 'npz')


def legacy_object_prob_path(output_dir, corner):
  return legacy_subvolume_path(output_dir, corner, 'prob')


def get_existing_subvolume_path(segmentation_dir, corner, allow_cpoint=False):
  """Returns the path of an existing result (or checkpoint) for `corner`."""
  target_path = segmentation_path(segmentation_dir, corner)
  if os.path.exists(target_path):
    return target_path
  target_path = legacy_segmentation_path(segmentation_dir, corner)
  if os.path.exists(target_path):
    return target_path
  if allow_cpoint:
    target_path = checkpoint_path(segmentation_dir, corner)
    if os.path.exists(target_path):
      return target_path
  return None


def clip_subvolume_to_bounds(corner, size, volume):
  """Clips a (z, y, x) box to the bounds of `volume` (3d or 4d czyx)."""
  volume_size = np.array(volume.shape[-3:])
  corner = np.array(corner)
  size = np.array(size)
  start = np.maximum(corner, 0)
  end = np.minimum(corner + size, volume_size)
  return start, np.maximum(end - start, 0)


def load_origins(segmentation_dir, corner):
  target_path = get_existing_subvolume_path(segmentation_dir, corner, False)
  if target_path is None:
    raise ValueError('Segmentation not found: %s, %s' %
                     (segmentation_dir, corner))
  with np.load(target_path, allow_pickle=True) as data:
    return data['origins'].item()


def threshold_segmentation(segmentation_dir, corner, labels, threshold):
  """Zeroes voxels whose saved object probability is below `threshold`, in
  place (reference storage.py:380-411)."""
  prob_path = object_prob_path(segmentation_dir, corner)
  if not os.path.exists(prob_path):
    prob_path = legacy_object_prob_path(segmentation_dir, corner)
    if not os.path.exists(prob_path):
      raise ValueError('Cannot find probability map %s' % prob_path)
  with np.load(prob_path) as data:
    if 'qprob' not in data:
      raise ValueError('Invalid FFN probability map.')
    prob = dequantize_probability(data['qprob'])
    labels[prob < threshold] = 0


def build_mask(masks, corner, subvol_size, mask_volume_map=None, image=None,
               alignment=None):
  """Boolean exclusion mask of a subvolume (reference storage.py:323-411).

  Args:
    masks: iterable of MaskConfig messages
    corner, subvol_size: the subvolume (z, y, x)
    mask_volume_map: optional cache {serialized volume message: open volume}
    image: the subvolume's image (zyx), needed by `image` mask sources
    alignment: optional Alignment (identity if omitted)

  Returns:
    bool ndarray of `subvol_size`: the logical OR of every config's mask (each
    the OR of its channel masks, optionally inverted); None without configs.
  """
  from . import align  # pylint:disable=g-import-not-at-top
  final_mask = None
  if mask_volume_map is None:
    mask_volume_map = {}
  corner = tuple(int(c) for c in corner)
  subvol_size = tuple(int(c) for c in subvol_size)
  if alignment is None:
    alignment = align.Alignment(corner, subvol_size)  # identity
  src_corner, src_size = alignment.expand_bounds(corner, subvol_size,
                                                 forward=False)
  for config in masks:
    curr_mask = np.zeros(subvol_size, dtype=bool)
    source_type = config.WhichOneof('source')
    if source_type == 'coordinate_expression':
      # pylint:disable=eval-used,unused-variable,possibly-unused-variable
      z, y, x = np.mgrid[src_corner[0]:src_corner[0] + src_size[0],
                         src_corner[1]:src_corner[1] + src_size[1],
                         src_corner[2]:src_corner[2] + src_size[2]]
      bool_mask = eval(config.coordinate_expression.expression)  # as the reference
      # pylint:enable=eval-used,unused-variable,possibly-unused-variable
      curr_mask |= alignment.align_and_crop(src_corner, bool_mask, corner,
                                            subvol_size)
    else:
      if source_type == 'image':
        assert image is not None
        channels = config.image.channels
        mask = np.asarray(image)[np.newaxis, ...]
      elif source_type == 'volume':
        channels = config.volume.channels
        volume_key = config.volume.mask.SerializeToString()
        if volume_key not in mask_volume_map:
          mask_volume_map[volume_key] = decorated_volume(config.volume.mask)
        volume = mask_volume_map[volume_key]
        clipped_corner, clipped_size = clip_subvolume_to_bounds(
            src_corner, src_size, volume)
        clipped_end = np.array(clipped_corner) + np.array(clipped_size)
        sel = tuple(slice(int(a), int(b))
                    for a, b in zip(clipped_corner, clipped_end))
        if volume.ndim == 4:
          mask = np.asarray(volume[np.index_exp[:] + sel])
        else:  # a 3-d volume is its own single channel
          mask = np.asarray(volume[sel])[np.newaxis, ...]
      else:
        raise ValueError('Unsupported mask source: %s' % source_type)
      for chan_config in channels:
        channel_mask = mask[chan_config.channel, ...]
        channel_mask = alignment.align_and_crop(src_corner, channel_mask,
                                                corner, subvol_size)
        if len(chan_config.values):
          bool_mask = np.isin(channel_mask, list(chan_config.values))
        else:
          bool_mask = ((channel_mask >= chan_config.min_value) &
                       (channel_mask <= chan_config.max_value))
        if chan_config.invert:
          bool_mask = np.logical_not(bool_mask)
        curr_mask |= bool_mask
    if config.invert:
      curr_mask = np.logical_not(curr_mask)
    if final_mask is None:
      final_mask = curr_mask
    else:
      final_mask |= curr_mask
  return final_mask


def load_segmentation(segmentation_dir, corner, allow_cpoint=False,
                      threshold=None, split_cc=True, min_size=0,
                      mask_config=None):
  """Loads a saved segmentation subvolume (reference storage.py:414-488).

  Returns (uint64 zyx array, {segment id: origin info}).  Connected-component
  splitting and dust removal run on the GPU (segmentation.clean_up).
  """
  target_path = get_existing_subvolume_path(segmentation_dir, corner,
                                            allow_cpoint)
  if target_path is None:
    raise ValueError('Segmentation not found, %s, %r.' %
                     (segmentation_dir, corner))
  with np.load(target_path, allow_pickle=True) as data:
    if 'segmentation' not in data:
      raise ValueError('FFN NPZ file %s does not contain valid segmentation.' %
                       target_path)
    seg = data['segmentation']
    origins = data['origins'].item()
  if not np.any(seg):
    return np.zeros(seg.shape, dtype=np.uint64), {}
  output = seg.astype(np.uint64)
  if threshold is not None:
    threshold_segmentation(segmentation_dir, corner, output, threshold)
  if mask_config is not None:  # exclusion mask (reference storage.py:470-472)
    mask = build_mask(mask_config.masks, corner, seg.shape)
    if mask is not None:
      output[mask] = 0
  if split_cc or min_size:
    # (the reference passes min_size in clean_up's `connectivity` slot,
    # storage.py:476-478; 6-connectivity is what its callers get for the
    # default min_size = 0 and is what is used here)
    new_to_old = segmentation.clean_up(output, split_cc, min_size=min_size,
                                       return_id_map=True)
    new_origins = {}
    for new_id, old_id in new_to_old.items():
      if old_id in origins:
        new_origins[new_id] = origins[old_id]
    origins = new_origins
  return output, origins


def load_segmentation_from_source(source, corner):
  """load_segmentation configured by a SegmentationSource message
  (reference storage.py:491-511)."""
  kwargs = {}
  if source.HasField('threshold'):
    kwargs['threshold'] = source.threshold
  if source.HasField('split_cc'):
    kwargs['split_cc'] = source.split_cc
  if source.HasField('min_size'):
    kwargs['min_size'] = source.min_size
  if source.HasField('mask'):
    kwargs['mask_config'] = source.mask
  return load_segmentation(source.directory, corner, **kwargs)


def dump_json(obj) -> str:
  return json.dumps(obj, sort_keys=True)
