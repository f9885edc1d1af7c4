"""`Runner`: orchestration of FFN inference runs on one MI355X.

Mirror of reference ffn/inference/runner.py (`start` :165-216, `make_canvas`
:307-414, `get_seed_policy` :416-431, `save_segmentation` :433-482, `run`
:484-544).  The TF session / graph / Saver of `_init_tf_model` (:116-163) are
replaced by `_init_hip_model`: a `ConvStack3DFFNModel` carrying the checkpoint
weights and a `HipBatchExecutor` that owns the GPU.

Reference defects NOT inherited: `run` no longer reads
`partial_segment_iters` unbound on a fresh run (:518-533); `save_segmentation`
writes the probability map only when the canvas tracks one (:480 passes None).
"""

from __future__ import annotations

import copy
import functools
import json
import logging
import os
from typing import Optional

import numpy as np

from ..training import model as ffn_model
from ..training.import_util import import_symbol
from . import align
from . import executor
from . import inference
from . import inference_utils
from . import movement
from . import seed
from . import storage
from .inference_utils import timer_counter


class _Box:
  """start / size / end in XYZ (the part of bounding_box.BoundingBox that
  MovementRestrictor reads: movement.py:283-284)."""

  def __init__(self, start, size):
    self.start = np.array(start)
    self.size = np.array(size)
    self.end = self.start + self.size


class Runner:
  """Helper for managing FFN inference runs."""

  ALL_MASKED = 1
  #: raw uint8 subvolumes are uploaded as they are and normalised on the device
  #: (ffn_canvas_create_u8); False = normalise on the host as the reference does
  DEVICE_U8 = True

  def __init__(self, device_id: int = 0, conv_variant: Optional[int] = None):
    """conv_variant: None = the engine's default (9: conv32mt for steps of one
    FoV; with batch_size > 1 the executor pins 8, conv32m, for every step); 2 =
    the exact-f32 kernel in the oracle's summation order (include/ffn_hip.h,
    DESIGN.md section 3)."""
    self.conv_variant = conv_variant
    self.counters = inference_utils.Counters()
    self.executor = None
    self._exec_interface = executor.ExecutorInterface()
    self.canvases = {}
    self.device_id = device_id
    self.request = None
    self._model_info: Optional[ffn_model.ModelInfo] = None
    self._image_volume = None
    self.init_seg_volume = None
    self._aligner = align.Aligner()
    self._direct = True

  def __del__(self):
    try:
      self.stop_executor()
    except Exception:  # pylint:disable=broad-except
      pass

  def stop_executor(self):
    """Shuts down the executor; no-op when none is active."""
    if self.executor is not None:
      try:
        self.executor.stop_server()
      except executor.TerminationException:
        pass
      self.executor = None

  def _get_model_class(self, model_name: str):
    return import_symbol(model_name)

  def _init_hip_model(self, request, batch_size: int):
    """Builds the model description, loads weights, creates the executor."""
    model_class = self._get_model_class(request.model_name)
    args = json.loads(request.model_args) if request.model_args else {}
    args['batch_size'] = batch_size
    model = model_class(**args)
    self._model_info = model.info
    with timer_counter(self.counters, 'restore-checkpoint'):
      if request.model_checkpoint_path:
        model.load_checkpoint(request.model_checkpoint_path)
      elif model.variables is None:
        raise ValueError('model_checkpoint_path is required')
    self.executor = executor.HipBatchExecutor(
        self._exec_interface, model, model.info, None, self.counters,
        batch_size, device_id=self.device_id)
    if self.conv_variant is not None:
      self.executor.engine.set_option('conv_variant', int(self.conv_variant))
    return model

  def start(self, request, batch_size: int = 1, session=None, direct=None,
            image_volume=None):
    """Opens input volumes and initialises the FFN engine.

    Args:
      request: InferenceRequest
      batch_size: max number of FoVs evaluated per engine call
      session: ignored (kept for signature compatibility with the reference)
      direct: None -> in-thread client when batch_size == 1 (no queue hop),
        else the reference's client/server threads; True / False to force
      image_volume: optional array-like overriding request.image
    """
    del session
    request = copy.deepcopy(request)
    self.request = request
    assert self.request.segmentation_output_dir
    os.makedirs(request.segmentation_output_dir, exist_ok=True)

    self.stop_executor()
    self._init_hip_model(request, batch_size)
    self._direct = (batch_size == 1) if direct is None else bool(direct)

    with timer_counter(self.counters, 'volstore-open'):
      if image_volume is not None:
        self._image_volume = image_volume
      else:
        self._image_volume = storage.decorated_volume(request.image)
      assert self._image_volume is not None
      if request.HasField('init_segmentation'):
        self.init_seg_volume = storage.decorated_volume(
            request.init_segmentation)
      else:
        self.init_seg_volume = None
      self._mask_volumes = {}
      self._shift_mask_volume = None
      if (request.HasField('shift_mask') and
          request.shift_mask.which_volume() is not None):
        self._shift_mask_volume = storage.decorated_volume(request.shift_mask)
      alignment_options = request.alignment_options
      if alignment_options.type != alignment_options.NO_ALIGNMENT:
        raise NotImplementedError('Only NO_ALIGNMENT is implemented')
      self._aligner = align.Aligner()

    if not self._direct:
      self.executor.start_server()

  def make_restrictor(self, corner, subvol_size, image, alignment):
    """Builds a MovementRestrictor from the request's masks, seed masks and
    shift mask (reference runner.py:218-305); None without any of them,
    ALL_MASKED if nothing is left to segment."""
    kwargs = {}
    request = self.request
    if len(request.masks):
      with timer_counter(self.counters, 'load-mask'):
        final_mask = storage.build_mask(request.masks, corner, subvol_size,
                                        self._mask_volumes, image, alignment)
        if np.all(final_mask):
          logging.info('Everything masked.')
          return self.ALL_MASKED
        kwargs['mask'] = final_mask
    if len(request.seed_masks):
      with timer_counter(self.counters, 'load-seed-mask'):
        seed_mask = storage.build_mask(request.seed_masks, corner, subvol_size,
                                       self._mask_volumes, image, alignment)
        if np.all(seed_mask):
          logging.info('All seeds masked.')
          return self.ALL_MASKED
        kwargs['seed_mask'] = seed_mask
    if self._shift_mask_volume is not None:
      with timer_counter(self.counters, 'load-shift-mask'):
        s = request.shift_mask_scale
        scale = np.array((1, s, s))
        shift_corner = np.array(corner) // scale
        shift_size = -(-np.array(subvol_size) // scale)
        shift_alignment = alignment.rescaled(np.array((1.0, 1.0, 1.0)) / scale)
        src_corner, src_size = shift_alignment.expand_bounds(
            shift_corner, shift_size, forward=False)
        src_corner, src_size = storage.clip_subvolume_to_bounds(
            src_corner, src_size, self._shift_mask_volume)
        src_end = np.array(src_corner) + np.array(src_size)
        expanded = np.asarray(self._shift_mask_volume[
            0:2, int(src_corner[0]):int(src_end[0]),
            int(src_corner[1]):int(src_end[1]),
            int(src_corner[2]):int(src_end[2])])
        shift_mask = np.array([
            shift_alignment.align_and_crop(src_corner, expanded[i],
                                           shift_corner, shift_size)
            for i in range(2)])
        shift_mask = alignment.transform_shift_mask(corner, s, shift_mask)
        if request.HasField('shift_mask_fov'):
          fov = request.shift_mask_fov
          shift_mask_fov = _Box(
              (fov.start.x, fov.start.y, fov.start.z),
              (fov.size.x, fov.size.y, fov.size.z))
        else:
          diameter = np.array(self._model_info.input_image_size)
          shift_mask_fov = _Box(-(diameter // 2), diameter)
        kwargs.update({
            'shift_mask': shift_mask,
            'shift_mask_fov': shift_mask_fov,
            'shift_mask_scale': request.shift_mask_scale,
            'shift_mask_threshold': request.shift_mask_threshold,
        })
    return movement.MovementRestrictor(**kwargs) if kwargs else None

  def make_canvas(self, corner, subvol_size, **canvas_kwargs):
    """Builds the Canvas for a subvolume (reference runner.py:307-414)."""
    subvol_counters = self.counters.get_sub_counters()
    with timer_counter(subvol_counters, 'load-image'):
      logging.info('Process subvolume: %r', corner)
      alignment = self._aligner.generate_alignment(corner, subvol_size)
      dst_corner, dst_size = alignment.expand_bounds(corner, subvol_size,
                                                     forward=True)
      src_corner, src_size = alignment.expand_bounds(dst_corner, dst_size,
                                                     forward=False)
      src_corner, src_size = storage.clip_subvolume_to_bounds(
          src_corner, src_size, self._image_volume)

      def get_data_3d(volume, start_zyx, size_zyx):
        slc = tuple(slice(int(s), int(s + n))
                    for s, n in zip(start_zyx, size_zyx))
        if volume.ndim == 4:
          slc = np.index_exp[0:1] + slc
        data = np.asarray(volume[slc])
        if data.ndim == 4:
          data = data.squeeze(axis=0)
        return data

      src_image = get_data_3d(self._image_volume, src_corner, src_size)

      def align_and_crop(image):
        return alignment.align_and_crop(src_corner, image, dst_corner, dst_size,
                                        forward=True)

      image = align_and_crop(src_image)
      logging.info('Image data loaded, shape: %r.', image.shape)

    restrictor = self.make_restrictor(dst_corner, dst_size, image, alignment)
    if restrictor == self.ALL_MASKED:
      return None, None

    exc = self.executor
    if exc is None:
      raise executor.TerminationException

    # (u8 -> f32 - mean) / stddev in f32, exactly as reference runner.py:383-385
    # -- on the device when the canvas lives there and the data is raw uint8 (the
    # image then stays 1 B / voxel in HBM and no f32 copy is made on the host)
    if (image.dtype == np.uint8 and self.DEVICE_U8 and
        hasattr(exc, 'engine') and 'storage_cls' not in canvas_kwargs):
      image = inference.NormalizedU8Image(image, self.request.image_mean,
                                          self.request.image_stddev)
    else:
      image = (image.astype(np.float32) -
               self.request.image_mean) / self.request.image_stddev

    canvas = inference.make_canvas(
        self._model_info,
        exc.get_client(subvol_counters, direct=self._direct),
        image,
        self.request.inference_options,
        counters=subvol_counters,
        restrictor=restrictor,
        movement_policy_fn=movement.get_policy_fn(self.request,
                                                  self._model_info),
        checkpoint_path=storage.checkpoint_path(
            self.request.segmentation_output_dir, corner),
        checkpoint_interval_sec=self.request.checkpoint_interval,
        corner_zyx=dst_corner,
        **canvas_kwargs)

    if self.request.HasField('init_segmentation'):
      end = np.array(src_corner) + np.array(src_size)
      canvas.init_segmentation_from_volume(self.init_seg_volume, src_corner,
                                           end, align_and_crop)
    return canvas, alignment

  def get_seed_policy(self, corner, subvol_size):
    """Seed policy factory (reference runner.py:416-431)."""
    policy_cls = getattr(seed, self.request.seed_policy)
    kwargs = {'corner': corner, 'subvol_size': subvol_size}
    if self.request.seed_policy_args:
      kwargs.update(json.loads(self.request.seed_policy_args))
    return functools.partial(policy_cls, **kwargs)

  def save_segmentation(self, canvas, alignment, target_path, prob_path):
    """Saves segmentation (+ probability map) (reference runner.py:433-482)."""

    def unalign_image(im3d):
      if alignment is None or im3d is None:
        return im3d
      return alignment.align_and_crop(canvas.corner_zyx, im3d,
                                      alignment.corner, alignment.size,
                                      forward=False)

    def unalign_origins(origins, canvas_corner):
      out_origins = dict()
      for key, value in origins.items():
        zyx = np.array(value.start_zyx) + canvas_corner
        zyx = alignment.transform(zyx[:, np.newaxis], forward=False).squeeze()
        zyx = zyx - canvas_corner
        out_origins[key] = value._replace(
            start_zyx=tuple(int(v) for v in zyx))
      return out_origins

    seg = np.array(np.asarray(canvas.segmentation))
    seg[seg < 0] = 0  # remove the -1 "excluded" markers
    corner = np.array(canvas.corner_zyx if canvas.corner_zyx is not None else
                      (0, 0, 0))
    storage.save_subvolume(
        unalign_image(seg), unalign_origins(canvas.origins, corner),
        target_path, request=self.request.SerializeToString(),
        counters=canvas.counters.dumps(), overlaps=canvas.overlaps)
    if canvas.seg_prob is not None:
      prob = unalign_image(np.asarray(canvas.seg_prob))
      with storage.atomic_file(prob_path) as fd:
        np.savez_compressed(fd, qprob=prob)

  def run(self, corner, subvol_size, reset_counters=True, **canvas_kwargs):
    """Runs FFN inference over a subvolume (reference runner.py:484-544)."""
    if reset_counters:
      self.counters.reset()
    corner = tuple(int(c) for c in corner)
    subvol_size = tuple(int(s) for s in subvol_size)
    out_dir = self.request.segmentation_output_dir
    seg_path = storage.segmentation_path(out_dir, corner)
    prob_path = storage.object_prob_path(out_dir, corner)
    cpoint_path = storage.checkpoint_path(out_dir, corner)
    if os.path.exists(seg_path):
      return None

    canvas, alignment = self.make_canvas(corner, subvol_size, **canvas_kwargs)
    if canvas is None:
      return None

    partial_segment_iters = 0
    if os.path.exists(cpoint_path):
      partial_segment_iters = canvas.restore_checkpoint(cpoint_path)

    if self.request.alignment_options.save_raw:
      image_path = storage.subvolume_path(out_dir, corner, 'align')
      with storage.atomic_file(image_path) as fd:
        np.savez_compressed(fd, im=canvas.image)

    self.canvases[corner] = canvas
    canvas.segment_all(seed_policy=self.get_seed_policy(corner, subvol_size),
                       partial_segment_iters=partial_segment_iters)
    self.save_segmentation(canvas, alignment, seg_path, prob_path)
    del self.canvases[corner]
    try:
      os.remove(cpoint_path)
    except OSError:
      pass
    return canvas

  def run_many(self, subvolumes, batch_size=None, reset_counters=True,
               save=True, window=None, on_done=None, keep_open=False, groups=1,
               max_steps_per_canvas=None):
    """Segments several subvolumes CONCURRENTLY on this GPU (BASELINE config C3).

    The reference gets this from one thread per subvolume calling `run()` on a
    shared Runner (doc/manual.md:89-97); here ONE thread advances all canvases
    in lock-free round-robin and every round is a single batched
    `ffn_canvas_step(n, ...)` (`inference.MultiCanvasDriver`).  Needs a Runner
    started with direct=True and batch_size >= the wanted concurrency.

    Like `run()`, a subvolume whose result exists is skipped, an existing
    `.cpoint` is restored first and removed after the result is written.  At
    most `window` canvases are open at once (each holds image + seed +
    segmentation in HBM and its image on the host): a subvolume is created when
    a slot frees up, and saved and closed the moment it finishes.

    Args:
      subvolumes: iterable of (corner_zyx, size_zyx); consumed LAZILY, one
        item whenever a canvas slot frees up -- a generator may decide what
        comes next only then (`distributed.BoxDealer`: sub-boxes dealt to the
        ranks of a job as they become free)
      batch_size: FoV steps per engine call (default: the engine's max batch)
      save: write each result like `run()` does (segmentation npz [+ prob])
      window: canvases open at once (default 2 x batch_size: two groups of
        batch_size keep two steps in flight)
      on_done: called as on_done(index, canvas) when subvolume `index` is
        finished and saved, before its canvas is closed
      keep_open: leave finished canvases open (the caller closes them)
      max_steps_per_canvas: bounded runs (benchmarks): a canvas is dropped
        after this many FoV steps
      groups: 2 = two groups of `batch_size` canvases, each advanced by its own
        host thread (`MultiCanvasDriver`): one group's steps run on the GPU
        while the other's ended segments are committed and re-seeded

    Returns:
      list of canvases (None where the output already existed / all masked),
      in the order of `subvolumes`; closed unless `keep_open`.
    """
    if not self._direct:
      raise ValueError('run_many needs Runner.start(..., direct=True)')
    if reset_counters:
      self.counters.reset()
    out_dir = self.request.segmentation_output_dir
    canvases = []
    meta = {}
    driver = inference.MultiCanvasDriver(
        self.executor.engine, batch_size, groups=groups,
        max_steps_per_canvas=max_steps_per_canvas)
    if window is None:
      window = 2 * driver.batch_size

    def jobs():
      for index, (corner, size) in enumerate(subvolumes):
        corner = tuple(int(c) for c in corner)
        size = tuple(int(v) for v in size)
        canvases.append(None)
        seg_path = storage.segmentation_path(out_dir, corner)
        if save and os.path.exists(seg_path):
          continue
        canvas, alignment = self.make_canvas(corner, size)
        if canvas is None:
          continue
        cpoint_path = storage.checkpoint_path(out_dir, corner)
        partial = 0
        if os.path.exists(cpoint_path):
          partial = canvas.restore_checkpoint(cpoint_path)
        canvases[index] = canvas
        self.canvases[corner] = canvas
        meta[id(canvas)] = (index, alignment, corner, cpoint_path)
        yield canvas, canvas._segment_all_gen(
            self.get_seed_policy(corner, size), partial)

    def finished(canvas):
      index, alignment, corner, cpoint_path = meta.pop(id(canvas))
      if save:
        self.save_segmentation(canvas, alignment,
                               storage.segmentation_path(out_dir, corner),
                               storage.object_prob_path(out_dir, corner))
        try:
          os.remove(cpoint_path)
        except OSError:
          pass
      del self.canvases[corner]
      if on_done is not None:
        on_done(index, canvas)
      if not keep_open and hasattr(canvas, 'close'):
        canvas.close()

    driver.run(jobs(), window=window, on_done=finished)
    #: the driver of the last run_many (its call / step / time tallies)
    self.last_driver = driver
    return canvases
