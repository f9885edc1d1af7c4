"""Consensus between FFN segmentations (reference ffn/inference/consensus.py)."""

import logging

import numpy as np

from . import request as request_lib
from . import segmentation
from . import storage


def compute_consensus_for_segmentations(v1, v2, request):
  """Split consensus of two segmentations (consensus.py:30-57): the
  intersection is computed on the GPU, ids are then narrowed."""
  if request.type == request_lib.ConsensusRequest.CONSENSUS_SPLIT:
    segmentation.split_segmentation_by_intersection(v1, v2,
                                                    request.split_min_size)
    v1 = segmentation.reduce_id_bits(v1)
  else:
    raise ValueError('Unsupported mode: %s' % request.type)
  return v1


def compute_consensus(corner, request):
  """Consensus segmentation between two FFN subvolumes (consensus.py:60-96).

  Returns (uint array zyx, {segment id: origin info of segmentation1}).
  """
  v1, v1_origins = storage.load_segmentation_from_source(
      request.segmentation1, corner)
  logging.info('consensus: v1 data loaded')
  v2, _ = storage.load_segmentation_from_source(request.segmentation2, corner)
  logging.info('consensus: v2 data loaded')
  v1 = compute_consensus_for_segmentations(v1, v2, request)
  relabeled_origins = {}
  for seg_id in np.unique(v1):
    if seg_id == 0:
      continue
    if seg_id in v1_origins:
      relabeled_origins[seg_id] = v1_origins[seg_id]
  return v1, relabeled_origins
