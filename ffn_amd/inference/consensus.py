"""Consensus between FFN segmentations.

Entry points and semantics of reference ffn/inference/consensus.py
(`compute_consensus_for_segmentations` :30-57, `compute_consensus` :60-96); the
voxel work (joint histogram of the two label volumes, relabelling) runs on the
GPU through `segmentation.split_segmentation_by_intersection`.
"""

import numpy as np

from . import request as request_lib
from . import segmentation
from . import storage

_SPLIT = request_lib.ConsensusRequest.CONSENSUS_SPLIT


def compute_consensus_for_segmentations(v1, v2, request):
  """Split consensus: every (id in v1, id in v2) overlap of at least
  `request.split_min_size` voxels becomes its own segment; the largest overlap
  of an id keeps that id.  Returns v1 narrowed to the smallest uint type."""
  if request.type != _SPLIT:
    raise ValueError('Unsupported mode: %s' % request.type)
  segmentation.split_segmentation_by_intersection(v1, v2,
                                                  request.split_min_size)
  # narrowed only now: the split mints ids above v1.max()
  return segmentation.reduce_id_bits(v1)


def compute_consensus(corner, request):
  """Consensus of the two subvolumes at `corner` (z, y, x) named by the
  request's SegmentationSources.

  Returns (uint zyx array, {segment id: origin info}) -- origins are those of
  segmentation1 for the ids that survive.
  """
  first, origins = storage.load_segmentation_from_source(request.segmentation1,
                                                         corner)
  second, _ = storage.load_segmentation_from_source(request.segmentation2,
                                                    corner)
  merged = compute_consensus_for_segmentations(first, second, request)
  alive = set(int(i) for i in np.unique(merged)) - {0}
  return merged, {k: v for k, v in origins.items() if int(k) in alive}
