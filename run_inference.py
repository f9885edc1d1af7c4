#!/usr/bin/env python3
"""Runs FFN inference within a bounding box on one MI355X (or one per rank).

Drop-in for the reference's run_inference.py (:38-56): same flags, same
text-format protos, same outputs (`seg-*.npz`, `counters.txt`).

  python run_inference.py \
      --inference_request="$(cat configs/inference_training_sample2.pbtxt)" \
      --bounding_box 'start { x:0 y:0 z:0 } size { x:250 y:250 z:250 }'

Under `torch.distributed.run` (one process per GPU) the bounding box is tiled
into overlapping sub-boxes dealt round-robin to the ranks
(ffn_amd/distributed.py); each rank writes its own `seg-*.npz` files.
"""

import argparse
import logging
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from ffn_amd import distributed as ffn_dist  # noqa: E402
from ffn_amd.inference import inference_flags  # noqa: E402
from ffn_amd.inference import request as req_lib  # noqa: E402
from ffn_amd.inference import runner as runner_lib  # noqa: E402


def main(argv=None):
  ap = argparse.ArgumentParser(description=__doc__)
  inference_flags.add_flags(ap)
  ap.add_argument('--bounding_box', required=True,
                  help='BoundingBox proto in text format (xyz start / size).')
  ap.add_argument('--subvolume_size', default='',
                  help='x,y,z size of the sub-boxes when sharding (default: '
                  'the whole bounding box on one rank).')
  ap.add_argument('--overlap', default='', help='x,y,z overlap of sub-boxes.')
  ap.add_argument('--batch_size', type=int, default=1)
  ap.add_argument('--conv_variant', type=int, default=None,
                  help='kernel behind the 32->32 convs (default: 9 = conv32mt for '
                  'single-FoV steps / conv32m for batched ones; 8 = conv32m; 2 = '
                  'exact f32 in the oracle\'s summation order; 6 = conv32d, '
                  'K-split); see DESIGN.md section 3')
  ap.add_argument('--assemble', default='',
                  help='With sharding: also assemble ONE global label volume '
                  '(RCCL all-reduce + union-find reconciliation of objects cut '
                  'by sub-box borders) and save it here (.npy, rank 0).  The '
                  "sub-boxes of a rank then advance concurrently on its GPU "
                  '(--batch_size of them per engine call).')
  ap.add_argument('--min_overlap_voxels', type=int,
                  default=ffn_dist.MIN_OVERLAP_VOXELS)
  ap.add_argument('--min_overlap_fraction', type=float,
                  default=ffn_dist.MIN_OVERLAP_FRACTION)
  args = ap.parse_args(argv)
  logging.basicConfig(level=logging.INFO)

  request = inference_flags.request_from_flags(args)
  os.makedirs(request.segmentation_output_dir, exist_ok=True)
  bbox = req_lib.parse_text(args.bounding_box, req_lib.BoundingBox())
  start_zyx = (bbox.start.z, bbox.start.y, bbox.start.x)
  size_zyx = (bbox.size.z, bbox.size.y, bbox.size.x)

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))

  runner = runner_lib.Runner(device_id=local_rank, conv_variant=args.conv_variant)
  runner.start(request, batch_size=args.batch_size,
               direct=True if args.assemble else None)

  if args.subvolume_size or world > 1:
    sub = ([int(v) for v in args.subvolume_size.split(',')][::-1]
           if args.subvolume_size else list(size_zyx))
    fov = runner._model_info.input_image_size[::-1]
    ov = ([int(v) for v in args.overlap.split(',')][::-1] if args.overlap else
          [int(v) for v in fov])
    if args.assemble:
      import numpy as np  # pylint:disable=g-import-not-at-top
      # the assembly (id offsets, cores into one volume, all-reduce over RCCL,
      # margin histograms, relabel) runs on the GPU that segmented
      import torch  # pylint:disable=g-import-not-at-top
      torch.cuda.set_device(local_rank)
      device = torch.device('cuda', local_rank)
      if world > 1:
        import torch.distributed as dist  # pylint:disable=g-import-not-at-top
        dist.init_process_group('nccl', device_id=device)
      merged, info = ffn_dist.segment_volume(
          runner, start_zyx, size_zyx, sub, ov, rank, world, device,
          batch_size=args.batch_size,
          min_overlap_voxels=args.min_overlap_voxels,
          min_overlap_fraction=args.min_overlap_fraction)
      if rank == 0:
        np.save(args.assemble, merged)
        logging.info('assembled %d sub-boxes, %d merge edges -> %s',
                     len(info['boxes']), len(info['edges']), args.assemble)
      mine = []
    else:
      boxes = ffn_dist.tile_volume(size_zyx, sub, ov)
      mine = ffn_dist.assign_round_robin(boxes, rank, world)
  else:
    boxes = ffn_dist.tile_volume(size_zyx, size_zyx, (0, 0, 0))
    mine = boxes

  for box in mine:
    corner = tuple(s + c for s, c in zip(start_zyx, box.corner))
    runner.run(corner, box.size)

  counter_path = os.path.join(request.segmentation_output_dir,
                              'counters.txt' if world == 1 else
                              'counters-%d.txt' % rank)
  if not os.path.exists(counter_path):
    runner.counters.dump(counter_path)
  runner.stop_executor()


if __name__ == '__main__':
  # (kernel arguments in device memory: see bench.py)
  os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
  main()
