"""Flags -> InferenceRequest (mirror of reference ffn/inference/inference_flags.py:24-43).

The reference uses absl flags holding text-format protos; absl is not a
dependency here, so the same two flags are argparse arguments."""

from . import request as req_lib


def add_flags(parser):
  parser.add_argument('--inference_request', default='',
                      help='InferenceRequest proto in text format.')
  parser.add_argument('--inference_options', default='',
                      help='InferenceOptions proto in text format; overrides '
                      'the options inside --inference_request.')


def options_from_flags(args):
  options = req_lib.InferenceOptions()
  if args.inference_options:
    req_lib.parse_text(args.inference_options, options)
  return options


def request_from_flags(args):
  return req_lib.request_from_text(args.inference_request,
                                   args.inference_options)
