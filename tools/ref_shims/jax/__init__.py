from . import numpy  # noqa


def jit(f):
  return f


def device_count():
  return 1
