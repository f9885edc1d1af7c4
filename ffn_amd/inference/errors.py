"""Exceptions of the inference path (reference ffn/inference/errors.py:18-20)."""


class TerminationException(Exception):
  """Raised to clients when the executor is shutting down."""
