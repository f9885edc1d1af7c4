"""`import_symbol` (reference ffn/training/import_util.py:20-23)."""

import importlib


def import_symbol(specifier: str,
                  default_packages: str = 'ffn_amd.training.models'):
  """Resolves 'module.Class' (default package first, then absolute)."""
  module_path, symbol = specifier.rsplit('.', 1)
  last_err = None
  for candidate in (default_packages + '.' + module_path, module_path):
    try:
      mod = importlib.import_module(candidate)
      return getattr(mod, symbol)
    except (ImportError, AttributeError) as e:
      last_err = e
  raise ImportError('cannot resolve %r: %s' % (specifier, last_err))
