import importlib


def import_symbol(specifier, default_packages='ffn.training.models'):
  module_path, symbol = specifier.rsplit('.', 1)
  try:
    mod = importlib.import_module(default_packages + '.' + module_path)
  except ImportError:
    mod = importlib.import_module(module_path)
  return getattr(mod, symbol)
