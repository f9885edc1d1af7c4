class _Flag:

  def __init__(self, default):
    self.value = default


def _define(name, default, *a, **k):
  return _Flag(default)


DEFINE_boolean = DEFINE_bool = DEFINE_string = DEFINE_integer = _define
DEFINE_float = DEFINE_list = DEFINE_enum = DEFINE_multi_string = _define
FLAGS = None
