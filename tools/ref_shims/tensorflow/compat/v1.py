class _T:

  def set_shape(self, *a, **k):
    pass


def placeholder(*a, **k):
  return _T()


def Variable(*a, **k):
  return _T()


class Session:
  pass


class Operation:
  pass


float32 = 'float32'
int64 = 'int64'
