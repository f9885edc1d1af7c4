"""Counters and timers -- the source of the benchmark metric.

Mirror of reference ffn/inference/inference_utils.py:32-198: `StatCounter`,
`Counters` (per-subvolume counters propagate to the parent), `timer_counter`
(`<name>-calls`, `<name>-time-ms`), `TimedIter`.

FoV-steps/sec = `update_at-calls` / `segment_all-time-ms`; voxels/sec =
`voxels-segmented` / the same wall time (reference inference.py:398,555,648).

One deliberate deviation: the reference truncates every increment with
`int(x)` (inference_utils.py:61), so sub-millisecond timer intervals
accumulate as 0.  Here increments are accumulated exactly and truncated only
when dumped (`dump` / `dumps` still emit integers, as the reference does).
"""

from __future__ import annotations

import contextlib
import json
import os
import tempfile
import threading
import time

MSEC_IN_SEC = 1000


# pylint: disable=invalid-name
class StatCounter:
  """Counter with the MR counter interface (Increment / IncrementBy / Set)."""

  def __init__(self, update, name, parent=None):
    self._counter = 0
    self._update = update
    self._lock = threading.Lock()
    self._parent = parent
    self.name = name

  def Increment(self):
    self.IncrementBy(1)

  def IncrementBy(self, x, export=True):
    with self._lock:
      self._counter += x
      self._update()
    if self._parent is not None:
      self._parent.IncrementBy(x)

  def Set(self, x, export=True):
    self.IncrementBy(x - self._counter, export=export)

  def __repr__(self):
    return 'StatCounter(total=%g)' % (self.value)

  @property
  def value(self):
    return self._counter


# pylint: enable=invalid-name


class Counters:
  """Container for counters; sub-containers forward to their parent."""

  def __init__(self, parent=None):
    self._lock = threading.Lock()
    self.reset()
    self.parent = parent

  def reset(self):
    with self._lock:
      self._counters = {}
    self._last_update = 0

  def __getitem__(self, name: str) -> StatCounter:
    return self.get(name)

  def get(self, name: str, **kwargs) -> StatCounter:
    with self._lock:
      c = self._counters.get(name)
      if c is None:
        c = self._make_counter(name, **kwargs)
        self._counters[name] = c
      return c

  def __iter__(self):
    return iter(list(self._counters.items()))

  def _make_counter(self, name: str, **kwargs) -> StatCounter:
    del kwargs
    parent = self.parent.get(name) if self.parent is not None else None
    return StatCounter(self.update_status, name, parent)

  def update_status(self):
    pass

  def get_sub_counters(self):
    return Counters(self)

  def dump(self, filename: str):
    d = os.path.dirname(os.path.abspath(filename))
    with tempfile.NamedTemporaryFile('w', dir=d, delete=False) as fd:
      for name, counter in sorted(self._counters.items()):
        fd.write('%s: %d\n' % (name, counter.value))
      tmp = fd.name
    os.replace(tmp, filename)

  def dumps(self) -> str:
    state = {name: int(counter.value) for name, counter in self._counters.items()}
    return json.dumps(state)

  def loads(self, encoded_state: str):
    state = json.loads(encoded_state)
    for name, value in state.items():
      self[name].Set(value, export=False)


@contextlib.contextmanager
def timer_counter(counters: Counters, name: str, export=True,
                  increment: int = 1):
  """Counts calls and milliseconds spent inside the context."""
  assert isinstance(counters, Counters)
  counter = counters.get(name + '-calls', export=export)
  timer = counters.get(name + '-time-ms', export=export)
  start_time = time.time()
  try:
    yield timer, counter
  finally:
    counter.IncrementBy(increment)
    timer.IncrementBy((time.time() - start_time) * MSEC_IN_SEC)


class TimedIter:
  """Wraps an iterator with a timing counter."""

  def __init__(self, it, counters, counter_name):
    self.it = it
    self.counters = counters
    self.counter_name = counter_name

  def __iter__(self):
    return self

  def __next__(self):
    with timer_counter(self.counters, self.counter_name):
      ret = next(self.it)
    return ret

  def next(self):
    return self.__next__()
