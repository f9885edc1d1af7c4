"""Counters and timers -- the source of the benchmark metric.

API-compatible with reference ffn/inference/inference_utils.py:32-198
(`StatCounter.Increment/IncrementBy/Set/value`, `Counters[...]`,
`get_sub_counters`, `dump/dumps/loads`, `timer_counter`, `TimedIter`) because
canvases, executors and result files use it; the implementation is this
repository's own and tuned for the per-FoV-step path (a slotted counter, a
class-based timer context instead of a generator, one lock per container).

FoV-steps/sec = `update_at-calls` / `segment_all-time-ms`; voxels/sec =
`voxels-segmented` / the same wall time (reference inference.py:398,555,648).

One deliberate deviation: the reference truncates every increment with
`int(x)` (inference_utils.py:61), so sub-millisecond timer intervals
accumulate as 0.  Here increments are accumulated exactly and truncated only
when written out (`dump` / `dumps` still emit integers, as the reference does).
"""

from __future__ import annotations

import json
import os
import tempfile
import threading
import time

MSEC_IN_SEC = 1000


class StatCounter:
  """One named tally; increments also flow into the parent container's tally
  of the same name (a subvolume's counters roll up into the runner's)."""

  __slots__ = ('name', '_total', '_parent', '_lock')

  def __init__(self, update=None, name='', parent=None):
    del update  # status export hook of the reference: nothing to export here
    self.name = name
    self._total = 0
    self._parent = parent
    self._lock = threading.Lock()

  # pylint: disable=invalid-name
  def IncrementBy(self, x, export=True):
    del export
    node = self
    while node is not None:  # this tally, then every ancestor's
      with node._lock:
        node._total += x
      node = node._parent

  def Increment(self):
    self.IncrementBy(1)

  def Set(self, x, export=True):
    self.IncrementBy(x - self._total, export)
  # pylint: enable=invalid-name

  @property
  def value(self):
    return self._total

  def __repr__(self):
    return 'StatCounter(total=%g)' % self._total


class Counters:
  """Name -> StatCounter, created on first use.  `get_sub_counters()` makes a
  child container whose tallies roll up into this one."""

  def __init__(self, parent=None):
    self.parent = parent
    self._guard = threading.Lock()
    self._tallies = {}

  def reset(self):
    with self._guard:
      self._tallies = {}

  def get(self, name: str, **kwargs) -> StatCounter:
    del kwargs
    tally = self._tallies.get(name)
    if tally is None:
      with self._guard:
        tally = self._tallies.get(name)
        if tally is None:
          up = self.parent.get(name) if self.parent is not None else None
          tally = self._tallies[name] = StatCounter(None, name, up)
    return tally

  __getitem__ = get

  def __iter__(self):
    return iter(list(self._tallies.items()))

  def update_status(self):
    """Hook of the reference (periodic status export); intentionally empty."""

  def get_sub_counters(self):
    return Counters(self)

  def _snapshot(self):
    return {name: int(t.value) for name, t in sorted(self._tallies.items())}

  def dump(self, filename: str):
    """`name: value` lines, written atomically (reference :139-143)."""
    folder = os.path.dirname(os.path.abspath(filename))
    with tempfile.NamedTemporaryFile('w', dir=folder, delete=False) as fd:
      fd.writelines('%s: %d\n' % kv for kv in self._snapshot().items())
    os.replace(fd.name, filename)

  def dumps(self) -> str:
    return json.dumps(self._snapshot())

  def loads(self, encoded_state: str):
    for name, value in json.loads(encoded_state).items():
      self.get(name).Set(value, export=False)


class timer_counter:  # pylint: disable=invalid-name
  """`with timer_counter(counters, 'x'):` adds `increment` to `x-calls` and the
  elapsed milliseconds to `x-time-ms` (reference :147-175)."""

  __slots__ = ('_calls', '_timer', '_increment', '_t0')

  def __init__(self, counters: Counters, name: str, export=True,
               increment: int = 1):
    del export
    assert isinstance(counters, Counters)
    self._calls = counters.get(name + '-calls')
    self._timer = counters.get(name + '-time-ms')
    self._increment = increment

  def __enter__(self):
    self._t0 = time.time()
    return self._timer, self._calls

  def __exit__(self, exc_type, exc, tb):
    self._calls.IncrementBy(self._increment)
    self._timer.IncrementBy((time.time() - self._t0) * MSEC_IN_SEC)
    return False


class TimedIter:
  """Iterator proxy that charges the time spent producing each item to
  `<counter_name>-calls` / `-time-ms` (reference :178-198)."""

  def __init__(self, it, counters, counter_name):
    self.it = it
    self.counters = counters
    self.counter_name = counter_name

  def __iter__(self):
    return self

  def __next__(self):
    with timer_counter(self.counters, self.counter_name):
      return next(self.it)

  next = __next__
