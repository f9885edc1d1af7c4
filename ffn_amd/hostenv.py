"""Where the process that drives a GPU runs, and where its launches' arguments live.

A single-FoV step is 25 kernel launches in a dependent chain, ~8 us apart: every
one of them is an AQL packet and a kernel-argument block the GPU's command
processor fetches from HOST memory before the kernel can start.  The same
bench.py command has measured 8.4 us per launch on some boxes of one pool and
13 - 14 us on others (profiles/r03_bench_loaded_host.json,
r03_ab_dev_kernarg_slow_launch_box.txt).  What helps on the slow ones is
HIP_FORCE_DEV_KERNARG=1 -- kernel arguments in device memory: 13.9 -> 10.5 us
per launch there, nothing measurable on the fast ones -- which the HIP runtime
reads when it starts, so bench.py and run_inference.py set it (as a default)
before anything initialises HIP; a host application embedding the library
should do the same.

`bind_to_gpu_node` pins the calling process to the CPUs of the GPU's own NUMA
node (call it before the engine -- its stream, hence the runtime's queues -- is
created).  On the one slow box it was tried on it changed nothing (2,848 against
2,843 FoV-steps/s), so it is OFF unless FFN_AMD_NUMA_BIND=1 asks for it: the
usual placement for one process per GPU on a two-socket, eight-GPU node, to be
measured there.  Nothing here is needed for correctness; every failure to find
or apply the binding is reported in the returned record and otherwise ignored.
"""

import os


def _parse_cpulist(text):
  cpus = set()
  for part in text.strip().split(','):
    if not part:
      continue
    lo, _, hi = part.partition('-')
    cpus.update(range(int(lo), int(hi or lo) + 1))
  return cpus


def gpu_pci_address(device_index: int):
  """'dddd:bb:dd.f' of HIP device `device_index` (torch's view of it)."""
  import torch
  p = torch.cuda.get_device_properties(device_index)
  return '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)


def gpu_local_cpus(device_index: int, sysfs='/sys/bus/pci/devices'):
  """(numa_node, set of CPU ids local to the GPU), or (None, None)."""
  try:
    base = os.path.join(sysfs, gpu_pci_address(device_index))
    with open(os.path.join(base, 'numa_node')) as f:
      node = int(f.read().strip())
    with open(os.path.join(base, 'local_cpulist')) as f:
      cpus = _parse_cpulist(f.read())
    return (node, cpus) if node >= 0 and cpus else (None, None)
  except (OSError, ValueError, RuntimeError, AssertionError):
    return None, None


def bind_to_gpu_node(device_index: int, sysfs='/sys/bus/pci/devices') -> dict:
  """Restricts the calling process (and the threads it starts from here on) to
  the CPUs of the GPU's NUMA node that its current affinity mask allows.
  Returns what was done: {'numa_node', 'cpus', 'bound'} (+ 'why' if not)."""
  if os.environ.get('FFN_AMD_NUMA_BIND', '0') != '1':
    return {'numa_node': None, 'cpus': 0, 'bound': False,
            'why': 'not requested (FFN_AMD_NUMA_BIND=1 turns it on)'}
  node, cpus = gpu_local_cpus(device_index, sysfs)
  if cpus is None:
    return {'numa_node': None, 'cpus': 0, 'bound': False,
            'why': 'no numa_node / local_cpulist for the device in sysfs'}
  try:
    allowed = os.sched_getaffinity(0)
    want = cpus & allowed
    if not want:
      return {'numa_node': node, 'cpus': 0, 'bound': False,
              'why': 'none of the node\'s CPUs is in this process\' affinity mask'}
    if want != allowed:
      os.sched_setaffinity(0, want)
    return {'numa_node': node, 'cpus': len(want), 'bound': True}
  except (OSError, AttributeError) as e:
    return {'numa_node': node, 'cpus': 0, 'bound': False, 'why': repr(e)}
