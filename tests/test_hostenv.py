"""ffn_amd/hostenv.py: the process that drives a GPU is pinned to that GPU's NUMA
node (sysfs numa_node / local_cpulist of its PCI function)."""
import os

from ffn_amd import hostenv


def _fake_sysfs(tmp_path, node, cpulist):
  d = tmp_path / '0000:05:00.0'
  d.mkdir()
  (d / 'numa_node').write_text('%d\n' % node)
  (d / 'local_cpulist').write_text(cpulist + '\n')
  return str(tmp_path)


def test_cpulist_parser():
  assert hostenv._parse_cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
  assert hostenv._parse_cpulist('') == set()


def test_bind_to_gpu_node(tmp_path, monkeypatch):
  monkeypatch.setenv('FFN_AMD_NUMA_BIND', '1')
  monkeypatch.setattr(hostenv, 'gpu_pci_address', lambda i: '0000:05:00.0')
  before = os.sched_getaffinity(0)
  allowed = sorted(before)
  try:
    # the node's CPUs that this process may use: the first one (and one that
    # does not exist here)
    sysfs = _fake_sysfs(tmp_path, 1, '%d,99999' % allowed[0])
    assert hostenv.gpu_local_cpus(0, sysfs) == (1, {allowed[0], 99999})
    rec = hostenv.bind_to_gpu_node(0, sysfs)
    assert rec == {'numa_node': 1, 'cpus': 1, 'bound': True}
    assert os.sched_getaffinity(0) == {allowed[0]}
  finally:
    os.sched_setaffinity(0, before)


def test_bind_is_skipped_quietly(tmp_path, monkeypatch):
  monkeypatch.setenv('FFN_AMD_NUMA_BIND', '1')
  monkeypatch.setattr(hostenv, 'gpu_pci_address', lambda i: '0000:05:00.0')
  before = os.sched_getaffinity(0)
  # no NUMA information (single-node hosts report -1)
  sysfs = _fake_sysfs(tmp_path, -1, '0-3')
  rec = hostenv.bind_to_gpu_node(0, sysfs)
  assert rec['bound'] is False and 'sysfs' in rec['why']
  # a device sysfs does not know
  monkeypatch.setattr(hostenv, 'gpu_pci_address', lambda i: '0000:99:00.0')
  assert hostenv.bind_to_gpu_node(0, sysfs)['bound'] is False
  # not asked for: the default
  monkeypatch.setattr(hostenv, 'gpu_pci_address', lambda i: '0000:05:00.0')
  monkeypatch.delenv('FFN_AMD_NUMA_BIND')
  (tmp_path / 'b').mkdir()
  rec = hostenv.bind_to_gpu_node(0, _fake_sysfs(tmp_path / 'b', 1, '0-3'))
  assert rec['bound'] is False and 'FFN_AMD_NUMA_BIND' in rec['why']
  assert os.sched_getaffinity(0) == before
