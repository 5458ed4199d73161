"""Host-side NUMA lookup of the device feed (csrc/numa.cpp; the split being fed: mmap.rs:297-322), against a fake sysfs
tree — no GPU, no real topology needed: PCI bus id -> numa_node -> that node's cpulist, intersected with the CPUs the calling
thread may run on.  The GPU half (feeder affinity, where the pinned ring landed) is tests/test_gpu_numa.py."""
import os

import pytest


@pytest.fixture(scope="module")
def ia():
    import ibu_amd
    return ibu_amd


def _tree(root, bdf, node, cpulists):
    d = root / "bus" / "pci" / "devices" / bdf
    d.mkdir(parents=True)
    if node is not None:
        (d / "numa_node").write_text(f"{node}\n")
    for k, cl in cpulists.items():
        nd = root / "devices" / "system" / "node" / f"node{k}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    return root


def test_node_and_cpulist_of_a_pci_function(ia, tmp_path):
    mine = sorted(os.sched_getaffinity(0))
    lo, hi = mine[0], mine[-1]
    root = _tree(tmp_path / "sys", "0000:c1:00.0", 1, {0: "900-903", 1: f"{lo}-{hi},1000-1003"})
    node, cpulist, usable = ia.numa_of_pci("0000:c1:00.0", root)
    assert node == 1 and cpulist == f"{lo}-{hi},1000-1003"
    assert usable == len([c for c in mine if lo <= c <= hi])         # CPUs beyond this process's mask do not count
    assert ia.numa_of_pci("0000:C1:00.0", root)[0] == 1              # HIP spells bus ids in upper case, sysfs in lower


def test_a_platform_that_does_not_say(ia, tmp_path):
    root = _tree(tmp_path / "sys", "0000:05:00.0", -1, {0: "0-7"})
    assert ia.numa_of_pci("0000:05:00.0", root) == (-1, "", 0)       # numa_node = -1: single-node hosts, VMs
    assert ia.numa_of_pci("0000:06:00.0", root) == (-1, "", 0)       # no such device
    root2 = _tree(tmp_path / "sys2", "0000:07:00.0", None, {})
    assert ia.numa_of_pci("0000:07:00.0", root2) == (-1, "", 0)      # no numa_node file
    root3 = _tree(tmp_path / "sys3", "0000:08:00.0", 3, {})          # a node without a cpulist: node known, nothing to pin to
    assert ia.numa_of_pci("0000:08:00.0", root3) == (3, "", 0)
    root4 = _tree(tmp_path / "sys4", "0000:09:00.0", 0, {0: "0-3,x"})
    node, cpulist, usable = ia.numa_of_pci("0000:09:00.0", root4)    # a malformed list pins nothing
    assert node == 0 and usable == 0


def test_the_real_sysfs_never_fails(ia):
    node, cpulist, usable = ia.numa_of_pci("0000:00:00.0")
    assert node >= -1 and usable >= 0
