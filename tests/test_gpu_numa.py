"""Where the device feed runs (option "numa"; the split being fed: mmap.rs:297-322): the context knows the NUMA node its GPU
hangs off, the stream's producer thread (and so the feeders it starts) runs on that node's CPUs, and the context can say where
the pinned ring's pages landed.  On a box whose platform does not say (node -1) everything must behave as with the option off."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ia():
    import ibu_amd
    return ibu_amd


def _cpus(cpulist):
    out = set()
    for part in cpulist.split(","):
        if part:
            a, _, b = part.partition("-")
            out.update(range(int(a), int(b or a) + 1))
    return out


def _feed_threads():
    """{tid: allowed CPUs} of the library's producer threads of this process (named ibu-feed)."""
    found = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/comm") as f:
                if f.read().strip() != "ibu-feed":
                    continue
            found[int(tid)] = os.sched_getaffinity(int(tid))
        except OSError:
            pass
    return found


def _file(oracle, path, n):
    recs = oracle.generate(0x1B00006, 0, n, 16, 12)
    w = oracle.Writer(header=oracle.header_new(16, 12), path=str(path))
    w.write_batch(recs)
    w.finish()
    w.drop()
    return recs


def test_context_reports_the_devices_node_consistently_with_sysfs(ia):
    ctx = ia.Context(0)
    info = ctx.numa()
    assert info["mode"] == 1 and info["pci_bus_id"]
    node, cpulist, usable = ia.numa_of_pci(info["pci_bus_id"])
    assert (info["node"], info["cpulist"], info["usable_cpus"]) == (node, cpulist, usable)
    assert info["ring_node"] == -1 and not info["ring_placed"]       # no ring yet
    ctx.close()


@pytest.mark.parametrize("mode", [1, 0])
def test_feeder_affinity_and_ring_placement(ia, oracle, tmp_path, mode):
    p = tmp_path / "n.ibu"
    n = 300_000
    recs = _file(oracle, p, n)
    ctx = ia.Context(0)
    ctx.set_option("numa", mode)
    mine = os.sched_getaffinity(0)
    m = ia.MmapReader.new(p)
    s = m.device_stream(ctx, ring={"slots": 3, "slot_records": 65536, "feeder_threads": 4})
    info = ctx.numa()
    feeders = _feed_threads()
    assert len(feeders) == 1
    allowed = next(iter(feeders.values()))
    if mode == 1 and info["node"] >= 0 and info["usable_cpus"] > 0:
        assert allowed <= _cpus(info["cpulist"]) and len(allowed) == info["usable_cpus"]
        if info["ring_placed"] and info["ring_node"] >= 0:           # the kernel took the policy and says where the pages are
            assert info["ring_node"] == info["node"]
    else:
        assert allowed == mine                                        # nothing known / option off: nothing pinned
        assert not info["ring_placed"]
    got = []
    for b in s:
        with b:
            got.append(b.download().copy())
    st = s.stats()
    s.close()
    assert np.concatenate(got).tobytes() == recs.tobytes()
    assert st.numa_node == (info["node"] if mode == 1 else -1) and st.ring_node == info["ring_node"]
    assert os.sched_getaffinity(0) == mine                            # the caller's own thread was never touched
    assert _feed_threads() == {}                                      # the producer ended with the stream
    # the synchronous ring users pin the calling thread only for the length of the call
    h, dptr, got_n, st2 = ctx.load_to_device(p, ring={"slots": 2, "slot_records": 65536, "feeder_threads": 4})
    assert got_n == n and os.sched_getaffinity(0) == mine
    assert ia.DeviceBuffer.wrap(ctx, dptr, 24 * n).download().tobytes() == recs.tobytes()
    ctx.free(dptr)
    res, st3 = m.process_device(ctx, ia.PROC_REDUCE, ring={"slots": 3, "slot_records": 65536, "feeder_threads": 4})
    assert res == oracle.reduce_records(recs) and st3.numa_node == st.numa_node
    m.close()
    ctx.close()


def test_a_fake_topology_pins_the_feeders_where_it_says(ia, oracle, tmp_path, monkeypatch):
    """IBU_SYSFS_ROOT: the context reads a test tree that puts the GPU on a node owning only the first two CPUs of this process."""
    ctx0 = ia.Context(0)
    bdf = ctx0.numa()["pci_bus_id"].lower()
    ctx0.close()
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) < 3:
        pytest.skip("needs three CPUs")
    root = tmp_path / "sys"
    d = root / "bus" / "pci" / "devices" / bdf
    d.mkdir(parents=True)
    (d / "numa_node").write_text("5\n")
    nd = root / "devices" / "system" / "node" / "node5"
    nd.mkdir(parents=True)
    (nd / "cpulist").write_text(f"{mine[0]},{mine[1]}\n")
    monkeypatch.setenv("IBU_SYSFS_ROOT", str(root))
    ctx = ia.Context(0)
    info = ctx.numa()
    assert (info["node"], info["usable_cpus"], info["cpulist"]) == (5, 2, f"{mine[0]},{mine[1]}")
    p = tmp_path / "f.ibu"
    recs = _file(oracle, p, 100_000)
    r = ia.Reader.from_path(p)
    s = r.device_stream(ctx, ring={"slots": 2, "slot_records": 49152, "feeder_threads": 2})
    assert list(_feed_threads().values()) == [{mine[0], mine[1]}]
    got = [b.download().copy() for b in iter(lambda: _next_released(s), None)]
    s.close()
    r.close()
    assert np.concatenate(got).tobytes() == recs.tobytes()
    assert ctx.numa()["ring_placed"] in (False, True)                # node 5 does not exist: the kernel may refuse the policy; never an error
    ctx.close()


class _Held:
    def __init__(self, b):
        self.b = b

    def download(self):
        h = self.b.download()
        self.b.release()
        return h


def _next_released(s):
    b = s.next_batch()
    return None if b is None else _Held(b)
