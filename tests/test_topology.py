"""Placement on multi-socket hosts (csrc/icsp_topology.cpp): device -> NUMA node -> CPUs from sysfs, thread
binding, first-touch population, chunk -> device dealing.  CPU tests run against a fake sysfs tree (ICSP_SYSFS_ROOT); the GPU test
runs icsp_enc through the placement path on a one-socket box by giving it a fake two-node tree that puts the real device on
node 1, and checks that the files are the reference's.  The reference's thread pool (ICSP_thread.cpp:39-77) places nothing."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from icspcodec_amd import capi, clipgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_tree(root, pci_nodes, node_cpus):
    for bus, node in pci_nodes.items():
        d = os.path.join(root, "sys", "bus", "pci", "devices", bus)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "numa_node"), "w").write(f"{node}\n")
    for node, cpus in node_cpus.items():
        d = os.path.join(root, "sys", "devices", "system", "node", f"node{node}")
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "cpulist"), "w").write(cpus + "\n")


def _cpus(lib, text, cap=64):
    buf = (C.c_int * cap)()
    n = lib.icsp_parse_cpulist(text.encode() if text is not None else None, buf, cap)
    return n, list(buf[: max(0, min(n, cap))])


def test_cpulist_parsing():
    lib = capi.load()
    assert _cpus(lib, "0-3,8,10-11\n") == (7, [0, 1, 2, 3, 8, 10, 11])
    assert _cpus(lib, "5") == (1, [5])
    assert _cpus(lib, "0-63,128-191")[0] == 128
    assert _cpus(lib, "\n") == (0, [])                      # a memory-only node
    assert _cpus(lib, "0-127", cap=4) == (128, [0, 1, 2, 3])     # counts past the capacity, stores what fits
    for bad in ("a-b", "3-1", "1,,2", "1-", "-1", "1;2"):
        assert _cpus(lib, bad)[0] == -1, bad
    assert lib.icsp_parse_cpulist(None, None, 0) == -1


def test_sysfs_lookup_on_a_fake_two_socket_tree(tmp_path, monkeypatch):
    lib = capi.load()
    _fake_tree(str(tmp_path), {"0000:c1:00.0": 1, "0000:05:00.0": 0, "0000:aa:00.0": -1}, {0: "0-3,8-11", 1: "4-7,12-15", 2: ""})
    monkeypatch.setenv("ICSP_SYSFS_ROOT", str(tmp_path))
    assert lib.icsp_numa_node_of_pci(b"0000:c1:00.0") == 1
    assert lib.icsp_numa_node_of_pci(b"0000:C1:00.0") == 1          # hipDeviceGetPCIBusId may print upper case
    assert lib.icsp_numa_node_of_pci(b"0000:05:00.0") == 0
    assert lib.icsp_numa_node_of_pci(b"0000:aa:00.0") == -1         # the kernel's "no affinity"
    assert lib.icsp_numa_node_of_pci(b"0000:ff:00.0") == -1         # no such device
    assert lib.icsp_numa_node_of_pci(b"") == -1 and lib.icsp_numa_node_of_pci(None) == -1
    assert lib.icsp_numa_nodes() == 2                                # node2 has memory only
    buf = (C.c_int * 16)()
    assert lib.icsp_numa_cpus(1, buf, 16) == 8 and list(buf[:8]) == [4, 5, 6, 7, 12, 13, 14, 15]
    assert lib.icsp_numa_cpus(2, buf, 16) == 0 and lib.icsp_numa_cpus(7, buf, 16) == -1 and lib.icsp_numa_cpus(-1, buf, 16) == -1


def test_binding_is_a_no_op_where_there_is_nothing_to_place(tmp_path, monkeypatch):
    lib = capi.load()
    bound = C.c_int(7)
    before = os.sched_getaffinity(0)
    # unknown node
    assert lib.icsp_bind_thread_to_node(-1, C.byref(bound)) == 0 and bound.value == 0
    # a single node (this container's real tree, or an empty fake one)
    monkeypatch.setenv("ICSP_SYSFS_ROOT", str(tmp_path))
    assert lib.icsp_numa_nodes() == 1
    assert lib.icsp_bind_thread_to_node(0, C.byref(bound)) == 0 and bound.value == 0
    assert os.sched_getaffinity(0) == before
    # two nodes: the calling thread moves to the node's CPUs (those that exist here), and back
    ncpu = sorted(before)
    if len(ncpu) >= 2:
        half = len(ncpu) // 2
        _fake_tree(str(tmp_path), {}, {0: ",".join(map(str, ncpu[:half])), 1: ",".join(map(str, ncpu[half:]))})
        assert lib.icsp_bind_thread_to_node(1, C.byref(bound)) == 0 and bound.value == 1
        assert os.sched_getaffinity(0) == set(ncpu[half:])
        os.sched_setaffinity(0, before)
    # a node whose CPUs this machine does not have: the call is refused by the kernel and reports "not bound", no error
    _fake_tree(str(tmp_path), {}, {0: "0", 1: "1000-1003"})
    assert lib.icsp_bind_thread_to_node(1, C.byref(bound)) == 0 and bound.value == 0
    assert os.sched_getaffinity(0) == before


def test_populate_and_chunk_dealing():
    lib = capi.load()
    lib.icsp_populate_here.argtypes = [C.c_void_p, C.c_size_t]
    import mmap
    m = mmap.mmap(-1, 1 << 20)
    addr = C.addressof(C.c_char.from_buffer(m))
    assert lib.icsp_populate_here(addr + 100, (1 << 20) - 200) == 0
    assert m[:16] == b"\0" * 16                                       # touching does not change the bytes
    assert lib.icsp_populate_here(None, 10) != 0
    assert [lib.icsp_chunk_device(c, 1) for c in range(5)] == [0] * 5
    assert [lib.icsp_chunk_device(c, 4) for c in range(9)] == [0, 1, 2, 3, 0, 1, 2, 3, 0]
    assert lib.icsp_device_numa_node(0) in (-1, 0, 1, 2, 3, 4, 5, 6, 7)      # no device here: -1


@pytest.mark.gpu
@pytest.mark.parametrize("period,extra", [(0, []), (6, ["--chunk", "6", "--streams", "2"]), (6, ["--gpus", "2", "--chunk", "6"])])
def test_icsp_enc_through_the_placement_path(tmp_path, golden_dir, period, extra):
    lib = capi.load()
    bus = C.create_string_buffer(32)
    assert lib.icsp_device_pci_bus_id(0, bus, 32) == 0 and len(bus.value) >= 12
    ncpu = sorted(os.sched_getaffinity(0))
    half = max(1, len(ncpu) // 2)
    fake = tmp_path / "fake"
    _fake_tree(str(fake), {bus.value.decode().lower(): 1}, {0: ",".join(map(str, ncpu[:half])), 1: ",".join(map(str, ncpu[half:] or ncpu))})
    n, qp = 12, 16
    clip = clipgen.synth_clip("foremanlike", n)
    fn = clipgen.file_name("foremanlike", n)
    clip.tofile(tmp_path / fn)
    env = dict(os.environ, ICSP_SYSFS_ROOT=str(fake))
    r = subprocess.run([os.path.join(ROOT, "icspcodec_amd", "icsp_enc"), "-i", fn, "-n", str(n), "-q", str(qp), "--intraPeriod", str(period), "--stats"] + extra,
                       cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0, r.stdout
    st = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("[icsp_enc]")][0][10:])
    assert st["numa"]["nodes"] == 2 and st["numa"]["placement"] is True and st["numa"]["device0_node"] == 1
    assert st["numa"]["threads_bound"] >= 1 and st["numa"]["output_bytes_placed"] == n * 352 * 288 * 3 // 2
    ref = [s for s in json.load(open(os.path.join(golden_dir, "streams.json")))
           if (s["clip"], s["nframes"], s["qp"], s["intra_period"]) == ("foremanlike", n, qp, period) and "bin_sha256" in s][0]
    assert hashlib.sha256((tmp_path / f"foremanlike_compCIF_{qp}_{qp}_{period}.bin").read_bytes()).hexdigest() == ref["bin_sha256"]
    assert hashlib.sha256((tmp_path / "test_yuv.yuv").read_bytes()).hexdigest() == ref["recon_sha256"]
