"""The C++ multi-GPU layer (include/nvbio_hip/multi_device.h, nvbio_amd/csrc/comm.hip): RCCL bound at run time, a communicator of the
ranks this box has, the record gather -- the path's only collective.  A 1-GPU box exercises the binding, a world of one
and the single-device DeviceGroup; with two or more GPUs the same tests run real ncclSend / ncclRecv between host threads."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import nvbio_amd as nvb

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_binds_and_a_world_of_one_gathers():
    L = nvb.lib()
    assert L.nvbio_hip_comm_available() == 1
    assert L.nvbio_hip_device_count() >= 1
    ident = (C.c_uint8 * 128)()
    assert L.nvbio_hip_comm_unique_id(ident) == 0
    comm = C.c_void_p()
    assert L.nvbio_hip_comm_init_rank(C.byref(comm), 1, 0, ident) == 0
    rank, world = C.c_int(-1), C.c_int(-1)
    assert L.nvbio_hip_comm_rank(comm, C.byref(rank), C.byref(world)) == 0 and (rank.value, world.value) == (0, 1)
    rec = torch.arange(4000 * 4, dtype=torch.int32, device="cuda").reshape(4000, 4)
    out = torch.zeros_like(rec)
    counts = (C.c_uint64 * 1)(4000)
    assert L.nvbio_hip_gather_records(comm, C.c_void_p(rec.data_ptr()), counts, 16, C.c_void_p(out.data_ptr()), 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, rec)
    # argument checks
    assert L.nvbio_hip_gather_records(comm, C.c_void_p(rec.data_ptr()), counts, 16, None, 0, None) != 0          # the root needs a receive buffer
    assert L.nvbio_hip_gather_records(comm, C.c_void_p(rec.data_ptr()), counts, 16, C.c_void_p(out.data_ptr()), 3, None) != 0
    assert L.nvbio_hip_comm_destroy(comm) == 0


def test_cxx_gatherer_of_the_python_layer():
    from nvbio_amd.distributed import CxxComm, CxxRecordGather
    comm = CxxComm()
    g = CxxRecordGather(comm, 1000, 4, dst=0, device="cuda")
    rec = torch.randint(-2 ** 31, 2 ** 31 - 1, (1000, 4), dtype=torch.int32, device="cuda")
    table = g.gather(rec)
    torch.cuda.synchronize()
    assert torch.equal(table, rec) and torch.equal(g.shard(0), rec)
    comm.close()


@pytest.mark.parametrize("n_devices", [1, 2, 0])
def test_device_group_one_host_thread_per_device(n_devices):
    """DeviceGroup::local + run + gather_records from C++ (tests/cxx/aligner_shim.cpp: nvbio_multi_device_selftest): 0 = every device"""
    have = int(nvb.lib().nvbio_hip_device_count())
    if n_devices > have:
        pytest.skip("needs %d GPUs, this box has %d" % (n_devices, have))
    shim = C.CDLL(os.path.join(ROOT, "tests", "cxx", "libaligner_shim.so"))
    shim.nvbio_multi_device_selftest.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32]
    for n_total, words in ((100003, 4), (7, 8), (1 << 20, 1)):
        assert shim.nvbio_multi_device_selftest(n_devices, n_total, words) == 0
