"""One rank of tests/test_gpu_archive.py::test_c_level_result_gather_two_ranks_over_rccl (not a pytest): N ranks, one device each,
a real ncclCommInitRank communicator, mzhip_gather_results -- the per-archive {crc, status} gather behind the C ABI
(north_star: "RCCL over xGMI only for the final per-archive CRC gather") with N > 1, which a one-GPU box cannot run."""
import ctypes as C
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
dist.init_process_group("gloo")        # only to hand the RCCL unique id around; the gather itself is the library's
mz = importlib.import_module("minizip-ng_amd")
mz.require_gpu()
L = mz.lib()
try:
    R = C.CDLL("librccl.so.1")
except OSError:
    R = C.CDLL("/opt/rocm/lib/librccl.so")


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


uid = UniqueId()
if rank == 0:
    assert R.ncclGetUniqueId(C.byref(uid)) == 0
t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
dist.broadcast(t, 0)
C.memmove(C.byref(uid), bytes(t.numpy().tobytes()), 128)
comm = C.c_void_p()
R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert R.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
dev = torch.device("cuda", torch.cuda.current_device())
n = 100003                                            # ragged slices: the padded-block path
bounds = [n * r // world + (7 if 0 < r < world else 0) for r in range(world + 1)]
b = (C.c_int64 * (world + 1))(*bounds)
lo, hi = bounds[rank], bounds[rank + 1]
full_crc = (torch.arange(n, dtype=torch.int64) * 2654435761 % (1 << 31)).to(torch.int32)
full_st = -(torch.arange(n, dtype=torch.int32) % 5)
crc, st = full_crc[lo:hi].to(dev), full_st[lo:hi].to(dev)
all_crc = torch.zeros(n, dtype=torch.int32, device=dev)
all_st = torch.ones(n, dtype=torch.int32, device=dev)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.mzhip_last_error.restype = C.c_char_p
for rep in range(3):
    rc = L.mzhip_gather_results(comm, rank, world, b, crc.data_ptr(), st.data_ptr(), all_crc.data_ptr(), all_st.data_ptr(), s)
    torch.cuda.synchronize()
    assert rc == 0, (rc, L.mzhip_last_error())
    assert torch.equal(all_crc.cpu(), full_crc) and torch.equal(all_st.cpu(), full_st), rank
# a table that runs backwards anywhere -- here in a slice that is not this rank's for most ranks -- is refused before anything
# is copied
if world >= 2:
    wrong = list(bounds)
    wrong[1] = wrong[2] + 5
    bad = (C.c_int64 * (world + 1))(*wrong)
    assert L.mzhip_gather_results(comm, rank, world, bad, crc.data_ptr(), st.data_ptr(), all_crc.data_ptr(), all_st.data_ptr(), s) == -102
R.ncclCommDestroy.argtypes = [C.c_void_p]
R.ncclCommDestroy(comm)
dist.barrier()
if rank == 0:
    print("gather over RCCL with %d ranks ok" % world, flush=True)
dist.destroy_process_group()
