#!/usr/bin/env python3
"""Known byte counts under the FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in
your own access pattern before trusting an absolute"): three kernels whose traffic is known exactly, run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE   (and WRITE_SIZE)   -- python profiles/pmc_calibrate.py
  k_crc32_batch      reads N bytes with 16-byte loads per lane (this library's own streaming read), writes nothing
  fill               torch x.fill_(7) on N bytes: writes N, reads nothing
  add                torch y = x + 1 on N bytes (uint8): reads N, writes N
  k_cal_*            profiles/cal/cal_kernels.hip: coalesced reads / writes of N bytes at 1, 4 and 16 bytes per lane
N = 2 GiB, far past the 256 MiB Infinity Cache.  profiles/calibrate_harvest.py turns the two CSVs into the factors
hbm_traffic*.json quotes beside the raw counter values."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import gpu_util  # noqa: E402

N = 2 << 30
dev = torch.device("cuda", 0)
x = torch.empty(N, dtype=torch.uint8, device=dev)
for _ in range(2):
    x.fill_(7)
torch.cuda.synchronize()
for _ in range(2):
    y = x + 1
torch.cuda.synchronize()
del y
mz = gpu_util.mz
L = mz.lib()
import ctypes as C  # noqa: E402

n = 32768
off = torch.arange(n, dtype=torch.int64, device=dev) * (N // n)
ln = torch.full((n,), N // n, dtype=torch.int32, device=dev)
crc = torch.zeros(n, dtype=torch.int32, device=dev)
L.mzhip_crc32_batch.restype = C.c_int32
L.mzhip_crc32_batch.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 3
for _ in range(2):
    rc = L.mzhip_crc32_batch(x.data_ptr(), off.data_ptr(), ln.data_ptr(), n, None, crc.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
torch.cuda.synchronize()
cal = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cal", "libcal.so"))
cal.cal_run.restype = C.c_int
cal.cal_run.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
sink = torch.zeros(16, dtype=torch.int32, device=dev)
for which in range(6):
    for _ in range(2):
        rc = cal.cal_run(which, x.data_ptr(), N, sink.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, (which, rc)
    torch.cuda.synchronize()
print("calibration kernels ran: N = %d bytes each" % N)
