"""Quick GPU probe (not a pytest): times mzhip_inflate_batch on N corpus slices."""
import sys
import time
import zlib

import numpy as np
import torch

import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import gpu_util, synth  # noqa: E402

n_unique = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_total = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
size = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
datas = synth.slices(n_unique, size, 1234)
pays = [synth.deflate_raw(d) for d in datas]
idx = np.arange(n_total) % n_unique
batch = gpu_util.make_batch([pays[i] for i in idx], [size] * n_total)
want = np.array([zlib.crc32(d) for d in datas], dtype=np.uint32)[idx]
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out_len, in_used, crc, status = gpu_util.mz.inflate_batch(batch["d_in"], batch["in_off"], batch["in_len"],
                                                              batch["d_out"], batch["out_off"], batch["out_cap"])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    ok = bool((status.cpu().numpy() == 0).all() and (gpu_util.mz.u32(crc) == want).all())
    cbytes = int(batch["in_len"].sum().item())
    print("rep %d: %.3f ms  %.2f GiB/s out  (%.2f GB/s in+out)  ok=%s ratio=%.3f" % (
        rep, ms, n_total * size / 2**30 / (ms / 1e3), (n_total * size + cbytes) / 1e9 / (ms / 1e3), ok,
        cbytes / (n_total * size)))

# measurement builds (make PROF=1) export the per-section cycle sums of K1
import ctypes  # noqa: E402

lib = ctypes.CDLL(os.environ.get("MZHIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "minizip-ng_amd", "_build", "libmzhip.so"))
if hasattr(lib, "mzhip_inflate_launch_geometry"):
    geo = [ctypes.c_uint32() for _ in range(3)]
    lib.mzhip_inflate_launch_geometry.restype = None
    lib.mzhip_inflate_launch_geometry(ctypes.c_uint32(0x7FFFFFFF), *(ctypes.byref(g) for g in geo))
    print("geometry: %d resident workgroups of %d waves, %d bytes of LDS each" % tuple(g.value for g in geo))
if hasattr(lib, "mzhip_prof_read"):
    buf = (ctypes.c_ulonglong * 32)()
    lib.mzhip_prof_read(buf, 1)
    names = {0: "header: up to the code-length code", 1: "header: code lengths", 2: "header: decode tables", 3: "step loop + window load",
             4: "counting passes | chase: pass 1", 5: "emitting passes | chase: pass 2", 6: "far copies", 7: "near copies", 8: "store",
             9: "crc", 10: "crossings | chase: the chain", 13: "choosing the chunk | chase: emit rounds", 11: "block tails (step loop)",
             12: "crc tail"}
    tot = float(sum(buf)) or 1.0
    for i in sorted(names, key=lambda k: -buf[k]):
        print("  %-36s %5.1f %%  %9.1f Mcycles" % (names[i], 100.0 * buf[i] / tot, buf[i] / 1e6))
