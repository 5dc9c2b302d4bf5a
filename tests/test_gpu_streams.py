"""Concurrent use of the batch ABI: several host threads, each on its own HIP stream, running the two encoders that
need per-launch device scratch (K4 tokens, K6 tokens) back to back without synchronising in between.  The scratch
cache (scratch_acquire in mzhip_kernels.hip) must never hand two in-flight launches on different streams the same
buffer; every result is checked by inflating / LZMA-decoding it on the CPU."""
import ctypes as C
import lzma
import threading
import zlib

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util

    gpu_util.mz.require_gpu()
    L = gpu_util.mz.lib()
    L.mzhip_deflate_batch.restype = C.c_int32
    L.mzhip_deflate_batch.argtypes = [C.c_void_p] * 7 + [C.c_uint32] + [C.c_void_p] * 4
    L.mzhip_lzma_encode_batch.restype = C.c_int32
    L.mzhip_lzma_encode_batch.argtypes = [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 4
    return gpu_util


def test_encoders_on_concurrent_streams(gpu):
    import torch

    L = gpu.mz.lib()
    n_threads, rounds = 4, 5
    errors, work = [], {}

    def worker(t):
        try:
            stream = torch.cuda.Stream()
            jobs = []
            with torch.cuda.stream(stream):
                for r in range(rounds):
                    # sizes differ per thread and round so the cache sees growing and shrinking requests
                    n = 40 + 37 * ((t + r) % 4)
                    size = (8192, 65536, 200000, 30000)[(t * 3 + r) % 4]
                    datas = synth.slices(n, size, 9000 + 100 * t + r)
                    caps = [len(d) + len(d) // 8 + 1024 for d in datas]
                    for kind in ("deflate", "lzma"):
                        b = gpu.make_batch(datas, caps)
                        dev = b["d_in"].device
                        out_len, crc, status = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
                        if kind == "deflate":
                            rc = L.mzhip_deflate_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                                       b["d_out"].data_ptr(), b["out_off"].data_ptr(), b["out_cap"].data_ptr(),
                                                       None, n, out_len.data_ptr(), crc.data_ptr(), status.data_ptr(),
                                                       stream.cuda_stream)
                        else:
                            rc = L.mzhip_lzma_encode_batch(b["d_in"].data_ptr(), b["in_off"].data_ptr(), b["in_len"].data_ptr(),
                                                           size, b["d_out"].data_ptr(), b["out_off"].data_ptr(),
                                                           b["out_cap"].data_ptr(), None, n, out_len.data_ptr(), crc.data_ptr(),
                                                           status.data_ptr(), stream.cuda_stream)
                        assert rc == 0, (t, r, kind, rc)
                        jobs.append((kind, datas, b, out_len, crc, status))      # no synchronisation between launches
            stream.synchronize()
            work[t] = jobs
        except Exception as e:                                                   # noqa: BLE001 - reported below
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    torch.cuda.synchronize()
    checked = 0
    for t, jobs in work.items():
        for kind, datas, b, out_len, crc, status in jobs:
            h = b["d_out"].cpu().numpy()
            ol, st, k = out_len.cpu().numpy(), status.cpu().numpy(), gpu.mz.u32(crc)
            assert (st == 0).all(), (t, kind)
            for i in range(0, len(datas), 7):
                z = gpu.entry_bytes(b, h, i, int(ol[i]))
                assert k[i] == zlib.crc32(datas[i]), (t, kind, i)
                if kind == "deflate":
                    assert zlib.decompress(z, -15) == datas[i], (t, kind, i)
                else:
                    assert lzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=lzma.FORMAT_ALONE) == datas[i], (t, kind, i)
                checked += 1
    assert checked > 300
