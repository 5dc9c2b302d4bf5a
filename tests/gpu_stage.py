"""Staged GPU bring-up probe (ctypes only, no torch): python -u tests/gpu_stage.py <stage>.
Each stage runs in its own process under `timeout`, so a device-side hang is localised."""
import ctypes as C
import importlib
import sys
import time
import zlib

sys.path.insert(0, ".")
from tests import synth  # noqa: E402

mz = importlib.import_module("minizip-ng_amd")
stage = sys.argv[1]
L = mz.lib()
t0 = time.time()
print("stage", stage, "devices", L.mzhip_device_count(), flush=True)
cases = {n: (d, z) for n, d, z in synth.edge_payloads()}


def inflate(name, extra=b""):
    d, z = cases[name]
    out = C.create_string_buffer(len(d) + 16)
    src = C.create_string_buffer(z + extra, len(z) + len(extra) + 1)
    ol, iu, crc = C.c_uint32(), C.c_uint32(), C.c_uint32()
    a = mz.InflateHostArgs(size=C.sizeof(mz.InflateHostArgs), in_len=len(z) + len(extra), buf_cap=len(d) + 8, in_=C.addressof(src),
                           buf=C.addressof(out), out_len=C.addressof(ol), in_used=C.addressof(iu), crc=C.addressof(crc))
    st = L.mzhip_inflate_host_a(C.byref(a))
    ok = st == 0 and out.raw[:ol.value] == d and crc.value == zlib.crc32(d) and iu.value == len(z)
    print(name, "status", st, "out", ol.value, "/", len(d), "used", iu.value, "/", len(z), "crc_ok",
          crc.value == zlib.crc32(d), "OK" if ok else "FAIL", "err=", L.mzhip_last_error(), flush=True)


if stage == "crc":
    for n in (5, 1000, 5000, 70000):
        b = bytes(range(256)) * (n // 256 + 1)
        b = b[:n]
        print("crc", n, hex(L.mzhip_crc32_host(0, b, n)), hex(zlib.crc32(b)), flush=True)
elif stage == "lzma":
    from tests.test_oracle import _zip_lzma

    d = synth.corpus()[:20000]
    z = _zip_lzma(d)
    L.mzhip_lzma_host.restype = C.c_int32
    L.mzhip_lzma_host.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int64] + [C.POINTER(C.c_uint32)] * 3
    out = C.create_string_buffer(len(d) + 16)
    ol, iu, crc = C.c_uint32(), C.c_uint32(), C.c_uint32()
    st = L.mzhip_lzma_host(z, len(z), out, len(d) + 8, len(d), C.byref(ol), C.byref(iu), C.byref(crc))
    print("lzma status", st, ol.value, iu.value, len(z), out.raw[:ol.value] == d, crc.value == zlib.crc32(d), flush=True)
else:
    inflate(stage)
print("done %.2fs" % (time.time() - t0), flush=True)
