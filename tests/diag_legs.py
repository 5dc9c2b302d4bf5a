import ctypes as C, os, sys, time, zlib, tempfile
ROOT='/root/repo' if os.path.isdir('/root/repo/minizip-ng_amd') else os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np, importlib
import bench
from tests import synth
mz = importlib.import_module("minizip-ng_amd"); mz.require_gpu()
c = synth.corpus(); rnd = np.random.RandomState(1)
uniq, n, size = 2048, 100000, 65536
pays, crcs = [], []
for i in range(uniq):
    o = int(rnd.randint(0, len(c) - size)); d = c[o:o+size]
    z = zlib.compressobj(6, zlib.DEFLATED, -15, 8); pays.append(z.compress(d) + z.flush()); crcs.append(zlib.crc32(d))
order = rnd.randint(0, uniq, size=n)
tmp = tempfile.mkdtemp(); path = os.path.join(tmp, "cfg2.zip")
bench.write_stream_zip(path, [pays[k] for k in order], [crcs[k] for k in order], size)
want = np.array([crcs[k] for k in order], dtype=np.uint32)
out, cb = bench.full_archive_legs(mz, path, n, size, want, bench.usable_cores(), True)
for k, v in out.items():
    if not k.endswith("_sample"): print(k, v)
    else: print("    ", v[-170:])
print(cb["value"] if cb else None, (cb or {}).get("sample", "")[-400:])
D = C.CDLL(os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so"))
D.mzdrop_extract_file.restype = C.c_double
D.mzdrop_extract_file.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
os.environ["MZDROP_TRACE"] = "1"
for T in (1, 8):
    mz.lib().mzhip_prime_clear()
    ne, nb, fe = C.c_int64(), C.c_int64(), C.c_int32()
    sec = D.mzdrop_extract_file(path.encode(), T, C.byref(ne), C.byref(nb), C.byref(fe))
    print("T=%d: %.3f s = %.2f GiB/s err %d" % (T, sec, nb.value / 2**30 / sec, fe.value), flush=True)
