import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import oracle
from tests import synth
L = C.CDLL("integration/_build/libmzhipdrop.so")
L.mz_crypt_crc32_update.restype = C.c_uint32
L.mz_crypt_crc32_update.argtypes = [C.c_uint32, C.c_void_p, C.c_int32]
buf = np.frombuffer((synth.bench_corpus()[0] * 4)[:1 << 20], dtype=np.uint8).copy()
for chunk in [int(x) for x in sys.argv[1:]] or (65535, 32768, 16384, 4096, 1 << 20):
    n = (256 << 20) // chunk
    crc = 0
    t0 = time.time()
    for i in range(n):
        crc = L.mz_crypt_crc32_update(crc, buf.ctypes.data, chunk)
    dt = time.time() - t0
    print("crc32_update, %7d-byte calls: %.1f us per call, %.2f GiB/s" % (chunk, dt / n * 1e6, n * chunk / dt / (1 << 30)))
