"""Quick GPU probe (not a pytest): the drop-in with T reader threads over a primed archive, T = 1 .. all cores
(integration/extract_threads.c).  Usage: python tests/perf_threads.py [entries=16384] [size=65536]"""
import ctypes as C, os, sys, tempfile, time, zlib
import numpy as np
os.environ.setdefault("TZ", "UTC"); time.tzset()  # (mktime() of the reference's header parser: see INTEGRATION.md)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib.util
from tests import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); sys.modules["bench"] = bench; spec.loader.exec_module(bench)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
size = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
c, _ = synth.bench_corpus()
offs, pays, crcs = bench.make_unique_deflate(c, min(n, 4096), size, 77, 60, 1)
k = len(pays)
path = os.path.join(tempfile.mkdtemp(), "t.zip")
bench.write_stream_zip(path, [pays[i % k] for i in range(n)], [crcs[i % k] for i in range(n)], size)
import importlib
mz = importlib.import_module("minizip-ng_amd"); L = mz.lib()
D = C.CDLL(os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so"))
D.mzdrop_extract_all.restype = C.c_double
D.mzdrop_extract_all.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
if os.environ.get("MZ_NEAR"):  # MZ_NEAR=16: the library's own placement call, at most that many CPUs of the GPU's node
    L.mzhip_init(0)
    print("bound near device 0:", L.mzhip_bind_thread_near_device(0, int(os.environ["MZ_NEAR"])), "CPUs", sorted(os.sched_getaffinity(0))[:3], "...")
if os.environ.get("MZ_PIN"):  # e.g. MZ_PIN=0-15: keep the process (and the threads it starts from here on) on these CPUs
    a, b = os.environ["MZ_PIN"].split("-")
    os.sched_setaffinity(0, set(range(int(a), int(b) + 1)))
    print("pinned to CPUs", os.environ["MZ_PIN"])
for mode in ([int(m) for m in os.environ.get("MZ_MODES", "1,2").split(",")]):  # 1: prime, then readers; 2: readers under the prime (mzhip_prime_mem_begin)
    for T in (1, 4, 8, 12, 16):
        best = None
        for _ in range(3):
            L.mzhip_prime_clear()
            ne, nb, tp, fe = C.c_int64(0), C.c_int64(0), C.c_double(0), C.c_int32(0)
            w0 = time.perf_counter()
            sec = D.mzdrop_extract_all(path.encode(), T, mode, C.byref(ne), C.byref(nb), C.byref(tp), C.byref(fe))
            if os.environ.get("MZDROP_TRACE"): print("   call took %.2f ms by the caller's clock, returned %.2f ms" % ((time.perf_counter() - w0) * 1e3, sec * 1e3))
            assert sec > 0 and fe.value == 0 and ne.value == n, (sec, fe.value, ne.value)
            if best is None or sec < best[0]: best = (sec, tp.value)
        print("mode %d T=%3d: total %.3f s (in prime calls %.3f s, rest %.3f s)  %.2f GiB/s" % (mode, T, best[0], best[1], best[0] - best[1], n * size / 2**30 / best[0]))
