"""Diagnostic (not a pytest): the whole config-2 archive through the unmodified file reader on the drop-in with T reader threads,
five passes per setting of the environment, every pass's rate (bench.py's leg keeps the best of three).
    python tests/diag_roll_T.py [T=8] ["ENV=V ENV=V" ...]        (MZ_DIAG_MAPPED=1: the readers on mz_stream_mem over one mapping)"""
import os, subprocess, sys, tempfile, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from tests import synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variants = sys.argv[2:] or [""]
c = synth.corpus(); rnd = np.random.RandomState(1)
uniq, n, size = 2048, 100000, 65536
pays, crcs = [], []
for i in range(uniq):
    o = int(rnd.randint(0, len(c) - size)); d = c[o:o + size]
    z = zlib.compressobj(6, zlib.DEFLATED, -15, 8); pays.append(z.compress(d) + z.flush()); crcs.append(zlib.crc32(d))
order = rnd.randint(0, uniq, size=n)
tmp = tempfile.mkdtemp(); path = os.path.join(tmp, "cfg2.zip")
bench.write_stream_zip(path, [pays[k] for k in order], [crcs[k] for k in order], size)
prog = r"""
import ctypes as C, os, sys, importlib
sys.path.insert(0, %r)
mz = importlib.import_module("minizip-ng_amd"); mz.require_gpu()
L = mz.lib()
L.mzhip_autoprime_stats.argtypes = [C.POINTER(C.c_uint64)] * 4
L.mzhip_prime_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
D = C.CDLL(os.path.join(%r, "integration", "_build", "libmzhipdrop.so"))
D.mzdrop_extract_file.restype = C.c_double
D.mzdrop_extract_file.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
D.mzdrop_extract_all.restype = C.c_double
D.mzdrop_extract_all.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
mapped = os.environ.get("MZ_DIAG_MAPPED") == "1"
out = []
for rep in range(5):
    L.mzhip_prime_clear()
    w0 = [C.c_uint64() for _ in range(4)]; L.mzhip_autoprime_stats(*[C.byref(x) for x in w0])
    ne, nb, fe = C.c_int64(), C.c_int64(), C.c_int32()
    tp = C.c_double()
    sec = D.mzdrop_extract_all(%r.encode(), %d, 0, C.byref(ne), C.byref(nb), C.byref(tp), C.byref(fe)) if mapped else D.mzdrop_extract_file(%r.encode(), %d, C.byref(ne), C.byref(nb), C.byref(fe))
    w1 = [C.c_uint64() for _ in range(4)]; L.mzhip_autoprime_stats(*[C.byref(x) for x in w1])
    e, h, m = C.c_uint64(), C.c_uint64(), C.c_uint64()
    L.mzhip_prime_stats(C.byref(e), C.byref(h), C.byref(m))
    out.append("%%.2f GiB/s (%%d primed, %%d look-ups missed%%s)" %% (nb.value / 2**30 / sec, w1[0].value - w0[0].value, m.value, "" if fe.value == 0 and ne.value == %d else " FAILED"))
print("; ".join(out))
""" % (ROOT, ROOT, path, T, path, T, n)
for v in variants:
    env = dict(os.environ)
    for kv in v.split():
        k, _, val = kv.partition("=")
        env[k] = val
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, env=env, timeout=900)
    print("T=%d %-44s %s" % (T, v or "(default)", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
os.remove(path); os.rmdir(tmp)
