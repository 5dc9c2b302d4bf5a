"""GPU test of the write-side prime (SURVEY 8b "Batching", BASELINE config 5): one batch launch compresses the buffers,
after which the reference's UNMODIFIED writer loop (mz_zip_writer_add_buffer -> mz_zip_entry_write ->
mz_stream_zlib_write / mz_stream_lzma_write -> mz_crypt_crc32_update, one entry at a time) is answered from the cache.
Parity = the archive is read back by the reference codecs (mz_zip reader + zlib / liblzma) to exactly the input, and an
entry that only partly equals a primed buffer takes the ordinary path with nothing lost."""
import ctypes as C
import importlib
import os
import tempfile
import time
import zlib

import numpy as np
import pytest

import oracle
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROP = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")


@pytest.fixture(scope="module")
def env():
    mz = importlib.import_module("minizip-ng_amd")
    mz.require_gpu()
    if not (os.path.exists(DROP) and oracle.have_ref()):
        pytest.skip("drop-in / reference libraries missing (built where /root/reference exists)")
    L = mz.lib()
    L.mzhip_prime_write.restype = C.c_int64
    L.mzhip_prime_write.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.mzhip_prime_write_stats.argtypes = [C.POINTER(C.c_uint64)] * 3
    return L, oracle.MzDriver(DROP), oracle.ref()


def _prime(L, method, blob, offs, lens):
    o = np.ascontiguousarray(offs, dtype=np.uint64)
    l = np.ascontiguousarray(lens, dtype=np.uint32)
    return L.mzhip_prime_write(method, blob.ctypes.data, o.ctypes.data, l.ctypes.data, len(l))


def _stats(L):
    e, h, m = C.c_uint64(), C.c_uint64(), C.c_uint64()
    L.mzhip_prime_write_stats(C.byref(e), C.byref(h), C.byref(m))
    return e.value, h.value, m.value


def _read_back(ref, path, blob, offs, lens):
    table = ref.zip_index(path)
    assert len(table) == len(lens)
    out_off = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
    out = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
    _, crc, ulen, st = ref.zip_read_all(path, table[:, 6].copy(), nthreads=4, own_crc=False, out=out, out_off=out_off)
    assert (st == 0).all() and (ulen == lens).all()
    for i in range(len(lens)):
        want = blob[int(offs[i]):int(offs[i]) + int(lens[i])]
        assert (out[int(out_off[i]):int(out_off[i]) + int(lens[i])] == want).all(), i
        if i % 50 == 0:
            assert int(crc[i]) == zlib.crc32(want.tobytes()), i
    return table


@pytest.mark.parametrize("method", [8, 14], ids=["deflate", "lzma"])
def test_write_prime_serves_unmodified_writer_loop(env, method):
    L, hip, ref = env
    c = np.frombuffer(synth.corpus(), dtype=np.uint8).copy()
    rnd = np.random.RandomState(17 + method)
    n = 1500 if method == 8 else 300
    lens = np.full(n, 65536, dtype=np.int32)
    lens[::7] = rnd.randint(16, 200000, size=len(lens[::7]))       # ragged: below, at and beyond one 65 535-byte chunk
    lens[:8] = (0, 1, 15, 16, 17, 65535, 65537, 131070)
    offs = rnd.randint(0, len(c) - 200000, size=n).astype(np.int64)
    cacheable = int(((lens >= 16) & (lens <= (8 << 20))).sum())
    with tempfile.TemporaryDirectory() as tmp:
        L.mzhip_prime_write_clear()
        plain = os.path.join(tmp, "plain.zip")
        t0 = time.perf_counter()
        hip.zip_write(plain, c, offs, lens, method=method, level=1)
        t_plain = time.perf_counter() - t0
        _read_back(ref, plain, c, offs, lens)

        t0 = time.perf_counter()
        assert _prime(L, method, c, offs, lens) == cacheable
        t_prime = time.perf_counter() - t0
        primed = os.path.join(tmp, "primed.zip")
        t0 = time.perf_counter()
        hip.zip_write(primed, c, offs, lens, method=method, level=1)
        t_primed = time.perf_counter() - t0
        ent, hits, miss = _stats(L)
        assert ent == cacheable and hits == cacheable and miss == 0
        tp = _read_back(ref, primed, c, offs, lens)
        t0 = _read_back(ref, plain, c, offs, lens)
        assert (tp[:, 2] == t0[:, 2]).all() and (tp[:, 3] == t0[:, 3]).all()   # same CRCs, same stream sizes
        assert open(primed, "rb").read() == open(plain, "rb").read()          # the batch codes what the per-entry path codes
        print("method %d: %d entries  per-entry launches %.3f s   prime %.3f s + writer loop %.3f s" %
              (method, n, t_plain, t_prime, t_primed))
        assert t_primed < t_plain
        L.mzhip_prime_write_clear()
        assert _stats(L)[0] == 0


def test_write_prime_divergence_falls_back(env):
    """Entries that start like a primed buffer but are not it: a proper prefix, an extension, one byte changed in the
    second chunk, one byte changed in the first chunk, plus the buffer itself."""
    L, hip, ref = env
    c = np.frombuffer(synth.corpus(), dtype=np.uint8).copy()
    a = c[1000:201000].copy()                                      # the primed buffer, 200 000 bytes
    for method in (8, 14):
        L.mzhip_prime_write_clear()
        assert _prime(L, method, a, [0], [len(a)]) == 1
        late = a.copy()
        late[150000] ^= 0x21
        early = a.copy()
        early[5] ^= 0x21
        parts = [a, a[:100000], np.concatenate((a, c[:777])), late, early, a[:65535], a]
        blob = np.concatenate(parts)
        lens = np.array([len(p) for p in parts], dtype=np.int32)
        offs = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "d.zip")
            hip.zip_write(path, blob, offs, lens, method=method, level=1)
            _read_back(ref, path, blob, offs, lens)
        ent, hits, miss = _stats(L)
        assert ent == 1 and hits == 2                              # only the two exact copies were served from the cache
    L.mzhip_prime_write_clear()
