"""GPU tests of the DROP-IN boundary: the reference's own mz_zip.c / mz_zip_rw.c / mz_strm*.c, compiled
unmodified and linked against libmzhip.so instead of mz_strm_zlib.o / mz_strm_lzma.o / the CRC symbol
(integration/_build/libmzhipdrop.so), are driven through the same oracle/mz_driver.c entry points as the
all-reference build (oracle/_ref/libmzref.so).  Every read() return value, TOTAL_IN/TOTAL_OUT, close() and
error() code and every output byte must agree."""
import os
import tempfile
import zlib

import numpy as np
import pytest

import oracle
from tests import synth
from tests.test_oracle import _zip_lzma

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROP = os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")


@pytest.fixture(scope="module")
def libs():
    import importlib

    importlib.import_module("minizip-ng_amd").require_gpu()
    if not os.path.exists(DROP):
        pytest.skip("integration/_build/libmzhipdrop.so missing (built by __graft_entry__.build() where /root/reference exists)")
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libmzref.so missing (built where /root/reference exists)")
    return oracle.MzDriver(DROP), oracle.ref()


def test_crc32_symbol(libs):
    hip, ref = libs
    rnd = np.random.RandomState(2)
    for n in (1, 9, 1000, 65535, 65536, 700001):
        d = rnd.bytes(n)
        assert hip.crc32(d) == ref.crc32(d) == zlib.crc32(d)
        assert hip.crc32(d, 0x1234ABCD) == ref.crc32(d, 0x1234ABCD)
    # chaining exactly as mz_zip_entry_read does it (65 535 + 1 byte, mz_zip_rw.c:55)
    d = rnd.bytes(65536)
    assert hip.crc32(d[65535:], hip.crc32(d[:65535])) == zlib.crc32(d)


def test_zlib_stream_parity(libs):
    hip, ref = libs
    keys = ("rets", "out", "total_in", "total_out", "close", "error", "open")
    for name, data, z in synth.edge_payloads():
        for chunk in (65535, 16384):
            for extra in (b"", b"\x00" * 40000):
                a = hip.stream_decode(8, z + extra, len(data) + 64, chunk=chunk)
                b = ref.stream_decode(8, z + extra, len(data) + 64, chunk=chunk)
                assert {k: a[k] for k in keys} == {k: b[k] for k in keys}, (name, chunk, len(extra))


def test_zlib_stream_error_parity(libs):
    hip, ref = libs
    data = synth.corpus()[:65536]
    z = synth.deflate_raw(data)
    for cname, bad in synth.corruptions(z):
        a = hip.stream_decode(8, bad, len(data) + 70000)
        b = ref.stream_decode(8, bad, len(data) + 70000)
        assert a["rets"][-1] == b["rets"][-1], (cname, a["rets"], b["rets"])
        assert (a["close"], a["error"]) == (b["close"], b["error"]), cname
    # TOTAL_IN_MAX truncation
    a = hip.stream_decode(8, z, len(data) + 64, max_in=len(z) // 2)
    b = ref.stream_decode(8, z, len(data) + 64, max_in=len(z) // 2)
    assert (a["rets"], a["close"], a["error"]) == (b["rets"], b["close"], b["error"]) == ([-5], -112, -5)


def test_lzma_stream_parity(libs):
    hip, ref = libs
    keys = ("rets", "out", "total_in", "total_out", "close", "error", "open")
    c = synth.corpus()
    for d in (c[:150000], c[:10], b"A" * 70000):
        z = _zip_lzma(d)
        a = hip.stream_decode(14, z, len(d) + 64, max_in=len(z), max_out=len(d))
        b = ref.stream_decode(14, z, len(d) + 64, max_in=len(z), max_out=len(d))
        assert {k: a[k] for k in keys} == {k: b[k] for k in keys}, len(d)
        bad = z[:len(z) // 3] + bytes([z[len(z) // 3] ^ 0x55]) + z[len(z) // 3 + 1:]
        a = hip.stream_decode(14, bad, len(d) + 70000)
        b = ref.stream_decode(14, bad, len(d) + 70000)
        assert a["rets"][-1] == b["rets"][-1] == -3 and a["close"] == b["close"] == -112


def test_archives_through_unmodified_mz_zip(libs):
    """Archives written by the reference writer are extracted by the reference's mz_zip reader running on the
    HIP codecs; the CRC verification inside mz_zip_entry_read_close (mz_zip.c:2116-2128) must pass."""
    hip, ref = libs
    c = np.frombuffer(synth.corpus(), dtype=np.uint8)
    rnd = np.random.RandomState(4)
    with tempfile.TemporaryDirectory() as tmp:
        for method, level, n, size in ((8, 6, 200, 65536), (8, 1, 50, 8192), (0, 0, 100, 30000), (14, 6, 12, 50000)):
            lens = rnd.randint(0, size + 1, size=n).astype(np.int32)
            lens[:3] = (0, 1, size)
            offs = rnd.randint(0, len(c) - size, size=n).astype(np.int64)
            path = os.path.join(tmp, "m%d_l%d.zip" % (method, level))
            ref.zip_write(path, c, offs, lens, method=method, level=level)
            t_ref = ref.zip_index(path)
            t_hip = hip.zip_index(path)
            assert (t_ref == t_hip).all()
            cd = t_ref[:, 6].copy()
            out_off = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
            o_ref = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
            o_hip = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
            _, crc_r, ulen_r, st_r = ref.zip_read_all(path, cd, nthreads=2, out=o_ref, out_off=out_off)
            _, crc_h, ulen_h, st_h = hip.zip_read_all(path, cd, nthreads=2, out=o_hip, out_off=out_off)
            assert (st_r == 0).all() and (st_h == 0).all(), (method, st_h[st_h != 0][:5])
            assert (crc_r == crc_h).all() and (ulen_r == ulen_h).all() and (ulen_h == lens).all()
            assert (crc_h == t_ref[:, 2].astype(np.uint32)).all()      # == the central directory's CRCs
            assert (o_ref == o_hip).all()
