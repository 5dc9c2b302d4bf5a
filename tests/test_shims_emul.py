"""CPU tests of the C SHIMS (minizip-ng_amd/csrc/shim_*.c -- the code that mirrors mz_strm_zlib.c / mz_strm_lzma.c /
mz_crypt_crc32_update call for call) behind the reference's unmodified zip layer, in a container without a GPU.

tests/emul/_build/libmockdrop.so links the shims against tests/emul/mock_device.cpp, which implements the
host-buffer entry points of include/mzhip.h on the 64-lane HOST EMULATION of the device cores (the same headers the
HIP build compiles).  The test bodies are the GPU tests' own (tests/test_gpu_dropin.py, tests/test_gpu_wrappers.py),
run here against the mock instead of the device; a GPU box runs them against the real thing.  Test infrastructure
only: nothing here is part of, or loaded by, the product."""
import os
import subprocess

import pytest

import oracle
from tests import test_gpu_dropin as D
from tests import test_gpu_wrappers as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "emul", "_build", "libmockdrop.so")


@pytest.fixture(scope="module")
def libs():
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul")], check=True, capture_output=True)
    if not os.path.exists(MOCK) or not oracle.have_ref():
        pytest.skip("tests/emul/_build/libmockdrop.so / oracle/_ref/libmzref.so need the reference sources at build time")
    return oracle.MzDriver(MOCK), oracle.ref()


def test_crc32_symbol(libs):
    D.test_crc32_symbol(libs)


def test_zlib_stream_parity(libs):
    D.test_zlib_stream_parity(libs)


def test_zlib_stream_error_parity(libs):
    D.test_zlib_stream_error_parity(libs)


def test_lzma_stream_parity(libs):
    D.test_lzma_stream_parity(libs)


def test_truncation_accounting_deflate(libs):
    D.test_truncation_accounting_deflate(libs)


def test_bitflip_verdicts_deflate(libs):
    D.test_bitflip_verdicts_deflate(libs)


def test_code_length_code_without_a_code(libs):
    D.test_code_length_code_without_a_code(libs)


def test_crafted_deflate_corners(libs):
    D.test_crafted_deflate_corners(libs)


def test_refusal_behind_a_full_buffer(libs):
    D.test_refusal_behind_a_full_buffer(libs)


def test_truncation_accounting_window_mode(libs):
    D.test_truncation_accounting_window_mode(libs)


def test_truncation_accounting_lzma(libs):
    D.test_truncation_accounting_lzma(libs)


def test_lzma_window_mode(libs):
    D.test_lzma_window_mode(libs)


def test_lzma_refusal_behind_a_full_buffer(libs):
    D.test_lzma_refusal_behind_a_full_buffer(libs)


def test_xz_window_mode(libs):
    D.test_xz_window_mode(libs)


def test_xz_window_mode_differential_fuzz(libs):
    D.test_xz_window_mode_differential_fuzz(libs)


def test_lzma_write_in_segments(libs):
    D.test_lzma_write_in_segments(libs)


def test_archives_through_unmodified_mz_zip(libs):
    D.test_archives_through_unmodified_mz_zip(libs)


def test_wrapped_read_parity(libs):
    W.test_wrapped_read_parity(libs)


def test_wrappers_in_window_mode(libs):
    W.test_wrappers_in_window_mode(libs)


def test_gzip_optional_header_fields(libs):
    W.test_gzip_optional_header_fields(libs)


def test_wrapper_error_parity(libs):
    W.test_wrapper_error_parity(libs)


def test_unsupported_windows_are_refused(libs):
    W.test_unsupported_windows_are_refused(libs)


def test_every_window_zlib_accepts(libs):
    W.test_every_window_zlib_accepts(libs)


def test_wrapped_write_roundtrip(libs):
    W.test_wrapped_write_roundtrip(libs)


# ---- the reference's CLI matrix (tests/test_gpu_cli.py) with minizip.c / minigzip.c linked against the mock

from tests import test_gpu_cli as K  # noqa: E402

MOCK_ZIP = os.path.join(ROOT, "tests", "emul", "_build", "minizip_mock")
MOCK_GZ = os.path.join(ROOT, "tests", "emul", "_build", "minigzip_mock")
cli_src = K.src           # the module-scoped fixture that lays out the stand-in test files


@pytest.mark.parametrize("fname,fargs", K.FLAVOURS, ids=[f[0] for f in K.FLAVOURS])
@pytest.mark.parametrize("mname,marg", K.METHODS[:3], ids=[m[0] for m in K.METHODS[:3]])     # raw, deflate, lzma
def test_cli_matrix_on_mock(libs, cli_src, tmp_path, mname, marg, fname, fargs):
    if not (os.path.exists(MOCK_ZIP) and os.path.exists(K.REF_ZIP)):
        pytest.skip("CLI binaries need the reference sources at build time")
    if fname == "zipcd" and mname == "raw":
        pytest.skip("CMakeLists.txt:813-816: the raw method is left out of the -z flavour")
    K._matrix(MOCK_ZIP, K.REF_ZIP, cli_src, tmp_path, mname, marg, fname, fargs)


def test_cli_gz_ungz_on_mock(libs, cli_src, tmp_path, monkeypatch):
    if not (os.path.exists(MOCK_GZ) and os.path.exists(K.REF_GZ)):
        pytest.skip("CLI binaries need the reference sources at build time")
    monkeypatch.setattr(K, "HIP_GZ", MOCK_GZ)
    K.test_cli_gz_ungz(cli_src, tmp_path)


def test_streams_with_a_prime_cache_present(libs):
    """With an archive primed somewhere in the process, a READ stream's first pull asks for only the 256 bytes the prime
    lookup compares (shim_zlib.c / shim_lzma.c pull_chunk); when the lookup misses, the ordinary decode goes on from
    there.  MZMOCK_PRIME_ANY=1 makes the mock say "something is primed" while every lookup misses: the stream, archive
    and CLI bodies above must pass unchanged (a fresh process: the mock reads the variable once)."""
    import sys

    env = dict(os.environ, MZMOCK_PRIME_ANY="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", os.path.abspath(__file__), "-k",
                        "not prime_cache_present"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


def test_window_mode_streams():
    """Entries larger than one decode window are decoded window by window (shim_zlib.c stream_next: the last 32 KiB carried
    as history, the compressed bytes in front of the current block dropped, the decode taken up at a token boundary the
    kernel reports).  Built here with a 192 KiB window and 48 KiB input gulps so that streams of a few hundred KiB cross
    many windows: same read() return values, bytes, TOTAL_IN / TOTAL_OUT, close() and error() as the reference, for whole,
    truncated and corrupted streams, Huffman and stored blocks, reads smaller and larger than a window."""
    import random

    from tests import synth

    if not os.path.isdir("/root/reference") or not oracle.have_ref():
        pytest.skip("needs the reference sources at build time")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul"), "B=_build_small",
                    'SHIM_DEFS=-DMZH_STREAM_WINDOW="(192<<10)" -DMZH_STREAM_GULP="(48<<10)"'], check=True, capture_output=True)
    hip = oracle.MzDriver(os.path.join(ROOT, "tests", "emul", "_build_small", "libmockdrop.so"))
    ref = oracle.ref()
    text, _ = synth.bench_corpus()
    rnd = random.Random(9)
    cases = [text[:450000], bytes(1200000), bytes(rnd.getrandbits(8) for _ in range(300000)), text[1000:150000] + bytes(250000) + text[:90000]]
    for i, d in enumerate(cases):
        for lvl in (6, 0):
            z = synth.deflate_raw(d, lvl)
            for chunk in (65535, 1000, 300000):
                assert ref.stream_decode(8, z, len(d) + 10, chunk=chunk) == hip.stream_decode(8, z, len(d) + 10, chunk=chunk), (i, lvl, chunk)
            for cut in (len(z) // 2, len(z) - 3):
                assert ref.stream_decode(8, z[:cut], len(d) + 10) == hip.stream_decode(8, z[:cut], len(d) + 10), (i, lvl, "cut", cut)
            zz = bytearray(z)
            zz[len(zz) * 2 // 3] ^= 0x10
            assert ref.stream_decode(8, bytes(zz), len(d) + 10) == hip.stream_decode(8, bytes(zz), len(d) + 10), (i, lvl, "flip")
    # delete() without close() after reads that went into window mode (legal in the reference, which leaks there): the
    # window-mode piece tables used to be freed twice -- glibc aborts the process on that
    d = text[:450000] * 2
    z = synth.deflate_raw(d, 6)
    for _ in range(3):
        got = hip.stream_delete_unclosed(8, z, len(d), chunk=65535, nreads=6)
        assert got == d[:len(got)] and len(got) == 6 * 65535
    assert ref.stream_delete_unclosed(8, z, len(d), chunk=65535, nreads=6) == d[:6 * 65535]
    # ... and through the zip layer: mz_zip_entry_read hands every 65 535 bytes it read to mz_crypt_crc32_update, and in window
    # mode those calls are answered from the CRCs the device computed of each window in exactly those pieces -- all but the
    # reads that straddle two windows (a launch per 64 KiB call made a 3 GiB entry take 22 s)
    import ctypes as C
    import tempfile
    import numpy as np
    L = C.CDLL(os.path.join(ROOT, "tests", "emul", "_build_small", "libmockdrop.so"))
    datas = [text[:450000] * 3, bytes(700000), text[:70000]]
    blob = np.frombuffer(b"".join(datas), dtype=np.uint8)
    lens = np.array([len(d) for d in datas], dtype=np.int32)
    offs = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "w.zip")
        ref.zip_write(path, blob, offs, lens, method=8, level=6)
        table = ref.zip_index(path)
        out = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
        c0, s0 = L.mzmock_crc_host_calls(), L.mzmock_seg_calls()
        _, crc, ulen, st = hip.zip_read_all(path, table[:, 6].copy(), nthreads=1, own_crc=False, out=out, out_off=offs)
        assert (st == 0).all() and (ulen == lens).all() and out[:-1].tobytes() == blob.tobytes()   # st 0 = the zip layer's CRC check passed
        launches, windows = L.mzmock_crc_host_calls() - c0, L.mzmock_seg_calls() - s0
        reads = int(sum((int(n) + 65534) // 65535 for n in lens))
        print("window mode through the zip layer: %d reads, %d windows, %d checksum launches" % (reads, windows, launches))
        assert windows >= 8 and launches <= windows + 4 and launches < reads // 2, (launches, windows, reads)


def test_window_mode_many_waves():
    """A large entry is decoded by a wave per DEFLATE block whenever window mode stands at a block header
    (mzhip_inflate_parallel_host, csrc/inflate_parallel.inc: header search, a parse per candidate, the chain from the known
    header, source map, pointer jumping), and by the serial kernel -- asked to stop at the next block header -- where that
    declines.  Built with a 1.5 MiB window, 256 KiB gulps and low thresholds so that streams of a few MiB go through many
    windows of both kinds: same read() return values, bytes, TOTAL_IN / TOTAL_OUT, close() and error() as the reference for
    whole, truncated and corrupted streams made of dynamic, fixed and stored blocks."""
    import ctypes as C
    import random
    import zlib

    from tests import synth

    if not os.path.isdir("/root/reference") or not oracle.have_ref():
        pytest.skip("needs the reference sources at build time")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul"), "B=_build_par",
                    'SHIM_DEFS=-DMZH_STREAM_WINDOW="(1536<<10)" -DMZH_STREAM_GULP="(256<<10)" -DMZH_PAR_MIN_IN="(32<<10)" '
                    '-DMZH_PAR_MIN_ROOM="(128<<10)" -DMZH_STREAM_EARLY="(64<<10)"'], check=True, capture_output=True)
    so = os.path.join(ROOT, "tests", "emul", "_build_par", "libmockdrop.so")
    hip = oracle.MzDriver(so)
    L = C.CDLL(so)
    ref = oracle.ref()
    text, _ = synth.bench_corpus()
    rnd = random.Random(11)
    noise = bytes(rnd.getrandbits(8) for _ in range(200000))

    def blocks_of(parts):
        """one raw stream out of pieces compressed with their own settings (full flushes between them: every piece starts
        on a block boundary, fixed-Huffman pieces via Z_FIXED, stored ones via level 0)"""
        out = b""
        for data, lvl, strat in parts[:-1]:
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strat)
            out += co.compress(data) + co.flush(zlib.Z_FULL_FLUSH)
        data, lvl, strat = parts[-1]
        co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strat)
        return out + co.compress(data) + co.flush()

    big = text[:900000] * 3 + bytes(400000) + text[200000:700000]
    cases = [
        ("dynamic", big, synth.deflate_raw(big, 6)),
        ("level 1", big[:2000000], synth.deflate_raw(big[:2000000], 1)),
        ("mixed", None, blocks_of([(text[:600000], 6, 0), (text[:300000], 6, zlib.Z_FIXED), (noise, 0, 0), (text[100000:900000], 9, 0),
                                   (noise[:70000], 6, 0), (text[:500000], 6, zlib.Z_FIXED), (text[:500000], 6, 0)])),
        ("fixed only", None, blocks_of([(text[:700000], 6, zlib.Z_FIXED)] * 3)),
    ]
    for name, d, z in cases:
        if d is None:
            d = zlib.decompress(z, -15)
        b0, c0 = L.mzmock_par_blocks(), L.mzmock_par_calls()
        for chunk in (65535, 300000):
            assert ref.stream_decode(8, z, len(d) + 10, chunk=chunk) == hip.stream_decode(8, z, len(d) + 10, chunk=chunk), (name, chunk)
        print("%s: %d bytes -> %d; many-wave windows %d, blocks %d" % (name, len(z), len(d), L.mzmock_par_calls() - c0, L.mzmock_par_blocks() - b0))
        if name in ("dynamic", "level 1", "mixed"):
            assert L.mzmock_par_blocks() - b0 >= 8, name
        for cut in (len(z) // 2, len(z) - 3):
            assert ref.stream_decode(8, z[:cut], len(d) + 10) == hip.stream_decode(8, z[:cut], len(d) + 10), (name, "cut", cut)
        for where in (len(z) // 3, len(z) * 2 // 3):
            zz = bytearray(z)
            zz[where] ^= 0x10
            cap = 2 * len(d) + (1 << 20)   # (room for whatever the damaged stream decodes to: TOTAL_IN of a caller that stops mid-stream is an estimate)
            a, b = ref.stream_decode(8, bytes(zz), cap), hip.stream_decode(8, bytes(zz), cap)
            # (where the base stream stands after a data error is not compared: window mode pulls up to a gulp ahead of the decode)
            a.pop("base_pos"), b.pop("base_pos")
            assert a == b, (name, "flip", where)
    # one window ahead: while a window is served, a thread of the stream decodes the next one (the default; everything above ran
    # that way) -- and with that off the device call and the serving take turns, as up to round 5: the same answers
    L.mzhip_stream_lookahead_windows.restype = C.c_uint64
    assert L.mzhip_stream_lookahead_windows() >= 20
    L.mzhip_set_stream_lookahead(0)
    w0 = L.mzhip_stream_lookahead_windows()
    for name, d, z in cases[:3]:
        d = d if d is not None else zlib.decompress(z, -15)
        for chunk in (65535, 300000, 7777):
            assert ref.stream_decode(8, z, len(d) + 10, chunk=chunk) == hip.stream_decode(8, z, len(d) + 10, chunk=chunk), (name, chunk, "no look-ahead")
    assert L.mzhip_stream_lookahead_windows() == w0
    L.mzhip_set_stream_lookahead(1)
    name, d, z = cases[0]
    assert ref.stream_decode(8, z, len(d) + 10, chunk=7777) == hip.stream_decode(8, z, len(d) + 10, chunk=7777)
    assert L.mzhip_stream_lookahead_windows() > w0
    # the switch: the same streams with the many-wave decode off take the serial windows only
    L.mzhip_set_stream_parallel(0)
    b0 = L.mzmock_par_blocks()
    name, d, z = cases[0]
    assert ref.stream_decode(8, z, len(d) + 10) == hip.stream_decode(8, z, len(d) + 10)
    assert L.mzmock_par_blocks() == b0
    L.mzhip_set_stream_parallel(1)


def test_large_entry_where_it_lies():
    """mz_large_entry (csrc/inflate_parallel.inc): one large device-resident entry window after window -- many-wave windows
    where a window starts at a block header, the serial kernel with "stop at the next block header" where not -- on the
    mock (2 MiB windows, the emulated device functions): bytes, consumed count, CRC and status against zlib for streams of
    dynamic, fixed and stored blocks; a buffer that is too small, a truncated and a corrupted stream."""
    import ctypes as C
    import random
    import zlib

    import numpy as np

    from tests import synth

    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul")], check=True, capture_output=True)
    L = C.CDLL(os.path.join(ROOT, "tests", "emul", "_build", "libmockdrop.so"))
    L.mzhip_inflate_large.restype = C.c_int32
    L.mzhip_inflate_large.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
    text = synth.bench_corpus()[0]
    rnd = random.Random(5)
    noise = bytes(rnd.getrandbits(8) for _ in range(150000))

    def blocks_of(parts):
        out = b""
        for i, (data, lvl, strat) in enumerate(parts):
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strat)
            out += co.compress(data) + (co.flush(zlib.Z_FULL_FLUSH) if i + 1 < len(parts) else co.flush())
        return out

    def run(z, cap):
        zin = np.frombuffer(z + bytes(64), dtype=np.uint8).copy()
        out = np.zeros(cap + 64, dtype=np.uint8)
        ol, iu, crc, st = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int32()
        assert L.mzhip_inflate_large(zin.ctypes.data, len(z), out.ctypes.data, cap, C.byref(ol), C.byref(iu), C.byref(crc), C.byref(st), None) == 0
        s = (C.c_uint32 * 3)()
        L.mzmock_large_stats(s)
        return st.value, ol.value, iu.value, crc.value, out[:ol.value].tobytes(), list(s)

    big = text * 7 + bytes(2000000) + text[::-1] * 2
    cases = [("dynamic", blocks_of([(big, 6, 0)])), ("level 1", blocks_of([(big, 1, 0)])),
             ("mixed", blocks_of([(text * 3, 6, 0), (text, 6, zlib.Z_FIXED), (noise, 0, 0), (text * 4, 9, 0), (noise[:70000], 6, 0),
                                  (text * 2, 6, zlib.Z_FIXED), (text * 3, 6, 0)])),
             ("fixed", blocks_of([(text * 2, 6, zlib.Z_FIXED)] * 3)), ("stored", blocks_of([(noise * 8, 0, 0)]))]
    for name, z in cases:
        d = zlib.decompress(z, -15)
        st, ol, iu, crc, got, s = run(z, len(d))
        assert (st, ol, iu, crc) == (0, len(d), len(z), zlib.crc32(d)) and got == d, name
        print("%s: %d -> %d bytes; many-wave windows %d (%d blocks), serial calls %d" % (name, len(z), len(d), s[0], s[1], s[2]))
        if name != "fixed":
            assert s[1] >= 10, (name, s)
        st, ol, _, _, got, _ = run(z, len(d) - 1000)                      # the entry is larger than it says
        assert st == -200 and got == d[:ol] and ol <= len(d) - 1000, name
        st, ol, iu, _, got, _ = run(z[:len(z) // 2], len(d))               # the stream ends short
        assert st == -5 and got == d[:ol], name
        zz = bytearray(z)
        zz[len(z) // 3] ^= 0x10
        st, ol, _, crc, got, _ = run(bytes(zz), 2 * len(d) + 100000)
        try:
            want = zlib.decompressobj(-15).decompress(bytes(zz))
            assert st == 0 and got == want and crc == zlib.crc32(want), name
        except zlib.error:
            assert st == -3, (name, st)


def test_write_segments_coded_ahead():
    """mz_stream_zlib WRITE hands a full segment to a thread of the stream and goes on collecting the next one
    (shim_zlib.c write_ahead_*; 8 MiB segments in the product, 256 KiB in this build so that a few MB cross many): the
    stream that reaches the base is byte for byte the one the synchronous path writes (mzhip_set_write_overlap(0)), with the
    same totals and close() / error() results, raw, zlib- and gzip-wrapped, for writes of 65 535, 1 000 and 700 000 bytes;
    the reference inflates it to the input."""
    import ctypes as C
    import zlib

    from tests import synth

    if not os.path.isdir("/root/reference") or not oracle.have_ref():
        pytest.skip("needs the reference sources at build time")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul"), "B=_build_wseg", 'SHIM_DEFS=-DMZH_WRITE_SEGMENT="(256<<10)"'],
                   check=True, capture_output=True)
    so = os.path.join(ROOT, "tests", "emul", "_build_wseg", "libmockdrop.so")
    hip = oracle.MzDriver(so)
    L = C.CDLL(so)
    text, _ = synth.bench_corpus()
    data = text[:900000] * 2 + bytes(300000) + text[5000:400000]
    for level in (1, 6):
        for wb, inflate_bits in ((0, -15), (15, 15), (31, 31)):
            for chunk in (65535, 1000, 700000):
                L.mzhip_set_write_overlap(1)
                a = hip.stream_encode(8, data, level=level, chunk=chunk, window_bits=wb)
                L.mzhip_set_write_overlap(0)
                b = hip.stream_encode(8, data, level=level, chunk=chunk, window_bits=wb)
                L.mzhip_set_write_overlap(1)
                assert a == b, (level, wb, chunk)
                assert zlib.decompress(a[0], inflate_bits) == data, (level, wb, chunk)
                assert a[1]["total_in"] == len(data) and a[1]["close"] == 0 and a[1]["error"] == 0
    # a stream that ends exactly where a segment ends, and one byte either side of it
    for n in (4 * (256 << 10) - 1, 4 * (256 << 10), 4 * (256 << 10) + 1, 256 << 10):
        d = (text * 4)[:n]
        a = hip.stream_encode(8, d, level=1, chunk=65535)
        assert zlib.decompress(a[0], -15) == d, n
