"""CPU tests of the DEVICE code's control flow via the 64-lane host emulation (tests/emul/emul.cpp
compiles minizip-ng_amd/csrc/*_core.h with -DMZHIP_HOST_EMUL).  This is debugging infrastructure
for a container without a GPU -- the product path is the HIP build of the same headers, exercised
by the -m gpu tests through the C ABI.  Checked against the oracle on the same inputs."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

import oracle
from tests import synth
from tests.test_oracle import _zip_lzma

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_u8p = C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def emu():
    out = os.path.join(ROOT, "tests", "emul", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libemul.so")
    subprocess.run(["g++", "-O1", "-g", "-Wno-unknown-pragmas", "-DMZHIP_HOST_EMUL"] + os.environ.get("MZ_EMUL_DEFS", "").split() +  # (build knobs of the device code, e.g. -DMZ_CRING_DW=8u)
                   ["-I" + os.path.join(ROOT, "minizip-ng_amd", "csrc"), "-shared", "-fPIC",
                    os.path.join(ROOT, "tests", "emul", "emul.cpp"), "-o", so], check=True)
    L = C.CDLL(so)
    L.emul_inflate.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32] + [C.POINTER(C.c_uint32)] * 3
    L.emul_inflate_steps.argtypes = L.emul_inflate.argtypes
    L.emul_lzma.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, C.c_int64] + [C.POINTER(C.c_uint32)] * 3
    L.emul_xz.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, C.c_int64] + [C.POINTER(C.c_uint32)] * 3
    L.emul_lzma_encode.argtypes = [_u8p, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.emul_deflate.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.emul_adler32.restype = C.c_uint32
    L.emul_adler32.argtypes = [_u8p, C.c_uint32]
    L.emul_adler32_combine.restype = C.c_uint32
    L.emul_adler32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
    L.emul_crc64.restype = C.c_uint64
    L.emul_crc64.argtypes = [_u8p, C.c_uint64]
    L.emul_sha.restype = None
    L.emul_sha.argtypes = [_u8p, C.c_uint64, C.c_int, _u8p]
    L.emul_crc32_super.restype = C.c_uint32
    L.emul_crc32_super.argtypes = [_u8p, C.c_uint32, C.c_uint32]
    L.emul_crc32.restype = C.c_uint32
    L.emul_crc32.argtypes = [_u8p, C.c_uint32]
    return L


def _run(fn, z, cap, *extra, mis=0, omis=0):
    """mis / omis: byte misalignment of the input / output pointers handed to the core (0..3)"""
    a = np.zeros(len(z) + 8, np.uint8)
    a[mis:mis + len(z)] = np.frombuffer(z, dtype=np.uint8)
    out = np.zeros(cap + 8, np.uint8)
    ol, iu, crc = C.c_uint32(), C.c_uint32(), C.c_uint32()
    pin = C.cast(a.ctypes.data + mis, _u8p)
    pout = C.cast(out.ctypes.data + omis, _u8p)
    st = fn(pin, len(z), pout, cap, *extra, C.byref(ol), C.byref(iu), C.byref(crc))
    return st, iu.value, out[omis:omis + ol.value].tobytes(), crc.value


def test_lds_budget(emu):
    # 4 waves x inflate slice (tables + the span path's window + the window's staging / match-list pool) + CRC table
    # must allow 2 workgroups (8 waves = 2 per SIMD, the kernel's launch bound) per 160 KiB CU
    assert (4 * ((emu.emul_lds_bytes() + 15) // 16 * 16) + 1024) * 2 <= 160 * 1024
    assert emu.emul_lzma_lds_bytes() + 1024 <= 16 * 1024 + 1024


def test_crc_tiles_and_tail(emu):
    rnd = np.random.RandomState(0)
    for n in (0, 1, 15, 16, 17, 1023, 1024, 1025, 2047, 2048, 5000, 65535, 65536, 100001):
        d = rnd.bytes(n)
        a = np.frombuffer(d, dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)
        assert emu.emul_crc32(a.ctypes.data_as(_u8p), n) == zlib.crc32(d) == oracle.crc32(d), n


def test_crc_super_tiles(emu):
    """The stand-alone CRC kernel's path (4 KiB super-tiles + 1 KiB tiles + tail) with chaining values."""
    rnd = np.random.RandomState(7)
    for n in (0, 1, 1023, 4095, 4096, 4097, 8191, 8192, 12288 + 1024 + 5, 65536, 100001, 300000):
        d = rnd.bytes(n)
        a = np.frombuffer(d, dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)
        for init in (0, 0xDEADBEEF):
            assert emu.emul_crc32_super(a.ctypes.data_as(_u8p), n, init) == zlib.crc32(d, init), (n, init)


def test_adler32_tiles_tail_and_combine(emu):
    """K5 (zlib-wrapper trailer) against zlib.adler32: tile boundaries, 0xFF worst case for the modular sums,
    and adler(A||B) from the two halves."""
    rnd = np.random.RandomState(1)
    for n in (0, 1, 15, 16, 17, 1023, 1024, 1025, 5000, 65535, 65536, 100001, 1 << 20):
        for d in (rnd.bytes(n), b"\xff" * n):
            a = np.frombuffer(d, dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)
            assert emu.emul_adler32(a.ctypes.data_as(_u8p), n) == zlib.adler32(d), n
            k = n // 3
            assert emu.emul_adler32_combine(zlib.adler32(d[:k]), zlib.adler32(d[k:]), n - k) == zlib.adler32(d), n


def test_crc64_and_sha_vs_hashlib(emu):
    """The .xz block checks (CRC-64, SHA-256) and the row-4 hashes (SHA-1/224/256): piece alignment, padding
    boundaries (55/56/63/64 bytes), and the reference's own KAT string (test/test_crypt.cc:26,50-116)."""
    import hashlib

    rnd = np.random.RandomState(4)
    kat = b"the quick and lazy fox did his thang"
    for n in (0, 1, 3, 55, 56, 57, 63, 64, 65, 111, 112, 113, 119, 120, 127, 128, 129, 239, 240, 1000, 4097, 100001):
        d = rnd.bytes(n)
        a = np.frombuffer(d, dtype=np.uint8).copy() if n else np.zeros(1, np.uint8)
        assert emu.emul_crc64(a.ctypes.data_as(_u8p), n) == oracle.crc64(d), n
        for alg, fn in ((20, hashlib.sha1), (22, hashlib.sha224), (23, hashlib.sha256), (24, hashlib.sha384),
                        (25, hashlib.sha512)):
            out = np.zeros(64, np.uint8)
            emu.emul_sha(a.ctypes.data_as(_u8p), n, alg, out.ctypes.data_as(_u8p))
            assert out.tobytes()[:fn().digest_size] == fn(d).digest(), (n, alg)
    a = np.frombuffer(b"123456789", dtype=np.uint8).copy()
    assert emu.emul_crc64(a.ctypes.data_as(_u8p), 9) == 0x995DC9BBDF1939FA
    a = np.frombuffer(kat, dtype=np.uint8).copy()
    for alg, want in ((20, "3efb8392b6cd8e14bd76bd08081521dc73df418c"),
                      (22, "9e444f5f0b6582a923bd48696155f4a2f0d914e044cb64b8729a6600"),
                      (23, "7a31ea0848525f7ebfeec9ee532bcc5d6d26772427e097b86cf440a56546541c"),
                      (24, "e1e42e5977965bb3621231a5df3a1e83c471fa91fde33b6a30c8c4fa0d8be29ba7171c7c9487db91e9ee7e85049f7b41"),
                      (25, "6627e7643ee7ce633e03f52d22329c3a32597364247c5275d4369985e1518626da46f595ad327667346479d246359b8b381af"
                           "791ce2ac8c53a4788050eea11fe")):
        out = np.zeros(64, np.uint8)
        emu.emul_sha(a.ctypes.data_as(_u8p), len(kat), alg, out.ctypes.data_as(_u8p))
        assert out.tobytes()[:len(want) // 2].hex() == want, alg


def test_inflate_edges(emu):
    for name, data, z in synth.edge_payloads():
        st, used, out, crc = _run(emu.emul_inflate, z + b"\x00junk", len(data) + 8)
        so, uo, oo = oracle.inflate_raw(z + b"\x00junk", len(data) + 8)
        assert (st, used, out) == (so, uo, oo) == (0, len(z), data), name
        assert crc == oracle.crc32(data), name


def test_inflate_long_codes(emu):
    """Codes of 13-15 bits over the full alphabet: the second-level tables (capacity = the exhaustive-search bound
    for the root width, inflate_core.h) and the long-code paths."""
    n = 0
    for name, data, z in synth.long_code_payloads():
        st, used, out, crc = _run(emu.emul_inflate, z, len(data) + 8)
        assert (st, used, out, crc) == (0, len(z), data, zlib.crc32(data)), name
        so, uo, oo = oracle.inflate_raw(z, len(data) + 8)
        assert (so, uo, oo) == (0, len(z), data), name
        n += 1
    assert n >= 8


def test_inflate_fixtures(emu, fixtures):
    for e in fixtures:
        if e["method"] != 8:
            continue
        st, used, out, crc = _run(emu.emul_inflate, e["payload"], e["usize"] + 4)
        assert (st, used, len(out), crc) == (0, e["csize"], e["usize"], e["crc"]), (e["archive"], e["entry"])


def test_inflate_malformed_status(emu):
    n = 0
    for name, data, z in synth.edge_payloads():
        if len(z) < 16:
            continue
        for cname, bad in synth.corruptions(z):
            cap = len(data) + 70000
            st, used, out, crc = _run(emu.emul_inflate, bad, cap)
            so, uo, oo = oracle.inflate_raw(bad, cap)
            assert st == so, (name, cname, st, so)
            if so == 0:
                assert (used, out) == (uo, oo), (name, cname)
            n += 1
    assert n > 100


def test_inflate_out_cap(emu):
    data = synth.corpus()[:30000]
    z = synth.deflate_raw(data)
    st, used, out, crc = _run(emu.emul_inflate, z, len(data) - 1)
    assert st == -200


def test_lzma_slot_build(emu):
    """K3's main kernel keeps MZ_LZMA_SLOTS literal contexts in LDS and the literal model in HBM (lzma_core.h
    LZ_LITERAL_SITE_SLOT): whatever it decodes itself must be what the full-model build decodes -- bytes, consumed count,
    CRC, verdicts of corrupted streams -- and what it gives back (MZHIP_RETRY = -300: the contexts keep swapping) must be
    data that deserves it: text at the parameters every ZIP writer uses is never given back, random bytes always."""
    import lzma as pylzma
    import random

    emu.emul_lzma_slots.argtypes = emu.emul_lzma.argtypes
    assert emu.emul_lzma_slots_lds_bytes() * 4 + 1024 <= 40 * 1024          # four waves + one CRC table per workgroup, four per CU
    text, _ = synth.bench_corpus()
    rnd = random.Random(4)
    noise = bytes(rnd.getrandbits(8) for _ in range(40000))
    cases = [("empty", b""), ("one", b"a"), ("text", text[:250000]), ("markov", synth.markov_entries(1, 300000, 5, text)[0]),
             ("noise", noise), ("run", b"A" * 100000), ("mixed", text[1000:70000] + noise[:3000] + text[:50000])]
    back = {}
    for name, d in cases:
        for lc, lp, pb in ((3, 0, 2), (0, 0, 2), (4, 0, 0), (1, 2, 2), (0, 4, 1)):
            raw = pylzma.compress(d, format=pylzma.FORMAT_ALONE, filters=[dict(id=pylzma.FILTER_LZMA1, preset=6, lc=lc, lp=lp, pb=pb)])
            z = bytes([5, 2, 5, 0]) + raw[:5] + raw[13:]
            a = _run(emu.emul_lzma, z, len(d) + 10, C.c_int64(len(d)))
            b = _run(emu.emul_lzma_slots, z, len(d) + 10, C.c_int64(len(d)))
            back[(name, lc, lp)] = b[0] == -300
            if b[0] != -300:
                assert a == b and a[2] == d and a[3] == zlib.crc32(d), (name, lc, lp, pb, a[0], b[0])
        z = bytearray(_zip_lzma(d))
        if len(z) > 40:
            z[len(z) // 2] ^= 0x21
            a = _run(emu.emul_lzma, bytes(z), len(d) + 100, C.c_int64(-1))
            b = _run(emu.emul_lzma_slots, bytes(z), len(d) + 100, C.c_int64(-1))
            if b[0] != -300:
                assert a[:3] == b[:3], (name, "corrupt", a[0], b[0])
    assert not back[("text", 3, 0)] and not back[("markov", 3, 0)] and not back[("run", 3, 0)] and not back[("text", 0, 0)]
    assert back[("noise", 3, 0)] and back[("text", 0, 4)]


def test_lzma_cases(emu):
    c = synth.corpus()
    rnd = np.random.RandomState(11)
    cases = [b"", b"a", c[:1000], c[:150000], rnd.bytes(5000), b"A" * 100000, c[1000:70000] + rnd.bytes(3000) + c[:50000]]
    for i, d in enumerate(cases):
        z = _zip_lzma(d)
        st, used, out, crc = _run(emu.emul_lzma, z, len(d) + 64, C.c_int64(len(d)))
        so, uo, oo = oracle.lzma_zip_decode(z, len(d) + 64, len(d))
        assert (st, used, out) == (so, uo, oo) == (0, len(z), d), i
        assert crc == oracle.crc32(d)
        if len(z) > 40:
            third = len(z) // 3
            for bad in (z[:len(z) // 2], z[:20], z[:9], z[:5], z[:third] + bytes([z[third] ^ 0x55]) + z[third + 1:],
                        z[:9] + b"\x01" + z[10:]):
                st, used, out, crc = _run(emu.emul_lzma, bad, len(d) + 70000, C.c_int64(-1))
                so, uo, oo = oracle.lzma_zip_decode(bad, len(d) + 70000, -1)
                if so == 0:
                    assert st == 0 and out == oo
                else:
                    assert st in (-3, -5), (i, len(bad), st)   # mz_stream_lzma_read maps both to MZ_DATA_ERROR
    # TOTAL_OUT_MAX clamp (mz_strm_lzma.c:214-215)
    z = _zip_lzma(c[:5000])
    st, used, out, crc = _run(emu.emul_lzma, z, 6000, C.c_int64(3000))
    assert st == 0 and out == c[:3000] and crc == zlib.crc32(c[:3000])
    # every lc / lp / pb liblzma accepts behind lzma_alone_decoder (mz_strm_lzma.c:126), lc + lp = 4 included
    import lzma as pylzma
    d = c[:90000] + bytes(range(256)) * 30
    for lc, lp, pb in ((4, 0, 2), (3, 1, 2), (0, 4, 0), (2, 2, 4), (1, 3, 1), (0, 0, 0), (3, 0, 4)):
        raw = pylzma.compress(d, format=pylzma.FORMAT_ALONE, filters=[dict(id=pylzma.FILTER_LZMA1, preset=6, lc=lc, lp=lp, pb=pb)])
        z = bytes([5, 2, 5, 0]) + raw[:5] + raw[13:]
        st, used, out, crc = _run(emu.emul_lzma, z, len(d) + 64, C.c_int64(len(d)))
        so, uo, oo = oracle.lzma_zip_decode(z, len(d) + 64, len(d))
        assert (st, used, out) == (so, uo, oo) == (0, len(z), d) and crc == zlib.crc32(d), (lc, lp, pb)
    bad = bytearray(_zip_lzma(c[:1000]))
    bad[4] = 4 + 9 * 1 + 45 * 2          # lc 4, lp 1: lc + lp > 4 is an options error -> MZ_DATA_ERROR
    st, used, out, crc = _run(emu.emul_lzma, bytes(bad), 2000, C.c_int64(-1))
    assert st == -3 == oracle.lzma_zip_decode(bytes(bad), 2000, -1)[0]


def _lzma_windows(emu, z, total, window, gulp, dict_keep):
    """decode z window by window through the resumable build, the way shim_lzma.c does: -> (bytes, consumed, final status)"""
    nmodel = emu.emul_lzma_model_u16()
    model = (C.c_uint16 * nmodel)()
    st = (C.c_uint32 * 16)()
    out = bytearray()
    buf = np.zeros(dict_keep + window + 16, dtype=np.uint8)
    hist = 0
    pos = 0       # compressed bytes done with
    have = 0      # compressed bytes handed over so far (pos + what the next call sees)
    resumed = False
    for _ in range(100000):
        have = min(len(z), max(have, pos + gulp))
        last = have == len(z)
        chunk = np.frombuffer(z[pos:have] + b"\0" * 8, dtype=np.uint8).copy()
        st[0] = (1 if resumed else 0) | (2 if last else 0)
        st[10] = hist
        sto = (C.c_uint32 * 16)()
        ol, iu = C.c_uint32(), C.c_uint32()
        rc = emu.emul_lzma_resume(chunk.ctypes.data_as(_u8p), have - pos, buf.ctypes.data_as(_u8p), hist + window, st, sto, model,
                                  C.byref(ol), C.byref(iu))
        out += buf[hist:ol.value].tobytes()
        pos += iu.value
        if rc == 0 or not sto[0]:
            return bytes(out), pos, rc
        # stopped in front of a packet: keep the dictionary (a multiple of 16 dropped in front), go on
        resumed = True
        for k in range(16):
            st[k] = sto[k]
        keep = min(ol.value, dict_keep)
        drop = (ol.value - keep) & ~15
        keep = ol.value - drop
        buf[:keep] = buf[drop:ol.value].copy()
        hist = keep
        if rc == -5 and last:
            return bytes(out), pos, rc
        if rc == -5:
            have = min(len(z), have + gulp)
    raise AssertionError("no progress")


def test_lzma_resumable_build(emu):
    """K3's resumable build (entries decoded window by window, shim_lzma.c): same bytes and consumed count as the one-shot
    decode for windows from 300 bytes to 64 KiB, input gulps down to 100 bytes, every lc / lp / pb class, a dictionary
    smaller than the entry; truncated streams end with -5 and the bytes up to the cut."""
    import lzma as pylzma
    import random

    emu.emul_lzma_resume.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rnd = random.Random(31)
    c = synth.corpus()
    cases = []
    for lc, lp, pb, dsz in ((3, 0, 2, 1 << 16), (0, 2, 0, 1 << 12), (4, 0, 4, 1 << 20), (1, 3, 1, 1 << 13), (0, 4, 2, 1 << 15)):
        d = c[rnd.randrange(1000):][:rnd.randrange(40000, 200000)] + bytes(rnd.randrange(256) for _ in range(3000)) + b"ab" * 5000
        raw = pylzma.compress(d, format=pylzma.FORMAT_ALONE, filters=[dict(id=pylzma.FILTER_LZMA1, lc=lc, lp=lp, pb=pb, dict_size=dsz)])
        cases.append((d, bytes([5, 2, 5, 0]) + raw[:5] + raw[13:], max(dsz, 4096)))
    for d, z, dsz in cases:
        for window, gulp in ((300, 100), (4096, 700), (65536, 20000), (1000, 1 << 30)):
            got, used, rc = _lzma_windows(emu, z, len(d), window, gulp, dsz)
            assert rc == 0 and got == d and used == len(z), (len(d), window, gulp, rc, len(got), used, len(z))
        cut = len(z) * 2 // 3
        got, used, rc = _lzma_windows(emu, z[:cut], len(d), 4096, 1000, dsz)
        so, uo, oo = oracle.lzma_zip_decode(z[:cut], len(d) + 64)  # the restatement of the one-shot decode
        assert rc == -5 and so < 0 and got == oo and d.startswith(got) and len(got) > 0, (rc, so, len(got), len(oo))


def _lzma2_windows(emu, z, window, gulp, dict_size, check_id=4, keep_all=False):
    """drive emul_lzma2_run as shim_lzma.c does: a sliding buffer [dictionary | window], input in gulps
    -> (bytes, consumed, status, check value, windows)"""
    emu.emul_lzma2_run.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    keep = (max(dict_size, 4096) + 15) & ~15
    cap = keep + window + 16
    buf = (C.c_uint8 * cap)()
    model = (C.c_uint16 * emu.emul_lzma_model_u16())()
    st_in = (C.c_uint32 * 20)()
    st_out = (C.c_uint32 * 20)()
    st_in[0] = 4 | 8
    st_in[9] = dict_size
    st_in[15] = check_id
    got = bytearray()
    hist = 0
    pos = 0       # consumed input
    have = min(len(z), gulp)
    windows = 0
    while True:
        windows += 1
        assert windows < 200000
        last = have >= len(z)
        st_in[0] = (st_in[0] & ~2) | (2 if last else 0)
        st_in[10] = hist
        chunk = z[pos:have]
        src = (C.c_uint8 * max(len(chunk), 1)).from_buffer_copy(chunk or b"\0")
        ol, iu = C.c_uint32(0), C.c_uint32(0)
        room = min(cap, hist + window)
        rc = emu.emul_lzma2_run(src, len(chunk), buf, room, st_in, st_out, model, C.byref(ol), C.byref(iu))
        assert hist <= ol.value <= room and iu.value <= len(chunk)
        got += bytes(buf[hist:ol.value])
        f = st_out[0]
        if rc == 0:
            assert f & 16
            return bytes(got), pos + iu.value, 0, st_out[16] | (st_out[17] << 32), windows
        if not (f & 1):
            return bytes(got), pos + iu.value, rc, 0, windows
        pos += iu.value
        if rc == -5:
            if last and ol.value == hist:
                return bytes(got), pos, rc, 0, windows
            have = min(len(z), max(have, pos) + gulp)
        else:
            assert rc == -200
        # slide: keep the dictionary
        out_len = ol.value
        k = out_len if keep_all else min(out_len, keep)
        drop = (out_len - k) & ~15   # position contexts look at the low four bits of the position
        k = out_len - drop
        ds = st_out[12]
        if drop:
            C.memmove(buf, C.addressof(buf) + drop, k)
        hist = k
        for i in range(20):
            st_in[i] = st_out[i]
        st_in[12] = max(0, ds - drop)
        st_in[0] = (st_out[0] & ~16) | 1


def test_lzma2_windows(emu):
    """mz_lzma2_run (xz_core.h; method 95 READ in window mode): a block's LZMA2 chunk sequence decoded in windows of 300 bytes
    to 256 KiB from gulps of input down to 90 bytes gives the bytes and the CRC-64 / CRC-32 of the one-shot decode, for
    every lc / lp / pb class, dictionaries smaller than the data (the buffer slides), uncompressed chunks (noise), chunks
    that keep state and chunks that reset it; cut streams end with -5 and the bytes liblzma's decoder gives up to the cut;
    corrupted ones never run away."""
    import lzma as pylzma
    import random
    import binascii

    def crc64(b):
        # .xz's CRC-64 through liblzma itself: the check field of a one-block stream
        x = pylzma.compress(b, format=pylzma.FORMAT_XZ, check=pylzma.CHECK_CRC64, preset=0)
        return int.from_bytes(x[-12 - 8 - _index_len(x):][:8], "little")

    def _index_len(x):
        return (int.from_bytes(x[-8:-4], "little") + 1) * 4

    rnd = random.Random(77)
    c = synth.corpus()
    noise = bytes(rnd.randrange(256) for _ in range(150000))
    for lc, lp, pb, dsz, preset in ((3, 0, 2, 1 << 16, 6), (0, 2, 0, 1 << 12, 1), (4, 0, 4, 1 << 20, 6), (1, 3, 1, 1 << 13, 0), (0, 4, 2, 1 << 15, 9)):
        d = c[rnd.randrange(1000):][:rnd.randrange(100000, 300000)] + noise[:rnd.randrange(70000, 150000)] + b"ab" * 50000 + c[:50000]
        z = pylzma.compress(d, format=pylzma.FORMAT_RAW, filters=[dict(id=pylzma.FILTER_LZMA2, preset=preset, lc=lc, lp=lp, pb=pb, dict_size=dsz)])
        want64 = crc64(d)
        for window, gulp in ((300, 90), (4096, 700), (65536, 20000), (1000, 1 << 30), (262144, 70000)):
            cid = 4 if window != 4096 else 1
            got, used, rc, chk, nwin = _lzma2_windows(emu, z, window, gulp, dsz, cid)
            assert rc == 0 and got == d and used == len(z), (lc, lp, pb, window, gulp, rc, len(got), len(d), used, len(z))
            assert chk == (want64 if cid == 4 else zlib.crc32(d)), (window, cid, hex(chk))
        cut = len(z) * 2 // 3
        got, used, rc, _, _ = _lzma2_windows(emu, z[:cut], 4096, 1000, dsz)
        dec = pylzma.LZMADecompressor(format=pylzma.FORMAT_RAW, filters=[dict(id=pylzma.FILTER_LZMA2, dict_size=dsz)])
        ref = dec.decompress(z[:cut])
        assert rc == -5 and got == ref and len(got) > 0, (rc, len(got), len(ref))
        for it in range(12):
            zz = bytearray(z)
            zz[rnd.randrange(len(zz))] ^= 1 << rnd.randrange(8)
            got, used, rc, _, _ = _lzma2_windows(emu, bytes(zz), 30000, 9000, dsz)
            dec = pylzma.LZMADecompressor(format=pylzma.FORMAT_RAW, filters=[dict(id=pylzma.FILTER_LZMA2, dict_size=dsz)])
            try:
                ref = dec.decompress(bytes(zz))
                ok = dec.eof
            except pylzma.LZMAError:
                ref, ok = None, False
            if ok:
                assert rc == 0 and got == ref
            else:
                assert rc != 0 or (ref is not None and got[:len(ref)] == ref), (it, rc)


def test_lzma_fixture(emu, fixtures):
    for e in fixtures:
        if e["method"] != 14:
            continue
        st, used, out, crc = _run(emu.emul_lzma, e["payload"], e["usize"] + 4, C.c_int64(e["usize"]))
        assert (st, used, len(out), crc) == (0, e["csize"], e["usize"], e["crc"])


def test_xz_cases_and_fuzz(emu, fixtures):
    """The .xz kernel (container + LZMA2 + checks) in emulation vs the oracle: every generated case, the xz.zip
    fixture, the TOTAL_OUT_MAX clamp, out_cap, and 1500 corrupted / truncated streams (same accept / reject
    decision; on accept the same bytes, consumed input and CRC)."""
    import random

    assert emu.emul_xz_lds_bytes() + 1024 <= 19 * 1024        # 8 single-wave workgroups per 160 KiB CU
    cases = synth.xz_cases()
    n_lclp4 = 0
    for name, d, x in cases:
        st, used, out, crc = _run(emu.emul_xz, x + b"tail", len(d) + 64, C.c_int64(-1))
        n_lclp4 += ("lp4" in name or "lc4" in name or "lc1lp3" in name) and len(d) > 1
        assert (st, used, out, crc) == (0, len(x), d, zlib.crc32(d)), (name, st, used, len(x))
    assert n_lclp4 >= 4          # lc + lp = 4 included: the upper half of the literal model lives outside the LDS slice
    for e in fixtures:
        if e["method"] == 95:
            st, used, out, crc = _run(emu.emul_xz, e["payload"], e["usize"] + 4, C.c_int64(e["usize"]))
            assert (st, used, len(out), crc) == (0, e["csize"], e["usize"], e["crc"])
    name, d, x = cases[0]
    st, used, out, crc = _run(emu.emul_xz, x, len(d) + 64, C.c_int64(3000))
    assert st == 0 and out == d[:3000] and crc == zlib.crc32(d[:3000])
    st, used, out, crc = _run(emu.emul_xz, x, len(d) - 1, C.c_int64(-1))
    assert st == -200
    assert sum(n.startswith("filter/") for n, _, _ in cases) >= 70     # Delta / BCJ chains in front of LZMA2 included
    for name, x in synth.xz_bad_chain_cases():                         # chains liblzma refuses: data errors
        assert _run(emu.emul_xz, x, 10000, C.c_int64(-1))[0] == -3, name
    rnd = random.Random(9)
    bases = [x for n, d, x in cases if 0 < len(d) <= 100000 and "lp4" not in n and "lc4" not in n and "lc1lp3" not in n]
    for it in range(1500):
        x = bytearray(rnd.choice(bases))
        k = rnd.randrange(5)
        if k == 0:
            x[rnd.randrange(len(x))] ^= 1 << rnd.randrange(8)
        elif k == 1:
            x[rnd.randrange(len(x))] = rnd.randrange(256)
        elif k == 2:
            del x[rnd.randrange(1, len(x)):]
        elif k == 3:
            x[rnd.randrange(min(len(x), 40))] = rnd.randrange(256)
        else:
            x[-rnd.randrange(1, 40)] = rnd.randrange(256)
        x = bytes(x)
        st, used, out, crc = _run(emu.emul_xz, x, 200000, C.c_int64(-1))
        so, uo, oo = oracle.xz_decode(x, 200000)
        if so == 0:
            assert (st, used, out, crc) == (0, uo, oo, zlib.crc32(oo)), (it, k)
        elif so == -109:
            assert st in (-109, -3), (it, k, st)
        else:
            assert st == so or (st, so) in ((-109, -3),), (it, k, st, so)


def _deflate(emu, d, final=1):
    a = np.frombuffer(d, dtype=np.uint8).copy() if len(d) else np.zeros(1, np.uint8)
    cap = len(d) + len(d) // 8 + 64
    out = np.zeros(cap, np.uint8)
    ol, crc = C.c_uint32(), C.c_uint32()
    st = emu.emul_deflate(a.ctypes.data_as(_u8p), len(d), out.ctypes.data_as(_u8p), cap, final, C.byref(ol), C.byref(crc))
    return st, out[:ol.value].tobytes(), crc.value


def test_deflate_roundtrip(emu):
    """K4 parity = valid DEFLATE that the reference side inflates back to the input (oracle + zlib), CRC equal."""
    c = synth.corpus()
    rnd = np.random.RandomState(5)
    cases = [b"", b"a", b"abc", b"abcd", b"aaaa", b"A" * 1000, b"ab" * 500, c[:100], c[:65536], c[1234:1234 + 8192],
             rnd.bytes(5000), c[:200000], bytes(70000), b"x" * 63, b"x" * 64, b"x" * 65, c[:70000] + c[:70000]]
    for d in cases:
        st, z, crc = _deflate(emu, d)
        assert st == 0 and crc == zlib.crc32(d) == oracle.crc32(d), len(d)
        assert zlib.decompress(z, -15) == d, len(d)
        so, used, out = oracle.inflate_raw(z, len(d) + 8)
        assert (so, used, out) == (0, len(z), d), len(d)
        # and back through the decoder core
        st2, used2, out2, crc2 = _run(emu.emul_inflate, z, len(d) + 8)
        assert (st2, used2, out2, crc2) == (0, len(z), d, crc)
    assert _deflate(emu, b"")[1] == b"\x03\x00"            # the canonical empty fixed block
    assert len(_deflate(emu, c[:65536])[1]) < 0.5 * 65536   # it does compress text
    # non-final pieces concatenate on byte boundaries into one valid stream
    parts = [c[:30000], c[30000:90000], b"", c[90000:100000]]
    zs = b"".join(_deflate(emu, p, 0)[1] for p in parts[:-1]) + _deflate(emu, parts[-1], 1)[1]
    assert zlib.decompress(zs, -15) == b"".join(parts)
    # out_cap too small
    a = np.frombuffer(rnd.bytes(4000), dtype=np.uint8).copy()
    out = np.zeros(100, np.uint8)
    ol, crc = C.c_uint32(), C.c_uint32()
    assert emu.emul_deflate(a.ctypes.data_as(_u8p), 4000, out.ctypes.data_as(_u8p), 100, 1, C.byref(ol), C.byref(crc)) == -200


def _lzma_encode(emu, d, mode=0):
    a = np.frombuffer(d, dtype=np.uint8).copy() if len(d) else np.zeros(1, np.uint8)
    cap = len(d) + len(d) // 8 + 1024
    out = np.zeros(cap, np.uint8)
    ol, crc = C.c_uint32(), C.c_uint32()
    st = emu.emul_lzma_encode(a.ctypes.data_as(_u8p), len(d), mode, out.ctypes.data_as(_u8p), cap, C.byref(ol), C.byref(crc))
    return st, out[:ol.value].tobytes(), crc.value


def test_lzma_encode_long_history(emu):
    """K6's chain pass (lzma_enc_core.h mz_lz_chain + the links the block parse follows): matches reach back 8 MiB like
    liblzma's preset 6 (mz_strm_lzma.c:81).  Ratio bar of VERDICT r3 item 8 on config-4 entries (<= 0.30; round 3: 0.42),
    distances beyond one block really occur, liblzma and the oracle decode the streams."""
    import lzma as pylzma

    emu.emul_lzma_encode_ways.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    datas = synth.markov_entries(2, 1 << 20, 77, synth.bench_corpus()[0])
    tin = tout = 0
    for d in datas + [synth.corpus()[:200000] + synth.corpus()[:200000]]:
        a = np.frombuffer(d, dtype=np.uint8).copy()
        out = np.zeros(len(d) + len(d) // 8 + 4096, dtype=np.uint8)
        ol, crc = C.c_uint32(), C.c_uint32()
        assert emu.emul_lzma_encode_ways(C.cast(a.ctypes.data, _u8p), len(d), 0, 4, C.cast(out.ctypes.data, _u8p), len(out), C.byref(ol), C.byref(crc)) == 0
        z = out[:ol.value].tobytes()
        assert crc.value == zlib.crc32(d)
        assert z[:9] == bytes([9, 20, 5, 0, 0x5D, 0, 0, 0x80, 0])
        assert pylzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=pylzma.FORMAT_ALONE) == d
        assert oracle.lzma_zip_decode(z, len(d) + 64, -1) == (0, len(z), d)
        if len(d) == 1 << 20:
            tin += len(d)
            tout += len(z)
        else:
            assert len(z) < 0.62 * len(pylzma.compress(d[:200000], format=pylzma.FORMAT_RAW, filters=[{"id": pylzma.FILTER_LZMA1, "preset": 6}])) * 2  # the second copy costs next to nothing: it lies 200 000 bytes back
    assert tout <= 0.30 * tin, (tout, tin)


def test_lzma_encode_far_fuzz(emu):
    """Seeded structure fuzz of K6's long-range path (chain pass, links followed 1 / 4 / 8 / 16 deep, 273-byte matches,
    LZMA2 chunks over one dictionary): inputs made of segments copied from anywhere earlier -- across block boundaries, a few
    bytes and megabytes back, overlapping themselves -- between runs of text and noise, at sizes around multiples of the
    64 KiB block.  Every stream must decode with liblzma (method 14 as .lzma, the chunks framed as .xz) to the input."""
    import lzma as pylzma
    import random

    emu.emul_lzma_encode_ways.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    emu.emul_lzma2_chunks_encode.argtypes = [_u8p, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_uint32)]
    emu.emul_set_far_depth.argtypes = [C.c_uint32]
    text = synth.corpus()
    rnd = random.Random(41)
    sizes = [65536, 65537, 131072, 131071, 200000, 3 * 65536 + 5, 1 << 20, (1 << 20) + 70001, 2500000]
    try:
        for case, size in enumerate(sizes):
            buf = bytearray()
            while len(buf) < size:
                k = rnd.random()
                if k < 0.25 or len(buf) < 100:
                    o = rnd.randrange(len(text) - 3000)
                    buf += text[o:o + rnd.randrange(1, 3000)]
                elif k < 0.35:
                    buf += rnd.randbytes(rnd.randrange(1, 400))
                elif k < 0.45:
                    buf += bytes([rnd.randrange(256)]) * rnd.randrange(1, 700)
                else:                       # a copy from anywhere earlier (it may run into itself: distance < length)
                    d = rnd.choice([1, 2, 3, 7, 8, 9, 63, 64, 65, 4096, 32768, 32769, 65535, 65536, 65537, 100000, 1 << 20, (1 << 20) + 1,
                                    rnd.randrange(1, len(buf) + 1)])
                    d = min(d, len(buf))
                    n = rnd.choice([2, 3, 4, 5, 7, 8, 16, 64, 272, 273, 274, 600, 5000, rnd.randrange(1, 90000)])
                    for i in range(n):
                        buf.append(buf[len(buf) - d])
            data = bytes(buf[:size])
            a = np.frombuffer(data, dtype=np.uint8).copy()
            for ways, depth in ((1, 1), (4, 4), (4, 8), (4, 16)):
                if size > (1 << 20) and depth not in (1, 8):
                    continue
                emu.emul_set_far_depth(depth)
                out = np.zeros(size + size // 8 + 4096, dtype=np.uint8)
                ol, crc = C.c_uint32(), C.c_uint32()
                assert emu.emul_lzma_encode_ways(C.cast(a.ctypes.data, _u8p), size, 0, ways, C.cast(out.ctypes.data, _u8p), len(out), C.byref(ol), C.byref(crc)) == 0
                z = out[:ol.value].tobytes()
                assert crc.value == zlib.crc32(data)
                assert pylzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=pylzma.FORMAT_ALONE) == data, (case, size, ways, depth)
                if depth == 8:
                    st2, used2, out2, crc2 = _run(emu.emul_lzma, z, size + 64, C.c_int64(-1))          # K3's core on K6's stream
                    assert (st2, used2, out2) == (0, len(z), data), (case, size)
            # the same bytes as the LZMA2 chunks of one .xz block
            emu.emul_set_far_depth(8)
            nch = (size + 65535) // 65536
            stride = 65536 + 8192 + 1024
            outb = np.zeros(nch * stride, dtype=np.uint8)
            lens = (C.c_uint32 * nch)()
            st = emu.emul_lzma2_chunks_encode(C.cast(a.ctypes.data, _u8p), size, 4, C.cast(outb.ctypes.data, _u8p), stride, lens)
            assert st in (0, -200), st
            body = b""
            for i in range(nch):
                piece = data[i * 65536:(i + 1) * 65536]
                zc = outb[i * stride:i * stride + lens[i]].tobytes()
                us, cs = len(piece), len(zc)
                if cs >= us or cs > 65536 or (st != 0 and cs + 16 >= stride):
                    body += bytes([1 if i == 0 else 2, (us - 1) >> 8, (us - 1) & 255]) + piece
                else:
                    body += bytes([(0xE0 if i == 0 else 0xC0) | ((us - 1) >> 16), ((us - 1) >> 8) & 255, (us - 1) & 255, (cs - 1) >> 8, (cs - 1) & 255, 0x5D]) + zc
            body += b"\x00"
            filt = [{"id": pylzma.FILTER_LZMA2, "dict_size": 8 << 20}]
            assert pylzma.decompress(body, format=pylzma.FORMAT_RAW, filters=filt) == data, (case, size)
    finally:
        emu.emul_set_far_depth(0)


def test_lzma_encode_roundtrip(emu):
    """LZMA encode parity = valid streams that the reference side decodes back to the input: ZIP method-14 payloads
    through the oracle restatement, liblzma (Python's lzma) and -- where built -- the compiled reference; LZMA2 chunk
    payloads wrapped into an .xz stream exactly like mzhip_xz_encode_host lays it out."""
    import lzma as pylzma

    c = synth.corpus()
    rnd = np.random.RandomState(3)
    cases = [b"", b"a", b"ab", b"a" * 20, c[:100], c[:5000], c[:65536], c[:200000], rnd.bytes(3000), bytes(100000),
             c[:70000] + rnd.bytes(500) + c[:70000], b"abcabcabc" * 1000, c[:65535], c[:65537]]
    for d in cases:
        st, z, crc = _lzma_encode(emu, d)
        assert st == 0 and crc == zlib.crc32(d), len(d)
        # the header's dictionary: 64 KiB for a stream of one block, 8 MiB (the reach of the chain pass's links) beyond
        assert z[:9] == bytes([9, 20, 5, 0, 0x5D]) + (bytes([0, 0, 1, 0]) if len(d) <= 65536 else bytes([0, 0, 0x80, 0]))
        so, used, out = oracle.lzma_zip_decode(z, len(d) + 64, -1)
        assert (so, used, out) == (0, len(z), d), len(d)
        assert pylzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=pylzma.FORMAT_ALONE) == d
        if oracle.have_ref():
            r = oracle.ref().stream_decode(14, z, len(d) + 64, max_in=len(z), max_out=len(d))
            assert r["out"] == d and r["total_in"] == len(z) and r["close"] == 0, len(d)
        st2, used2, out2, crc2 = _run(emu.emul_lzma, z, len(d) + 64, C.c_int64(-1))      # and back through K3's core
        assert (st2, used2, out2, crc2) == (0, len(z), d, crc)
    assert len(_lzma_encode(emu, c[:65536])[1]) < 0.4 * 65536
    # the default class (presets 4-9 and -1: four hash candidates + two-position lazy rule): valid streams, and smaller
    emu.emul_lzma_encode_ways.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    fast = best = 0
    for d in cases:
        a = np.frombuffer(d, dtype=np.uint8) if d else np.zeros(1, np.uint8)
        out = np.zeros(len(d) + len(d) // 8 + 1024, np.uint8)
        ol, crc = C.c_uint32(), C.c_uint32()
        st = emu.emul_lzma_encode_ways(C.cast(a.ctypes.data, _u8p), len(d), 0, 4, C.cast(out.ctypes.data, _u8p), len(out), C.byref(ol), C.byref(crc))
        z = out[:ol.value].tobytes()
        assert st == 0 and crc.value == zlib.crc32(d), len(d)
        assert pylzma.decompress(z[4:9] + b"\xff" * 8 + z[9:], format=pylzma.FORMAT_ALONE) == d
        assert oracle.lzma_zip_decode(z, len(d) + 64, -1) == (0, len(z), d)
        fast += len(_lzma_encode(emu, d)[1])
        best += len(z)
    assert best < 0.97 * fast
    # LZMA2 chunk payloads -> one .xz stream (single block, CRC32 check) exactly like mzhip_xz_encode_host lays it out: the
    # block is parsed as one stream (a chunk's matches reach back over the chunks before it), every 64 KiB of it is a
    # chunk with a fresh coder and model -- the first resets the dictionary (0xE0 / stored 0x01), the others keep it
    # (0xC0 / stored 0x02)
    emu.emul_lzma2_chunks_encode.argtypes = [_u8p, C.c_uint32, C.c_uint32, _u8p, C.c_uint32, C.POINTER(C.c_uint32)]
    xz_sizes = {}
    for name, d in (("text", c[:150000]), ("noise+text", rnd.bytes(60000) + c[:1000]), ("x", b"x"), ("twice", c[:100000] + rnd.bytes(70000) + c[:100000]),
                    ("markov", synth.markov_entries(1, 1 << 20, 78, synth.bench_corpus()[0])[0])):
        nch = (len(d) + 65535) // 65536
        stride = 65536 + 8192 + 1024
        a = np.frombuffer(d, dtype=np.uint8).copy()
        outb = np.zeros(nch * stride, dtype=np.uint8)
        lens = (C.c_uint32 * nch)()
        st = emu.emul_lzma2_chunks_encode(C.cast(a.ctypes.data, _u8p), len(d), 4, C.cast(outb.ctypes.data, _u8p), stride, lens)
        assert st in (0, -200), st       # (-200: a chunk of noise did not fit its room -- it is stored, as below)
        body = b""
        for i in range(nch):
            piece = d[i * 65536:(i + 1) * 65536]
            z = outb[i * stride:i * stride + lens[i]].tobytes()
            us, cs = len(piece), len(z)
            if cs >= us or cs > 65536 or (st != 0 and cs + 16 >= stride):
                body += bytes([1 if i == 0 else 2, (us - 1) >> 8, (us - 1) & 255]) + piece
            else:
                body += bytes([(0xE0 if i == 0 else 0xC0) | ((us - 1) >> 16), ((us - 1) >> 8) & 255, (us - 1) & 255, (cs - 1) >> 8, (cs - 1) & 255, 0x5D]) + z
        body += b"\x00"
        flags = b"\x00\x01"
        bh = bytes([2, 0, 0x21, 1, 0x16 if nch > 1 else 8, 0, 0, 0])
        xz_sizes[name] = (len(d), len(body))
        x = b"\xfd7zXZ\x00" + flags + zlib.crc32(flags).to_bytes(4, "little") + bh + zlib.crc32(bh).to_bytes(4, "little")
        x += body + b"\x00" * (-len(body) % 4) + zlib.crc32(d).to_bytes(4, "little")
        idx = b"\x00\x01" + synth._vli(12 + len(body) + 4) + synth._vli(len(d))
        idx += b"\x00" * (-len(idx) % 4)
        idx += zlib.crc32(idx).to_bytes(4, "little")
        tail = (len(idx) // 4 - 1).to_bytes(4, "little") + flags
        x += idx + zlib.crc32(tail).to_bytes(4, "little") + tail + b"YZ"
        assert pylzma.decompress(x) == d
        assert oracle.xz_decode(x, len(d) + 64) == (0, len(x), d)
        st2, used2, out2, crc2 = _run(emu.emul_xz, x, len(d) + 64, C.c_int64(-1))         # and back through the .xz kernel's core
        assert (st2, used2, out2) == (0, len(x), d)
    # the second copy of "twice" lies 170 000 bytes behind the first: it costs next to nothing now; a config-4 entry
    # (VERDICT r3 missing 5, for method 95 as for method 14): <= 0.30 where chunks that were streams of their own made 0.40
    assert xz_sizes["twice"][1] < 70000 + 1.25 * xz_sizes["text"][1] * 100000 / 150000
    assert xz_sizes["markov"][1] <= 0.30 * xz_sizes["markov"][0], xz_sizes["markov"]


def test_inflate_differential_fuzz(emu):
    """Random single-byte / bit corruptions and truncations of valid streams: the device core (emulated) and the
    oracle must agree on the status class, and on bytes / consumed input whenever the stream still decodes."""
    import random

    rnd = random.Random(2024)
    c = synth.corpus()
    bases = [synth.deflate_raw(c[o:o + n], level=lv) for o, n, lv in ((100, 3000, 6), (5000, 20000, 9), (70000, 9000, 1))]
    bases.append(synth.deflate_raw(c[:6000], strategy=zlib.Z_FIXED))
    bases.append(synth.stored_blocks(c[:3000], block=1000))
    n_ok = n_err = 0
    for it in range(400):
        z = bytearray(rnd.choice(bases))
        kind = rnd.randrange(4)
        if kind == 0:
            z[rnd.randrange(len(z))] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            z[rnd.randrange(len(z))] = rnd.randrange(256)
        elif kind == 2:
            del z[rnd.randrange(1, len(z)):]
        else:
            i = rnd.randrange(min(len(z), 40))          # hit the block header / code-length area
            z[i] = rnd.randrange(256)
        z = bytes(z)
        cap = 120000
        st, used, out, crc = _run(emu.emul_inflate, z, cap)
        so, uo, oo = oracle.inflate_raw(z, cap)
        assert st == so, (it, kind, st, so)
        if so == 0:
            assert (used, out) == (uo, oo) and crc == oracle.crc32(oo), it
            n_ok += 1
        else:
            n_err += 1
    assert n_ok > 20 and n_err > 100


def _build_variant(tag, flags):
    out = os.path.join(ROOT, "tests", "emul", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libemul_%s.so" % tag)
    subprocess.run(["g++", "-O1", "-g", "-Wno-unknown-pragmas", "-DMZHIP_HOST_EMUL"] + flags +
                   ["-I" + os.path.join(ROOT, "minizip-ng_amd", "csrc"), "-shared", "-fPIC",
                    os.path.join(ROOT, "tests", "emul", "emul.cpp"), "-o", so], check=True)
    L = C.CDLL(so)
    L.emul_inflate.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32] + [C.POINTER(C.c_uint32)] * 3
    L.emul_inflate_steps.argtypes = L.emul_inflate.argtypes
    return L


@pytest.fixture(scope="module")
def emu_staged():
    """Other K1 builds in the same emulation (the default build is `emu` itself): record caps and span limits so low that
    every window runs into them, chases cross several spans and the span limit halves; 4 records per lane and emit round in
    a pool that cuts every round; the code-length decode without its 64-bit front end.  (The round-2 window and the knobs of
    rounds 2 - 4 whose A/B is closed went with their code in round 5.)"""
    return [_build_variant("c_caps", ["-DMZ_REC_CAP1=16u", "-DMZ_REC_CAP2=8u", "-DMZ_CHASE_SMAX=512u"]),
            _build_variant("c_short", ["-DMZ_CHASE_SMAX=128u", "-DMZ_REC_CAP2=4u", "-DMZ_EMIT_GROUP=4u"]),
            _build_variant("c_pool", ["-DMZ_POOL_BYTES=656u", "-DMZ_EMIT_GROUP=4u"]),
            _build_variant("c_serial_cl", ["-DMZ_CL_PARALLEL=0", "-DMZ_CHASE_SMAX=1024u"])]


def test_inflate_span_and_step_paths(emu, emu_staged):
    """K1 has two decode front ends: the span path (every lane walks its own 256-bit span, then the
    walks are chained) and the step loop (64 candidate offsets of one 64-bit window), which also owns the last span of
    a stream and every error verdict.  Both must agree with the oracle on streams long enough for the span path to
    engage, including corrupted and truncated ones and tight output caps."""
    import random

    c = synth.corpus()
    rnd = random.Random(99)
    bases = []
    for lvl, strat, n in ((6, zlib.Z_DEFAULT_STRATEGY, 65536), (1, zlib.Z_DEFAULT_STRATEGY, 30000), (9, zlib.Z_FILTERED, 65536),
                          (6, zlib.Z_FIXED, 20000), (6, zlib.Z_HUFFMAN_ONLY, 9000), (6, zlib.Z_RLE, 40000)):
        o = rnd.randrange(0, len(c) - n)
        co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, strat)
        bases.append(co.compress(c[o:o + n]) + co.flush())
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    bases.append(co.compress(bytes(rnd.randrange(256) for _ in range(3000)) + c[:30000]) + co.flush())   # several blocks
    for d in (bytes(70000), b"abc" * 20000, c[:500] * 100, b"ab" * 300 + c[:2000] + b"x" * 5000):    # runs, short periods
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        bases.append(co.compress(d) + co.flush())
    n_ok = 0
    for it in range(700):
        z = bytearray(rnd.choice(bases))
        kind = it % 5
        if kind == 0:
            z[rnd.randrange(len(z))] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            del z[rnd.randrange(1, len(z)):]
        elif kind == 2:
            z[rnd.randrange(len(z))] = rnd.randrange(256)
        z = bytes(z)
        cap = rnd.choice((120000, 120000, 5000, 66000))
        so, uo, oo = oracle.inflate_raw(z, cap)
        fns = [emu.emul_inflate, emu.emul_inflate_steps]
        for v in emu_staged:
            fns += [v.emul_inflate, v.emul_inflate_steps]
        for k, fn in enumerate(fns):
            st, used, out, crc = _run(fn, z, cap, mis=it % 4, omis=(it // 4) % 4)
            assert st == so, (it, kind, st, so, k)
            if so == 0:
                assert (used, out) == (uo, oo) and crc == oracle.crc32(oo), (it, kind)
        n_ok += so == 0
    assert n_ok > 150


def test_inflate_block_header_fuzz(emu):
    """Dynamic block headers are decoded 64 bits of code lengths at a time, with the one-symbol-at-a-time loop behind
    it for everything doubtful.  Bit flips, byte smashes and cuts confined to the header bytes of real streams (repeat
    codes with nothing before them, runs over nlen + ndist, unused code-length codes, over-subscribed sets, input that
    ends inside the lengths): verdicts, bytes and consumed counts equal the oracle's."""
    import random

    c = synth.corpus()
    rnd = random.Random(2024)
    bases = []
    for lvl, n in ((6, 20000), (9, 3000), (1, 60000), (6, 700), (6, 120)):
        for _ in range(3):
            o = rnd.randrange(0, len(c) - n)
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15)
            bases.append(co.compress(c[o:o + n]) + co.flush())
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    bases.append(co.compress(bytes(range(256)) * 40 + c[:5000]) + co.flush())      # every literal in use: long headers
    n_ok = n_err = 0
    for it in range(1500):
        z = bytearray(rnd.choice(bases))
        span = min(len(z), 130)
        kind = it % 4
        if kind == 0:
            z[rnd.randrange(span)] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            z[rnd.randrange(span)] = rnd.randrange(256)
        elif kind == 2:
            del z[rnd.randrange(1, span):]
        else:
            for _ in range(3):
                z[rnd.randrange(span)] ^= 1 << rnd.randrange(8)
        z = bytes(z)
        so, uo, oo = oracle.inflate_raw(z, 70000)
        st, used, out, crc = _run(emu.emul_inflate, z, 70000, mis=it % 4)
        assert st == so, (it, kind, st, so)
        if so == 0:
            assert (used, out) == (uo, oo) and crc == oracle.crc32(oo), it
            n_ok += 1
        else:
            n_err += 1
    assert n_ok > 20 and n_err > 300


def test_inflate_moving_view():
    """The decoder addresses the stream through a view that moves forward (32-bit bit cursor, streams of any length):
    built here with a 256 KiB view that moves every 32 KiB, so that streams of a few hundred KiB cross many moves -- in
    stored blocks, in Huffman blocks on the span path and on the step loop -- and end (or are cut) at every distance
    from one.  Same verdicts, bytes, consumed counts and CRCs as the oracle."""
    import random

    v = _build_variant("view", ["-DMZ_VIEW_MAX=(1u<<18)", "-DMZ_REBASE_BITS=(1u<<18)"])
    c = synth.corpus()
    rnd = random.Random(7)
    noise = bytes(rnd.randrange(256) for _ in range(400000))
    mixed = bytearray()
    while len(mixed) < 1200000:                                  # text with incompressible islands: ratio ~0.6
        o = rnd.randrange(0, len(c) - 3000)
        mixed += c[o:o + rnd.randrange(200, 3000)]
        o = rnd.randrange(0, len(noise) - 2000)
        mixed += noise[o:o + rnd.randrange(100, 2000)]
    mixed = bytes(mixed)
    streams = []
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    streams.append((co.compress(mixed) + co.flush(), mixed))
    co = zlib.compressobj(0, zlib.DEFLATED, -15)                 # stored blocks only
    streams.append((co.compress(noise) + co.flush(), noise))
    co = zlib.compressobj(1, zlib.DEFLATED, -15)                 # stored blocks, then text behind several moves
    both = noise[:300000] + c[:200000]
    streams.append((co.compress(both) + co.flush(), both))
    for z, d in streams:
        assert len(z) > (1 << 18) + 70000
        for fn in (v.emul_inflate, v.emul_inflate_steps):
            st, used, out, crc = _run(fn, z, len(d) + 8, mis=1, omis=3)
            assert (st, used, out, crc) == (0, len(z), d, zlib.crc32(d))
    z, d = streams[0]
    for it in range(12):
        cut = rnd.randrange(1 << 15, len(z))
        zz = bytearray(z[:cut]) if it % 2 == 0 else bytearray(z)
        if it % 2:
            zz[cut] ^= 1 << rnd.randrange(8)
        zz = bytes(zz)
        so, uo, oo = oracle.inflate_raw(zz, len(d) + 8)
        st, used, out, crc = _run(v.emul_inflate, zz, len(d) + 8, mis=it % 4)
        assert st == so, (it, st, so)
        if so == 0:
            assert (used, out) == (uo, oo)


def test_deflate_default_class_roundtrip(emu):
    """K4's classes above the fast one in the emulation -- "lazy" (levels 4-6, -1: four candidates per hash bucket, matches
    handed on to the next positions, two-position lazy rule) and "best" (levels 7-9: the cost parse on top): zlib inflates
    the bytes back, every class is smaller than the one below, and the cost parse lands within 4 % of zlib-9."""
    emu.emul_deflate_best.argtypes = emu.emul_deflate.argtypes
    emu.emul_deflate_lazy.argtypes = emu.emul_deflate.argtypes
    text, _ = synth.bench_corpus()
    rnd = np.random.RandomState(12)
    datas = [text[o:o + 65536] for o in rnd.randint(0, len(text) - 65536, size=6)]
    datas += [b"", b"a", b"abc", b"abcd", b"abcabcabcabc" * 300, bytes(70000), text[:200000], rnd.bytes(4000), b"ab" * 40000,
              text[:65], text[:259], bytes(rnd.randint(0, 2, size=5000, dtype=np.uint8))]
    tot = {"fast": 0, "lazy": 0, "best": 0, "zlib9": 0}
    for d in datas:
        a = np.frombuffer(d, dtype=np.uint8).copy() if d else np.zeros(1, np.uint8)
        for name, fn in (("fast", emu.emul_deflate), ("lazy", emu.emul_deflate_lazy), ("best", emu.emul_deflate_best)):
            for final in (1, 0):
                out = np.zeros(len(d) + len(d) // 8 + 1000, np.uint8)
                ol, crc = C.c_uint32(), C.c_uint32()
                st = fn(a.ctypes.data_as(_u8p), len(d), out.ctypes.data_as(_u8p), len(out), final, C.byref(ol), C.byref(crc))
                z = out[:ol.value].tobytes()
                back = zlib.decompress(z, -15) if final else zlib.decompressobj(-15).decompress(z)
                assert st == 0 and back == d and crc.value == zlib.crc32(d), (name, final, len(d))
                if len(d) == 65536 and final:
                    tot[name] += ol.value
        if len(d) == 65536:
            tot["zlib9"] += len(zlib.compress(d, 9)) - 6
    assert tot["lazy"] < 0.92 * tot["fast"] and tot["lazy"] <= 0.32 * 6 * 65536, tot
    assert tot["best"] < 0.985 * tot["lazy"] and tot["best"] <= 1.04 * tot["zlib9"], tot


def test_deflate_classes_fuzz(emu):
    """Round trips of the lazy class and the cost parse (deflate_core.h, deflate_select.inc) over the shapes that break
    parsers: every length around the step (64), the group (4) and the match limits (3, 4, 258, 259), runs, short periods,
    two-symbol noise, incompressible bytes, blocks beyond 64 KiB, final and non-final pieces, small windows."""
    import random

    emu.emul_deflate_best.argtypes = emu.emul_deflate.argtypes
    emu.emul_deflate_lazy.argtypes = emu.emul_deflate.argtypes
    c = synth.corpus()
    rnd = random.Random(20)
    sizes = [0, 1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 66, 127, 128, 129, 255, 256, 257, 258, 259, 260, 511, 512, 513, 4095, 4096, 4097,
             16383, 16384, 16385, 65535, 65536, 65537, 70001]
    for it in range(72):
        n = sizes[it % len(sizes)] if it < 2 * len(sizes) else rnd.randrange(1, 140000)
        k = it % 6
        if k == 0: d = c[rnd.randrange(len(c) - n):][:n]
        elif k == 1: d = bytes(rnd.randrange(256) for _ in range(min(n, 6000)))
        elif k == 2: d = bytes([rnd.randrange(3)]) * n
        elif k == 3: d = (c[rnd.randrange(1000):][:rnd.randrange(1, 300)] * 2000)[:n]
        elif k == 4: d = bytes(rnd.choice(b"ab") for _ in range(min(n, 20000)))
        else: d = (bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 40))) * 70000)[:n]
        emu.emul_deflate_window(rnd.choice([9, 12, 15, 15]))
        a = np.frombuffer(d, dtype=np.uint8).copy() if d else np.zeros(1, np.uint8)
        for name, fn in (("best", emu.emul_deflate_best), ("lazy", emu.emul_deflate_lazy)):
            final = (it + (name == "lazy")) & 1
            out = np.zeros(len(d) + len(d) // 8 + 1024, np.uint8)
            ol, crc = C.c_uint32(), C.c_uint32()
            st = fn(a.ctypes.data_as(_u8p), len(d), out.ctypes.data_as(_u8p), len(out), final, C.byref(ol), C.byref(crc))
            z = out[:ol.value].tobytes()
            back = zlib.decompress(z, -15) if final else zlib.decompressobj(-15).decompress(z)
            assert st == 0 and back == d and crc.value == zlib.crc32(d), (it, name, final, len(d))
    emu.emul_deflate_window(15)


def test_parallel_window_candidate_order(tmp_path):
    """mz_parallel_window (inflate_parallel.inc) puts the block-header candidates of a window in ascending order on the host
    with counting passes instead of std::sort (5 of a 64 MiB window's 13 ms went there): tests/unit_par_order.cpp holds it
    against std::sort on 300 vectors of 0 - 300 000 keys of every width."""
    exe = str(tmp_path / "unit_par_order")
    subprocess.run(["g++", "-O2", os.path.join(ROOT, "tests", "unit_par_order.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "0 of 300 differ" in r.stdout, r.stdout + r.stderr


def test_deflate_pieces_with_history(emu):
    """K4 cuts ONE stream into pieces of 16 KiB for as many waves (the WRITE shim's segments, mzhip_prime_write's entries); every
    piece but the first is handed the 32 KiB in front of it (`warm`): they are hashed into the buckets before its first position
    is coded and its matches may reach back into them.  The pieces' bytes, concatenated, are one stream zlib inflates to the
    input; each piece's CRC covers its own bytes only; and the stream is SMALLER than with pieces of 64 KiB that do not see each
    other (what rounds 2 - 5 wrote), far smaller than 16 KiB pieces without the history."""
    import random
    import zlib

    from tests import synth

    emu.emul_deflate_lazy.argtypes = emu.emul_deflate.argtypes
    emu.emul_deflate_best.argtypes = emu.emul_deflate.argtypes
    text, _ = synth.bench_corpus()
    rnd = random.Random(3)

    def encode(data, piece, fn, history):
        a = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(1, dtype=np.uint8)
        out, pos, n = b"", 0, len(data)
        while True:
            k = min(piece, n - pos)
            warm = (min(pos, 32768) // 64) * 64 if history else 0
            cap = k + k // 8 + 128
            o = np.zeros(cap, dtype=np.uint8)
            ol, cr = C.c_uint32(), C.c_uint32()
            emu.emul_deflate_warm(warm)
            try:
                st = fn(a[pos - warm:].ctypes.data_as(_u8p), k + warm, o.ctypes.data_as(_u8p), cap, 1 if pos + k >= n else 0, C.byref(ol), C.byref(cr))
            finally:
                emu.emul_deflate_warm(0)
            assert st == 0 and cr.value == zlib.crc32(data[pos:pos + k]), (pos, k, st)
            out += o[:ol.value].tobytes()
            pos += k
            if pos >= n:
                return out

    cases = [("text", (text * 2)[:400000]), ("binary + text", bytes((i * 7 + (i >> 3)) & 255 for i in range(150000)) + text[:100000]),
             ("one piece and a bit", text[:16384 + 70]), ("exactly four", text[:65536]), ("short", text[:100]), ("empty", b""),
             ("noise", bytes(rnd.getrandbits(8) for _ in range(70000))), ("zeros", bytes(200000))]
    for name, data in cases:
        for cls, fn in (("fast", emu.emul_deflate), ("lazy", emu.emul_deflate_lazy), ("best", emu.emul_deflate_best)):
            if cls == "best" and len(data) > 200000:
                continue
            z16 = encode(data, 16384, fn, True)
            assert zlib.decompress(z16, -15) == data, (name, cls)
            if name == "text" and cls != "best":
                z64_blind, z16_blind = encode(data, 65536, fn, False), encode(data, 16384, fn, False)
                assert zlib.decompress(z64_blind, -15) == data
                print("%s, %s: 16 KiB pieces with history %.4f, 64 KiB pieces without %.4f, 16 KiB without %.4f" % (
                    name, cls, len(z16) / len(data), len(z64_blind) / len(data), len(z16_blind) / len(data)))
                assert len(z16) < len(z64_blind) < len(z16_blind)
