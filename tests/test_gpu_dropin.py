"""GPU tests of the DROP-IN boundary: the reference's own mz_zip.c / mz_zip_rw.c / mz_strm*.c, compiled
unmodified and linked against libmzhip.so instead of mz_strm_zlib.o / mz_strm_lzma.o / the CRC symbol
(integration/_build/libmzhipdrop.so), are driven through the same integration/mz_driver.c entry points as the
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


ALL = ("rets", "out", "total_in", "total_out", "close", "error", "base_pos", "open")


def _appendix_b_deflate():
    text, _ = synth.bench_corpus()
    d = text[:65536]
    return d, synth.deflate_raw(d)


def _check_truncations_deflate(hip, ref, chunks=(65535,)):
    """what TOTAL_IN / TOTAL_OUT, the read() sequence, the bytes, close(), error() and the base
    stream's position are after the input ran out -- every field against the all-reference build, the two rows of the
    appendix as literals (the entry is the appendix's own: the first 65 536 bytes of appnote.txt, 16 778 bytes at level 6)."""
    d, z = _appendix_b_deflate()
    for chunk in chunks:
        for how in ("eof", "max_in"):
            kw = dict(chunk=chunk)
            cut = len(z) // 2
            a = hip.stream_decode(8, z[:cut] if how == "eof" else z, len(d) + 64, max_in=cut if how == "max_in" else 0, **kw)
            b = ref.stream_decode(8, z[:cut] if how == "eof" else z, len(d) + 64, max_in=cut if how == "max_in" else 0, **kw)
            assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (how, chunk, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
            if len(z) == 16778 and chunk == 65535:  # the reference corpus travelled: the appendix's numbers themselves
                assert (a["rets"], a["close"], a["error"], a["total_in"], a["total_out"], a["base_pos"]) == ([-5], -112, -5, 8389, 28634, 8389)
        for cut in (0, 1, 2, 5, 100, len(z) // 4, len(z) // 2 + 1, len(z) - 300, len(z) - 3, len(z) - 1):
            a = hip.stream_decode(8, z[:cut], len(d) + 64, chunk=chunk)
            b = ref.stream_decode(8, z[:cut], len(d) + 64, chunk=chunk)
            assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (cut, chunk, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})


def test_truncation_accounting_deflate(libs):
    hip, ref = libs
    _check_truncations_deflate(hip, ref, chunks=(65535, 16384, 1000))


def test_code_length_code_without_a_code(libs):
    """A dynamic block whose code-length code has no code at all (every HCLEN field 0).  zlib does not refuse it where it
    stands: inflate_table() answers "no symbols" with a table of invalid one-bit entries, inflate()'s CODELENS state reads the
    nlen + ndist lengths through it -- a 0 each, one bit each -- and then refuses the block for having no end-of-block code
    (inflate.c: "invalid code -- missing end-of-block").  Same verdict, nlen + ndist bits later: every field, TOTAL_IN included,
    as the all-reference build -- with the input ending inside those bits too (found by tests/fuzz_gpu_windows.py 1500 94)."""
    hip, ref = libs
    for hlit, hdist, hclen in ((0, 0, 0), (29, 29, 15), (7, 3, 2)):
        # BFINAL = 1, BTYPE = 2, HLIT, HDIST, HCLEN, then (hclen + 4) x 3 zero bits and zeros on
        hdr = 1 | (2 << 1) | (hlit << 3) | (hdist << 8) | (hclen << 13)
        bits = 17 + 3 * (hclen + 4) + (hlit + 257) + (hdist + 1)
        z = hdr.to_bytes(3, "little") + bytes(80)
        for n in (3, (17 + 3 * (hclen + 4)) // 8 + 1, bits // 8, (bits + 7) // 8, (bits + 7) // 8 + 1, len(z)):
            for chunk in (65535, 7):
                a = hip.stream_decode(8, z[:n], 4096, chunk=chunk)
                b = ref.stream_decode(8, z[:n], 4096, chunk=chunk)
                assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (hlit, hdist, hclen, n, chunk, {k: (a[k], b[k]) for k in ALL if a[k] != b[k]})
        assert b["error"] == -3 and b["total_in"] == (bits + 7) // 8, (b["error"], b["total_in"], bits)


def _crafted_deflate_corners():
    """Hand-made raw-DEFLATE streams around the corners of the format that no compressor writes (a small bit writer): a block
    of nothing but its end-of-block code (a one-bit code: the incomplete set zlib accepts) and the unused pattern of that code;
    literals only with NO distance code; a single one-bit distance code, used, and its unused pattern; a match where there is
    no distance code; a distance in front of the stream; the fixed code's length symbol 286 and distance code 30; empty stored
    blocks, a stored block whose lengths disagree, one cut inside its data; block type 3; 287 literal/length codes; a
    code-length code that is over-subscribed, incomplete, a single one-bit code.  -> [(name, stream)]"""
    import math

    class Bits:
        def __init__(self):
            self.bits = []

        def put(self, value, n):  # a field, least significant bit first
            self.bits += [(value >> i) & 1 for i in range(n)]

        def code(self, code_len):  # a Huffman code, most significant bit first
            code, n = code_len
            self.bits += [(code >> i) & 1 for i in range(n - 1, -1, -1)]

        def bytes(self, pad=0):
            b = self.bits + [0] * (-len(self.bits) % 8)
            return bytes(sum(b[i + j] << j for j in range(8)) for i in range(0, len(b), 8)) + bytes(pad)

    def canonical(lens):  # -> {symbol: (code, length)}, appnote.txt:2139-2166
        top = max(lens) if lens else 0
        count = [0] * (top + 2)
        for n in lens:
            if n:
                count[n] += 1
        code, nxt = 0, [0] * (top + 2)
        for n in range(1, top + 1):
            code = (code + count[n - 1]) << 1
            nxt[n] = code
        out = {}
        for sym, n in enumerate(lens):
            if n:
                out[sym] = (nxt[n], n)
                nxt[n] += 1
        return out

    order = (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15)

    def dynamic_header(bw, lit, dist):
        """BFINAL = 1, BTYPE = 2, the code-length code (a complete code over the lengths that occur; a second one-bit code
        beside a single one) and the lengths themselves, one symbol each (no repeats) -> the two codes"""
        seq = list(lit) + list(dist)
        used = sorted(set(seq))
        clens = [0] * 19
        if len(used) == 1:
            clens[used[0]] = clens[(used[0] + 1) % 16] = 1
        else:
            n = math.ceil(math.log2(len(used)))
            short = (1 << n) - len(used)
            for i, sym in enumerate(used):
                clens[sym] = n - 1 if i < short else n
        cc = canonical(clens)
        bw.put(1, 1)
        bw.put(2, 2)
        bw.put(len(lit) - 257, 5)
        bw.put(len(dist) - 1, 5)
        hclen = 19
        while hclen > 4 and clens[order[hclen - 1]] == 0:
            hclen -= 1
        bw.put(hclen - 4, 4)
        for i in range(hclen):
            bw.put(clens[order[i]], 3)
        for n in seq:
            bw.code(cc[n])
        return canonical(lit), canonical(dist)

    def fixed_lit(bw, sym):  # appnote.txt:2054-2059
        if sym < 144:
            bw.code((0x30 + sym, 8))
        elif sym < 256:
            bw.code((0x190 + sym - 144, 9))
        elif sym < 280:
            bw.code((sym - 256, 7))
        else:
            bw.code((0xC0 + sym - 280, 8))

    cases = []
    only_eob = [0] * 257
    only_eob[256] = 1
    bw = Bits()
    lc, _ = dynamic_header(bw, only_eob, [0])
    bw.code(lc[256])
    cases.append(("only the end-of-block code, one bit", bw.bytes(4)))
    bw = Bits()
    dynamic_header(bw, only_eob, [0])
    bw.code((1, 1))
    cases.append(("the unused pattern of a one-bit literal/length code", bw.bytes(4)))

    abc = [0] * 257
    abc[65] = abc[66] = abc[67] = abc[256] = 2
    bw = Bits()
    lc, _ = dynamic_header(bw, abc, [0])
    for ch in b"ABCABCCBA":
        bw.code(lc[ch])
    bw.code(lc[256])
    cases.append(("literals only, no distance code", bw.bytes(4)))

    ab = [0] * 258
    ab[65] = ab[66] = ab[256] = ab[257] = 2  # 257 = a match of 3 bytes
    for dist_bit, name in ((0, "a single one-bit distance code, used"), (1, "the unused pattern of a one-bit distance code")):
        bw = Bits()
        lc, _ = dynamic_header(bw, ab, [1])
        for ch in b"ABAB":
            bw.code(lc[ch])
        bw.code(lc[257])
        bw.code((dist_bit, 1))
        bw.code(lc[256])
        cases.append((name, bw.bytes(4)))
    bw = Bits()
    lc, _ = dynamic_header(bw, ab, [0])
    for ch in b"ABAB":
        bw.code(lc[ch])
    bw.code(lc[257])
    bw.put(0, 1)
    bw.code(lc[256])
    cases.append(("a match where there is no distance code", bw.bytes(4)))
    dist = [0] * 30
    dist[0] = dist[4] = 1
    bw = Bits()
    lc, dc = dynamic_header(bw, ab, dist)
    for ch in b"AB":
        bw.code(lc[ch])
    bw.code(lc[257])
    bw.code(dc[4])
    bw.put(1, 1)  # distance code 4: 5 + one extra bit = 6, two bytes into the stream
    bw.code(lc[256])
    cases.append(("a distance in front of the stream", bw.bytes(4)))

    for sym, name in ((None, "fixed block, distance code 30"), (286, "fixed block, length symbol 286")):
        bw = Bits()
        bw.put(1, 1)
        bw.put(1, 2)
        for ch in b"ABCD":
            fixed_lit(bw, ch)
        if sym is None:
            fixed_lit(bw, 257)
            bw.code((30, 5))
        else:
            fixed_lit(bw, sym)
        fixed_lit(bw, 256)
        cases.append((name, bw.bytes(4)))

    cases.append(("an empty stored block, then an empty final one", b"\x00\x00\x00\xff\xff" + b"\x01\x00\x00\xff\xff"))
    cases.append(("stored block, LEN and NLEN disagree", b"\x01\x05\x00\xfa\xfe" + b"hello"))
    cases.append(("stored block, cut inside its data", b"\x01\x05\x00\xfa\xff" + b"hel"))
    cases.append(("block type 3", b"\x07\x00\x00\x00\x00"))
    bw = Bits()
    bw.put(1, 1)
    bw.put(2, 2)
    bw.put(30, 5)
    bw.put(0, 5)
    bw.put(0, 4)
    cases.append(("287 literal/length codes", bw.bytes(40)))
    for clens, name in (([1] * 19, "code-length code over-subscribed"), ([2, 0, 0, 0], "code-length code incomplete"),
                        ([0, 0, 0, 1], "code-length code of a single one-bit code")):
        bw = Bits()
        bw.put(1, 1)
        bw.put(2, 2)
        bw.put(0, 5)
        bw.put(0, 5)
        bw.put(len(clens) - 4, 4)
        for n in clens:
            bw.put(n, 3)
        cases.append((name, bw.bytes(60)))
    return cases


def test_crafted_deflate_corners(libs):
    """Every crafted corner stream, whole and cut 1 .. 11 bytes short, in 65 535-byte and 3-byte read() calls: every read() return
    value, byte, TOTAL_IN / TOTAL_OUT, close() and error() as the all-reference build (17 streams x up to 12 lengths x 2 call sizes)."""
    hip, ref = libs
    n = 0
    for name, z in _crafted_deflate_corners():
        for cut in (0,) + tuple(range(1, min(len(z), 12))):
            data = z[:len(z) - cut]
            for chunk in (65535, 3):
                a = hip.stream_decode(8, data, 70000, chunk=chunk, window_bits=-15)
                b = ref.stream_decode(8, data, 70000, chunk=chunk, window_bits=-15)
                assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (name, cut, chunk, {k: (a[k], b[k]) for k in ALL if a[k] != b[k]})
                n += 1
    assert n > 300


def test_refusal_behind_a_full_buffer(libs):
    """40 000 bytes of valid output (more than any distance reaches back), then a block type inflate() refuses: read() calls whose
    sizes divide the 40 000 exactly.  The call that would have returned the last of those bytes has a full buffer and inflate()
    still walks on -- a refusal that needs no room for output is met right there: that call fails and its bytes are lost.  In
    front of 32 KiB of output the refusal may be a distance too far back, which waits for room; the device's verdict does not
    say, so the shim asks it once more behind 32 KiB of make-believe history (shim_zlib.c refusal_is_not_a_distance): both kinds
    are among the cases.  One buffer and in windows; raw, zlib and gzip framing."""
    import ctypes as C

    hip, ref = libs
    text, _ = synth.bench_corpus()
    d = text[5000:45000]
    L = hip.L
    L.mzhip_set_stream_window.argtypes = [C.c_int64, C.c_int64]
    L.mzhip_set_stream_window.restype = None
    for win in (0, 192 << 10, 16384 + 32768):
        L.mzhip_set_stream_window(win, 48 << 10 if win else 0)
        try:
            for wb in (-15, 15, 31):
                co = zlib.compressobj(6, zlib.DEFLATED, wb)
                z = co.compress(d) + co.flush(zlib.Z_FULL_FLUSH) + b"\x07\x00\x00\x00"  # BFINAL = 1, BTYPE = 3
                for chunk in (40000, 20000, 8000, 5000, 7000, 65535):
                    a = hip.stream_decode(8, z, len(d) + 70000, chunk=chunk, window_bits=wb)
                    b = ref.stream_decode(8, z, len(d) + 70000, chunk=chunk, window_bits=wb)
                    assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (win, wb, chunk, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
                    assert b["rets"][-1] == -3 and (len(b["out"]) < len(d) if len(d) % chunk == 0 or chunk > len(d) else True), (wb, chunk, b["rets"])
            # in front of 32 KiB of output the refusal may be a distance too far back -- the one kind that waits for room: 14 bytes,
            # then block type 3 (met in the call that fills the buffer) against two literals, then a distance of 6 (met one call later)
            early = [b"\x00\x0e\x00\xf1\xffABCDEFGHIJKLMN" + b"\x07\x00\x00\x00"]
            early += [z for name, z in _crafted_deflate_corners() if name in ("a distance in front of the stream", "fixed block, length symbol 286",
                                                                               "the unused pattern of a one-bit distance code")]
            assert len(early) == 4
            for z in early:
                for chunk in (1, 2, 4, 7, 14):
                    a = hip.stream_decode(8, z, 70000, chunk=chunk, window_bits=-15)
                    b = ref.stream_decode(8, z, 70000, chunk=chunk, window_bits=-15)
                    assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (win, z[:8], chunk, {k: (a[k], b[k]) for k in ALL if a[k] != b[k]})
        finally:
            L.mzhip_set_stream_window(0, 0)


def test_bitflip_verdicts_deflate(libs):
    """One flipped bit anywhere: the same read() sequence, bytes, close() and error().  TOTAL_IN / TOTAL_OUT after a data
    error are zlib's internal detection point ("treat as best-effort"): compared, counted, not asserted."""
    import random

    hip, ref = libs
    d, z = _appendix_b_deflate()
    rnd = random.Random(11)
    same_totals = n = 0
    for k in range(24):
        i = len(z) // 3 if k == 0 else rnd.randrange(len(z))
        bad = z[:i] + bytes([z[i] ^ (0x55 if k == 0 else 1 << rnd.randrange(8))]) + z[i + 1:]
        for chunk in (65535, 4096):
            a = hip.stream_decode(8, bad, len(d) + 70000, chunk=chunk)
            b = ref.stream_decode(8, bad, len(d) + 70000, chunk=chunk)
            assert (a["rets"], a["out"], a["close"], a["error"]) == (b["rets"], b["out"], b["close"], b["error"]), (i, chunk, a["rets"], b["rets"])
            n += 1
            same_totals += (a["total_in"], a["total_out"]) == (b["total_in"], b["total_out"])
    print("bit flips: %d of %d cases also agree in TOTAL_IN / TOTAL_OUT (best-effort fields)" % (same_totals, n))


def test_truncation_accounting_window_mode(libs):
    """The same truncations with the READ stream in window mode (a 192 KiB window, 48 KiB gulps: mzhip_set_stream_window),
    on the device: entries of a few hundred KiB cross several windows, whole, cut at a dozen places, bit-flipped."""
    import ctypes as C

    hip, ref = libs
    L = hip.L  # (libmzhipdrop.so resolves the symbol in libmzhip.so, which it is linked against)
    L.mzhip_set_stream_window.argtypes = [C.c_int64, C.c_int64]
    L.mzhip_set_stream_window.restype = None
    L.mzhip_set_stream_window(192 << 10, 48 << 10)
    try:
        text, _ = synth.bench_corpus()
        for d in (text[:450000], text[1000:150000] + bytes(250000) + text[:90000]):
            for lvl in (6, 0):
                z = synth.deflate_raw(d, lvl)
                for chunk in (65535, 300000):
                    a = hip.stream_decode(8, z, len(d) + 10, chunk=chunk)
                    b = ref.stream_decode(8, z, len(d) + 10, chunk=chunk)
                    assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (lvl, chunk)
                for cut in (len(z) // 5, len(z) // 2, len(z) * 4 // 5, len(z) - 3, len(z) - 1):
                    for how in ("eof", "max_in"):
                        a = hip.stream_decode(8, z[:cut] if how == "eof" else z, len(d) + 10, max_in=cut if how == "max_in" else 0)
                        b = ref.stream_decode(8, z[:cut] if how == "eof" else z, len(d) + 10, max_in=cut if how == "max_in" else 0)
                        assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (lvl, cut, how, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
                zz = bytearray(z)
                zz[len(zz) * 2 // 3] ^= 0x10
                a = hip.stream_decode(8, bytes(zz), len(d) + 10)
                b = ref.stream_decode(8, bytes(zz), len(d) + 10)
                assert (a["rets"], a["out"], a["close"], a["error"]) == (b["rets"], b["out"], b["close"], b["error"]), (lvl, "flip")
        _check_truncations_deflate(hip, ref)  # (the appendix entry is smaller than a window: the one-buffer path, knob set)
        # delete() without close() after reads that went into window mode (the piece tables were freed twice)
        d = text[:450000] * 2
        z = synth.deflate_raw(d, 6)
        got = hip.stream_delete_unclosed(8, z, len(d), chunk=65535, nreads=6)
        assert got == d[:6 * 65535]
    finally:
        L.mzhip_set_stream_window(0, 0)


def test_window_mode_many_waves(libs):
    """Window mode with a wave per DEFLATE block (mzhip_inflate_parallel_host; shim_zlib.c stream_next offers it every
    window that starts at a block header): 4 MiB windows and 1 MiB gulps over streams of 8 - 25 MB made of dynamic, fixed and
    stored blocks -- whole, in small and large read() calls, cut, bit-flipped: every read() return value, byte, TOTAL_IN /
    TOTAL_OUT, close() and error() as the all-reference build; the same streams with the many-wave decode switched off."""
    import ctypes as C
    import random
    import zlib

    hip, ref = libs
    L = hip.L
    L.mzhip_set_stream_window.argtypes = [C.c_int64, C.c_int64]
    L.mzhip_set_stream_window.restype = None
    L.mzhip_set_stream_window(4 << 20, 1 << 20)
    text, _ = synth.bench_corpus()
    rnd = random.Random(12)
    noise = bytes(rnd.getrandbits(8) for _ in range(300000))

    def blocks_of(parts):
        out = b""
        for i, (data, lvl, strat) in enumerate(parts):
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strat)
            out += co.compress(data) + (co.flush(zlib.Z_FULL_FLUSH) if i + 1 < len(parts) else co.flush())
        return out

    big = (text + text[::-1][:200000]) * 36 + bytes(3000000) + text[200000:400000] * 5
    streams = [("dynamic 6", synth.deflate_raw(big, 6)), ("level 1", synth.deflate_raw(big[:12000000], 1)),
               ("mixed", blocks_of([(text * 6, 6, 0), (text, 6, zlib.Z_FIXED), (noise, 0, 0), (text * 5, 9, 0), (noise[:70000], 6, 0),
                                    (text * 2, 6, zlib.Z_FIXED), (text * 6, 6, 0)])),
               ("fixed only", blocks_of([(text * 2, 6, zlib.Z_FIXED)] * 3))]
    try:
        for name, z in streams:
            d = zlib.decompress(z, -15)
            for chunk in (65535, 3000000):
                a = hip.stream_decode(8, z, len(d) + 10, chunk=chunk)
                b = ref.stream_decode(8, z, len(d) + 10, chunk=chunk)
                assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (name, chunk, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
            for cut in (len(z) // 2, len(z) - 3):
                a = hip.stream_decode(8, z[:cut], len(d) + 10)
                b = ref.stream_decode(8, z[:cut], len(d) + 10)
                assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (name, "cut", cut, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
            for where in (len(z) // 3, len(z) * 2 // 3):
                zz = bytearray(z)
                zz[where] ^= 0x10
                cap = 2 * len(d) + (1 << 20)
                a = hip.stream_decode(8, bytes(zz), cap)
                b = ref.stream_decode(8, bytes(zz), cap)
                assert all(a[k] == b[k] for k in ALL if k != "base_pos"), (name, "flip", where, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
        L.mzhip_set_stream_parallel(0)
        name, z = streams[2]
        d = zlib.decompress(z, -15)
        a = hip.stream_decode(8, z, len(d) + 10)
        b = ref.stream_decode(8, z, len(d) + 10)
        assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, "serial windows only"
    finally:
        L.mzhip_set_stream_parallel(1)
        L.mzhip_set_stream_window(0, 0)


def test_truncation_accounting_lzma(libs):
    """SURVEY Appendix B, mz_stream_lzma READ: the first 150 000 bytes of appnote.txt written by the reference's own WRITE
    stream (34 458 bytes); cut in half the reference returns 65 535, then -3, with TOTAL_IN / TOTAL_OUT = 17 229 / 74 787."""
    hip, ref = libs
    text, _ = synth.bench_corpus()
    d = text[:150000]
    z, _ = ref.stream_encode(14, d, level=6)
    for chunk in (65535, 10000):
        for how in ("eof", "max_in"):
            cut = len(z) // 2
            a = hip.stream_decode(14, z[:cut] if how == "eof" else z, len(d) + 64, chunk=chunk, max_in=cut if how == "max_in" else 0)
            b = ref.stream_decode(14, z[:cut] if how == "eof" else z, len(d) + 64, chunk=chunk, max_in=cut if how == "max_in" else 0)
            assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (how, chunk, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
            if len(z) == 34458 and chunk == 65535:
                assert (a["rets"], a["close"], a["error"], a["total_in"], a["total_out"]) == ([65535, -3], -112, 10, 17229, 74787)
        for cut in (0, 3, 9, 12, 13, 14, 100, len(z) // 4, len(z) - 500, len(z) - 6, len(z) - 1):
            a = hip.stream_decode(14, z[:cut], len(d) + 64, chunk=chunk)
            b = ref.stream_decode(14, z[:cut], len(d) + 64, chunk=chunk)
            assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (cut, chunk, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
    # bit flips: verdicts compared, TOTAL_* best-effort (liblzma's detection point)
    import random

    rnd = random.Random(5)
    same = n = 0
    for k in range(10):
        i = len(z) // 3 if k == 0 else rnd.randrange(13, len(z))
        bad = z[:i] + bytes([z[i] ^ (0x55 if k == 0 else 1 << rnd.randrange(8))]) + z[i + 1:]
        a = hip.stream_decode(14, bad, len(d) + 70000)
        b = ref.stream_decode(14, bad, len(d) + 70000)
        assert (a["rets"], a["out"], a["close"], a["error"]) == (b["rets"], b["out"], b["close"], b["error"]), (i, a["rets"], b["rets"], a["error"], b["error"])
        n += 1
        same += (a["total_in"], a["total_out"]) == (b["total_in"], b["total_out"])
    print("lzma bit flips: %d of %d cases also agree in TOTAL_IN / TOTAL_OUT" % (same, n))


def test_lzma_refusal_behind_a_full_buffer(libs):
    """Method 14, corrupted streams read ONE byte (and seven) at a time, so that the bytes in front of the refusal always fill the
    caller's buffer: liblzma is not told the entry's size and, with the buffer full, still decodes the next packet to see whether
    it is the end marker -- a packet it refuses is refused in that call, which then fails instead of returning its byte(s).  The
    same read() sequence, bytes, close() and error() as the all-reference build, one buffer and in windows."""
    import ctypes as C
    import random

    hip, ref = libs
    L = hip.L
    L.mzhip_set_stream_window.argtypes = [C.c_int64, C.c_int64]
    L.mzhip_set_stream_window.restype = None
    text, _ = synth.bench_corpus()
    d = text[20000:23000]
    z = _zip_lzma(d, preset=3)
    rnd = random.Random(5)
    refused = 0
    for win in (0, 4096):  # (4096: the smallest window the knob takes -- every entry is decoded in windows)
        L.mzhip_set_stream_window(win, 4096 if win else 0)
        try:
            for k in range(14):
                i = 30 + rnd.randrange(len(z) - 40)
                bad = z[:i] + bytes([z[i] ^ (1 << rnd.randrange(8))]) + z[i + 1:]
                for chunk in (1, 7):
                    # (no TOTAL_OUT_MAX: a corrupted stream that runs past it is the case test_lzma_window_mode leaves out)
                    a = hip.stream_decode(14, bad, len(d) + 70000, chunk=chunk)
                    b = ref.stream_decode(14, bad, len(d) + 70000, chunk=chunk)
                    assert (a["rets"], a["out"], a["close"], a["error"]) == (b["rets"], b["out"], b["close"], b["error"]), \
                        (win, i, chunk, len(a["rets"]), len(b["rets"]), a["rets"][-2:], b["rets"][-2:], a["error"], b["error"])
                    refused += b["error"] != 0
        finally:
            L.mzhip_set_stream_window(0, 0)
    assert refused > 10, refused


def test_lzma_window_mode(libs):
    """mz_stream_lzma READ in window mode (shim_lzma.c: the resumable build of K3, a 64-byte coder state and the adaptive
    model carried from launch to launch, out[] = dictionary so far + one window): entries of many windows -- presets whose
    dictionary is smaller than the entry (the dictionary slides) and the default 8 MiB one -- whole, with and without the
    limits mz_zip sets, in small and large read() calls, cut at several places, bit-flipped: every read() return value,
    byte, TOTAL_IN / TOTAL_OUT, close() and error() as the all-reference build."""
    import ctypes as C

    hip, ref = libs
    L = hip.L
    L.mzhip_set_stream_window.argtypes = [C.c_int64, C.c_int64]
    L.mzhip_set_stream_window.restype = None
    L.mzhip_set_stream_window(192 << 10, 48 << 10)
    try:
        text, _ = synth.bench_corpus()
        d = text[:450000] + bytes(100000) + text[:300000] + bytes(range(256)) * 300
        for level in (6, 0, 2):
            z, _ = ref.stream_encode(14, d, level=level)
            # (TOTAL_OUT_MAX below the stream's real size is not compared: the reference's own accounting goes negative
            # there -- read() returns e.g. -133355 -- which no caller can rely on; mz_zip sets the exact size)
            for kw in (dict(max_in=len(z), max_out=len(d)), dict(), dict(chunk=10000), dict(chunk=300000)):
                a = hip.stream_decode(14, z, len(d) + 64, **kw)
                b = ref.stream_decode(14, z, len(d) + 64, **kw)
                assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (level, kw, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
            # cut: also inside the end marker WITH the entry's size set, as mz_zip sets it -- the call that would return the
            # entry's last bytes still has room, liblzma decodes on into it and fails there: -3 instead of those bytes
            for cut in (len(z) // 5, len(z) // 2, len(z) - 12, len(z) - 7, len(z) - 5, len(z) - 3, len(z) - 1):
                for how in ("eof", "max_in"):
                    for mo in (-1, len(d)):
                        kw = dict(max_in=cut if how == "max_in" else 0, max_out=mo)
                        a = hip.stream_decode(14, z[:cut] if how == "eof" else z, len(d) + 64, **kw)
                        b = ref.stream_decode(14, z[:cut] if how == "eof" else z, len(d) + 64, **kw)
                        assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (level, cut - len(z), how, mo, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
            bad = z[:len(z) * 2 // 3] + bytes([z[len(z) * 2 // 3] ^ 0x55]) + z[len(z) * 2 // 3 + 1:]
            a = hip.stream_decode(14, bad, len(d) + 64)
            b = ref.stream_decode(14, bad, len(d) + 64)
            assert (a["rets"], a["out"], a["close"], a["error"]) == (b["rets"], b["out"], b["close"], b["error"]), (level, "flip", a["rets"], b["rets"])
        # through the zip layer: an archive whose LZMA entries are larger than a window, CRC verified by mz_zip_entry_read_close
        import tempfile
        datas = [d, text[:300000] * 2]
        blob = np.frombuffer(b"".join(datas), dtype=np.uint8)
        lens = np.array([len(x) for x in datas], dtype=np.int32)
        offs = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "l.zip")
            ref.zip_write(path, blob, offs, lens, method=14, level=1)
            table = ref.zip_index(path)
            out = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
            _, crc, ulen, st = hip.zip_read_all(path, table[:, 6].copy(), nthreads=1, own_crc=False, out=out, out_off=offs)
            assert (st == 0).all() and (ulen == lens).all() and out[:-1].tobytes() == blob.tobytes()
    finally:
        L.mzhip_set_stream_window(0, 0)


def test_xz_window_mode(libs):
    """mz_stream_lzma READ of method 95 in window mode (shim_lzma.c xz_stream_*: the container walked on the host side, each
    block's LZMA2 chunks decoded window by window by mzhip_lzma2_run_host, the block check carried on the device): .xz
    entries of many windows -- CRC-64 / CRC-32 / no check, dictionaries smaller than the entry and the default 8 MiB,
    uncompressed chunks, several blocks in one stream -- whole, with and without the limits mz_zip sets, in small and large
    read() calls, cut at several places (inside a chunk, a check field, the index, the footer), corrupted in the payload,
    in a check field and in the index: every read() return value, byte, TOTAL_IN / TOTAL_OUT, close() and error() as the
    all-reference build.  Streams window mode does not take (a BCJ filter, a SHA-256 check) still decode."""
    import ctypes as C
    import lzma as pylzma

    hip, ref = libs
    L = hip.L
    L.mzhip_set_stream_window.argtypes = [C.c_int64, C.c_int64]
    L.mzhip_set_stream_window.restype = None
    L.mzhip_set_stream_window(192 << 10, 48 << 10)
    counter = getattr(L, "mzmock_lzma2_windows", None)  # (the mock counts the windows; the device library does not)
    try:
        text, _ = synth.bench_corpus()
        rnd = np.random.RandomState(5)
        d = text[:450000] + bytes(100000) + rnd.bytes(90000) + text[:300000] + bytes(range(256)) * 300
        streams = [("crc64 preset 6", d, pylzma.compress(d, format=pylzma.FORMAT_XZ)),
                   ("crc32 preset 0", d, pylzma.compress(d, format=pylzma.FORMAT_XZ, check=pylzma.CHECK_CRC32, preset=0)),
                   ("no check, 64 KiB dictionary, lc0 lp2", d,
                    pylzma.compress(d, format=pylzma.FORMAT_XZ, check=pylzma.CHECK_NONE,
                                    filters=[dict(id=pylzma.FILTER_LZMA2, preset=4, dict_size=1 << 16, lc=0, lp=2, pb=0)]))]
        parts = [text[:260000], rnd.bytes(70000), b"", text[260000:700000], bytes(200000)]
        streams.append(("five blocks", b"".join(parts), synth.xz_join([pylzma.compress(p, format=pylzma.FORMAT_XZ, preset=1) for p in parts])))
        for name, d, z in streams:
            w0 = counter() if counter else 0
            for kw in (dict(max_in=len(z), max_out=len(d)), dict(), dict(chunk=10000), dict(chunk=300000)):
                a = hip.stream_decode(95, z, len(d) + 64, **kw)
                b = ref.stream_decode(95, z, len(d) + 64, **kw)
                assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (name, kw, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
                assert a["out"] == d and a["close"] == 0
            if counter:
                assert counter() - w0 >= 4 * (len(d) // (192 << 10)), (name, counter() - w0)
            for cut in (len(z) // 5, len(z) // 2, len(z) - 40, len(z) - 13, len(z) - 5, len(z) - 1):
                for how in ("eof", "max_in"):
                    a = hip.stream_decode(95, z[:cut] if how == "eof" else z, len(d) + 64, max_in=cut if how == "max_in" else 0)
                    b = ref.stream_decode(95, z[:cut] if how == "eof" else z, len(d) + 64, max_in=cut if how == "max_in" else 0)
                    assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL}, (name, cut - len(z), how, {k: (a[k], b[k]) for k in ALL if k != "out" and a[k] != b[k]})
            for at in (len(z) * 2 // 3, len(z) - 30, len(z) - 14, len(z) - 3):   # payload, check field / index, footer
                bad = z[:at] + bytes([z[at] ^ 0x55]) + z[at + 1:]
                a = hip.stream_decode(95, bad, len(d) + 64)
                b = ref.stream_decode(95, bad, len(d) + 64)
                assert (a["rets"], a["out"], a["close"], a["error"] != 0) == (b["rets"], b["out"], b["close"], b["error"] != 0), (name, at - len(z), a["rets"][-3:], b["rets"][-3:])
        # what window mode leaves to the one-buffer path
        d = streams[0][1]
        for name, z in (("x86 filter", pylzma.compress(d, format=pylzma.FORMAT_XZ, filters=[dict(id=pylzma.FILTER_X86), dict(id=pylzma.FILTER_LZMA2, preset=1)])),
                        ("sha-256", pylzma.compress(d, format=pylzma.FORMAT_XZ, check=pylzma.CHECK_SHA256, preset=1))):
            w0 = counter() if counter else 0
            a = hip.stream_decode(95, z, len(d) + 64, max_in=len(z), max_out=len(d))
            b = ref.stream_decode(95, z, len(d) + 64, max_in=len(z), max_out=len(d))
            assert {k: a[k] for k in ALL} == {k: b[k] for k in ALL} and a["out"] == d, name
            assert not counter or counter() == w0
        # through the zip layer: CRC verified by mz_zip_entry_read_close
        import tempfile
        datas = [streams[0][1], text[:300000] * 2]
        blob = np.frombuffer(b"".join(datas), dtype=np.uint8)
        lens = np.array([len(x) for x in datas], dtype=np.int32)
        offs = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64))))
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "x.zip")
            ref.zip_write(path, blob, offs, lens, method=95, level=1)
            table = ref.zip_index(path)
            out = np.zeros(int(lens.sum()) + 1, dtype=np.uint8)
            _, crc, ulen, st = hip.zip_read_all(path, table[:, 6].copy(), nthreads=1, own_crc=False, out=out, out_off=offs)
            assert (st == 0).all() and (ulen == lens).all() and out[:-1].tobytes() == blob.tobytes()
    finally:
        L.mzhip_set_stream_window(0, 0)


def test_xz_window_mode_differential_fuzz(libs):
    """tests/fuzz_xz_windows.py as a test: random .xz streams (presets, hand-set lc / lp / pb / dictionary, three check
    kinds, one to four blocks) through the drop-in's method-95 READ in window mode and through the all-reference build --
    whole, with mz_zip's limits, cut, bit-flipped anywhere.  No mismatch; a corrupted stream may be refused one read()
    call apart (counted, bounded)."""
    from tests import fuzz_xz_windows as F

    hip, ref = libs
    n = 40 if getattr(hip.L, "mzmock_lzma2_windows", None) else 6      # (the emulation decodes 80 MB/s, one wave 3)
    cases, mism, soft = F.run(n, 31, hip=hip, ref=ref, verbose=True)
    print("xz window fuzz: %d cases, %d mismatches, %d corrupted streams refused a read() call apart" % (cases, mism, soft))
    assert mism == 0 and soft * 20 <= cases


def test_lzma_write_in_segments(libs):
    """mz_stream_lzma WRITE in bounded memory (shim_lzma.c): an entry larger than one segment leaves in segments.  Method 14:
    the range coder's state and the adaptive model are carried from launch to launch, so the payload is byte for byte what
    the one-shot coder makes -- and the all-reference READ stream decodes it.  (Method 95, one .xz block per segment:
    tests/test_gpu_lzma_enc.py, the mock has no .xz encoder.)"""
    import ctypes as C

    hip, ref = libs
    L = hip.L
    L.mzhip_set_write_segment.argtypes = [C.c_int64]
    L.mzhip_set_write_segment.restype = None
    text, _ = synth.bench_corpus()
    d = (text[:450000] + bytes(100000) + text[:300000]) * 2 + bytes(range(256)) * 100
    try:
        for lvl in (1, 6):
            L.mzhip_set_write_segment(0)
            z0, i0 = hip.stream_encode(14, d, level=lvl)              # one launch (the entry is smaller than 8 MiB)
            L.mzhip_set_write_segment(128 << 10)                      # segments of 128 KiB
            for chunk in (65535, 1000, 400000):
                z1, i1 = hip.stream_encode(14, d, level=lvl, chunk=chunk)
                assert z1 == z0 and i1 == i0, (lvl, chunk, len(z0), len(z1))
            b = ref.stream_decode(14, z1, len(d) + 64, max_in=len(z1), max_out=len(d))
            assert b["out"] == d and b["close"] == 0 and b["error"] == 0, lvl
    finally:
        L.mzhip_set_write_segment(0)


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


def test_window_mode_differential_fuzz(libs):
    """A short seed of tests/fuzz_gpu_windows.py in the suite (the fuzz was a script the driver never ran): random
    streams of 1 - 6 MB -- text, noise, runs, mixtures; levels 0 - 9; dynamic, fixed and stored blocks with flush points; raw,
    zlib and gzip framing -- through the drop-in's READ stream with a 3 MiB window and 512 KiB gulps (many-wave and serial
    windows both happen) and through the all-reference build: whole, cut at a random byte, with a random bit flipped.  Every
    return value, byte, total and verdict agrees; the ONE tolerated difference is TOTAL_IN at a DATA error, by at most
    TOTAL_IN_SLACK = 2 bytes (where inflate()'s bit buffer stood: SURVEY appendix B calls that field best effort)."""
    from tests import fuzz_gpu_windows as F

    hip, ref = libs
    cases, mism, soft, worst = F.run(8, 21, hip=hip, ref=ref, verbose=True)
    assert cases == 24 and mism == 0, (cases, mism, soft, worst)
    assert worst <= F.TOTAL_IN_SLACK, worst
