"""CPU tests: pin the oracle restatement (oracle/*.c) against
(a) the golden (payload, bytes, CRC) triples of the reference's own fixture archives and
(b) the compiled reference itself (oracle/_ref: mz_strm_zlib.c / mz_strm_lzma.c / mz_crypt.c over
    zlib 1.2.11 / liblzma 5.2.5)."""
import hashlib
import lzma
import os
import zlib

import numpy as np
import pytest

import oracle
from tests import synth

needs_ref = pytest.mark.skipif(not (oracle.have_ref() or os.path.exists("/root/reference/mz_zip.c")),
                               reason="oracle/_ref not built and /root/reference absent")


def test_crc32_known_answers():
    # appnote.txt:837-847 polynomial; the classic check value
    assert oracle.crc32(b"123456789") == 0xCBF43926
    assert oracle.crc32(b"") == 0
    assert oracle.crc32(b"a") == 0xE8B7BE43


def test_crc32_chaining_and_combine():
    rnd = np.random.RandomState(1)
    data = rnd.bytes(100000)
    whole = oracle.crc32(data)
    assert whole == zlib.crc32(data)
    for cut in (0, 1, 7, 4096, 65535, 99999, 100000):
        a, b = data[:cut], data[cut:]
        assert oracle.crc32(b, oracle.crc32(a)) == whole            # mz_zip.c:2049 chaining
        assert oracle.crc32_combine(oracle.crc32(a), oracle.crc32(b), len(b)) == whole
    # 1-byte calls with inverted state, as mz_strm_pkcrypt.c:79,86 issues them
    v = 0x12345678
    for byte in data[:64]:
        v2 = ~oracle.crc32(bytes([byte]), ~v & 0xFFFFFFFF) & 0xFFFFFFFF
        v = v2
    assert v == (~zlib.crc32(data[:64], ~0x12345678 & 0xFFFFFFFF)) & 0xFFFFFFFF


def test_fixtures_golden(fixtures):
    """Every fixture entry decodes to the size and CRC its archive pins."""
    seen = {0: 0, 8: 0, 14: 0, 95: 0}
    for e in fixtures:
        if e["method"] == 0:
            data = e["payload"]
        elif e["method"] == 8:
            st, used, data = oracle.inflate_raw(e["payload"], e["usize"] + 16)
            assert st == 0 and used == e["csize"], (e["archive"], e["entry"], st, used)
            assert used == e["ref"]["total_in"]
        elif e["method"] == 95:
            st, used, data = oracle.xz_decode(e["payload"], e["usize"] + 16, e["usize"])
            assert st == 0 and used == e["csize"] == e["ref"]["total_in"], (e["archive"], e["entry"], st, used)
        else:
            st, used, data = oracle.lzma_zip_decode(e["payload"], e["usize"] + 16, e["usize"])
            assert st == 0 and used == e["csize"], (e["archive"], e["entry"], st, used)
            assert used == e["ref"]["total_in"]
        assert len(data) == e["usize"]
        assert oracle.crc32(data) == e["crc"], (e["archive"], e["entry"])
        assert hashlib.sha256(data).hexdigest() == e["sha256"]
        seen[e["method"]] += 1
    assert seen[0] >= 10 and seen[8] >= 10 and seen[14] >= 1 and seen[95] >= 1


def test_inflate_edges_vs_zlib():
    for name, data, z in synth.edge_payloads():
        st, used, out = oracle.inflate_raw(z + b"\x00garbage", len(data) + 8)
        assert st == 0, name
        assert used == len(z), name
        assert out == data, name


@needs_ref
def test_inflate_status_parity_with_reference():
    """Error class and exact TOTAL_IN for malformed streams, against the live reference."""
    ref = oracle.ref()
    n = 0
    for name, data, z in synth.edge_payloads():
        if len(z) < 16:
            continue
        for cname, bad in synth.corruptions(z):
            r = ref.stream_decode(8, bad, len(data) + 70000)
            st, used, out = oracle.inflate_raw(bad, len(data) + 70000)
            last = r["rets"][-1] if r["rets"] else 0
            ref_status = last if last < 0 else 0
            assert st == ref_status, (name, cname, st, r["rets"], r["error"])
            if st == 0:
                assert out == r["out"], (name, cname)
                assert used == r["total_in"], (name, cname)
            n += 1
    assert n > 100


INCOMPLETE_DISTANCE_SET = (  # tests/fuzz_gpu.py 40000 606, streams 35526 and 137369: a block whose distance set is ONE code of one
    # bit (what zlib writes when a block uses one distance), a flipped bit makes a match take the unused code, and the input
    # ends within the fourteen bits behind it
    bytes.fromhex("edc10109000000c32058ffd0cf71040b00000000002e0c"),
    bytes.fromhex("edc1010d000000c2a0bd7f6983088b0000000000000000000000000000000000000000000000007067"),
)


def _zlib_status(z, wbits=-15):
    d = zlib.decompressobj(wbits)
    try:
        out = d.decompress(z)
    except zlib.error:
        return -3, None
    return (0 if d.eof else -5), out


def test_unused_code_of_an_incomplete_set_is_refused_on_its_one_bit():
    """zlib only accepts an incomplete Huffman set whose longest code has one bit and refuses the unused one-bit code as soon
    as it sees it (Z_DATA_ERROR) -- also when the input ends right behind it, where a bit-by-bit canonical walk would still ask
    for more bits (Z_BUF_ERROR).  The restatement against the zlib of this interpreter at every cut of the two streams the
    device fuzz found (the device agreed with the reference, the restatement did not)."""
    for z in INCOMPLETE_DISTANCE_SET:
        for n in range(1, len(z) + 1):
            want, out = _zlib_status(z[:n])
            st, used, got = oracle.inflate_raw(z[:n], 100000)
            assert st == want, (z.hex(), n, st, want)
            if st == 0:
                assert got == out
        assert oracle.inflate_raw(z, 100000)[0] == -3


@needs_ref
def test_incomplete_set_status_with_reference():
    ref = oracle.ref()
    for z in INCOMPLETE_DISTANCE_SET:
        for n in range(1, len(z) + 1):
            r = ref.stream_decode(8, z[:n], 100000)
            st, used, out = oracle.inflate_raw(z[:n], 100000)
            last = r["rets"][-1] if r["rets"] else 0
            assert st == (last if last < 0 else 0), (z.hex(), n, st, r["rets"])
            if st == -3:
                assert len(out) == r["total_out"], (n, len(out), r["total_out"])


@needs_ref
def test_empty_code_length_code_status_with_reference():
    """A dynamic block whose code-length code has no code at all: inflate() reads the nlen + ndist lengths as zeros of one bit
    each before it refuses the block (-3), and asks for more input (-5) when the stream ends inside those bits.  The oracle's
    verdict against the live reference at every cut."""
    ref = oracle.ref()
    for hlit, hdist, hclen in ((0, 0, 0), (29, 29, 15), (7, 3, 2)):
        hdr = 1 | (2 << 1) | (hlit << 3) | (hdist << 8) | (hclen << 13)
        bits = 17 + 3 * (hclen + 4) + (hlit + 257) + (hdist + 1)
        z = hdr.to_bytes(3, "little") + bytes(80)
        for n in range(3, (bits + 7) // 8 + 3):
            r = ref.stream_decode(8, z[:n], 4096)
            st, used, out = oracle.inflate_raw(z[:n], 4096)
            assert st == r["rets"][-1], (hlit, hdist, hclen, n, st, r["rets"])
            assert st == (-5 if 8 * n < bits else -3), (n, bits, st)


@needs_ref
def test_crc32_matches_reference():
    ref = oracle.ref()
    rnd = np.random.RandomState(5)
    for n in (0, 1, 2, 3, 255, 256, 65535, 65536, 1 << 20):
        d = rnd.bytes(n)
        assert oracle.crc32(d) == ref.crc32(d)
        assert oracle.crc32(d, 0xDEADBEEF) == ref.crc32(d, 0xDEADBEEF)


def _zip_lzma(data, preset=6, eos=True):
    """ZIP method-14 payload as mz_stream_lzma writes it (mz_strm_lzma.c:94-104,250-265)."""
    filt = [dict(id=lzma.FILTER_LZMA1, preset=preset)]
    raw = lzma.compress(data, format=lzma.FORMAT_ALONE, filters=filt)
    # .lzma alone = props(5) + size(8) + stream; python writes size = -1 + EOS marker
    assert raw[5:13] == b"\xff" * 8
    return bytes([5, 2, 5, 0]) + raw[:5] + raw[13:]


@needs_ref
def test_lzma_parity_with_reference():
    ref = oracle.ref()
    c = synth.corpus()
    rnd = np.random.RandomState(11)
    cases = [b"", b"a", c[:1000], c[:150000], rnd.bytes(5000), b"A" * 100000, c[1000:70000] + rnd.bytes(3000) + c[:50000]]
    for i, data in enumerate(cases):
        z = _zip_lzma(data)
        r = ref.stream_decode(14, z, len(data) + 64, max_in=len(z), max_out=len(data))
        assert r["out"] == data and r["error"] == 0, i
        st, used, out = oracle.lzma_zip_decode(z, len(data) + 64, len(data))
        assert st == 0 and out == data, i
        assert used == r["total_in"] == len(z), (i, used, r["total_in"], len(z))
        # reference-encoded (mz_stream_lzma WRITE) payloads too
        z2, info = ref.stream_encode(14, data)
        st, used, out = oracle.lzma_zip_decode(z2, len(data) + 64, len(data))
        assert st == 0 and out == data and used == len(z2), i
        # malformed variants: status class only
        if len(z) > 40:
            for bad in (z[:len(z) // 2], z[:20], z[:9], z[:5],
                        z[:len(z) // 3] + bytes([z[len(z) // 3] ^ 0x55]) + z[len(z) // 3 + 1:],
                        z[:9] + b"\x01" + z[10:]):
                r = ref.stream_decode(14, bad, len(data) + 70000)
                last = r["rets"][-1] if r["rets"] else r["open"]
                st, used, out = oracle.lzma_zip_decode(bad, len(data) + 70000, -1)
                if last < 0:
                    assert st == -3, (i, len(bad), st, r)
                else:
                    assert st == 0 and out == r["out"], (i, len(bad), st)


def test_lzma_properties_byte_rules():
    """liblzma refuses lc + lp > 4 with the header (lzma_lzma_lclppb_decode) although the LZMA specification allows lc up to 8,
    and every properties byte from 225 up; a run of one byte decodes to the same bytes under ANY lc / lp (no literal context is
    ever used twice), so only the header check tells them apart (tests/fuzz_oracle_lzma.py: 224 of 420 000 cases, round 6)."""
    data = b"q" * 5000
    z = _zip_lzma(data)
    ref = oracle.ref() if oracle.have_ref() else None
    for props in range(256):
        lc, lp = props % 9, (props // 9) % 5
        bad = z[:4] + bytes([props]) + z[5:]
        st, used, out = oracle.lzma_zip_decode(bad, len(data) + 64, -1)
        if props >= 225 or lc + lp > 4:
            assert st == -3 and out == b"", (props, st)
        if ref is not None:
            r = ref.stream_decode(14, bad, len(data) + 64)
            last = r["rets"][-1] if r["rets"] else r["open"]
            assert (st == -3) == (last < 0), (props, st, r["rets"], r["open"])
            if last >= 0:
                assert st == 0 and out == r["out"], props


def test_xz_checks_known_answers():
    """CRC-64/XZ and SHA-256 (the .xz check ids 4 and 10) against published check values and hashlib."""
    assert oracle.crc64(b"123456789") == 0x995DC9BBDF1939FA          # CRC-64/XZ check value (ECMA-182 reflected)
    assert oracle.crc64(b"") == 0
    rnd = np.random.RandomState(2)
    for n in (0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 1000, 100001):
        d = rnd.bytes(n)
        assert oracle.sha256(d).hex() == hashlib.sha256(d).hexdigest(), n
    assert oracle.sha256(b"abc").hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"


def test_xz_decode_cases_vs_python_lzma():
    """Streams written by liblzma (through Python's lzma) -- presets, all verified check ids, lc/lp/pb variety,
    stored chunks, multi-chunk, hand-joined multi-block -- decode to the input; trailing bytes stay unconsumed."""
    for name, d, x in synth.xz_cases():
        st, used, out = oracle.xz_decode(x + b"tail", len(d) + 64)
        assert (st, used, out) == (0, len(x), d), name
        assert lzma.decompress(x) == d, name


def test_xz_filter_chain_rules_vs_liblzma():
    """The chains liblzma refuses (Python's lzma module is liblzma: LZMA_OPTIONS_ERROR) are data errors here, as
    mz_stream_lzma_read reports every liblzma failure (mz_strm_lzma.c:236-237)."""
    for name, x in synth.xz_bad_chain_cases():
        with pytest.raises(lzma.LZMAError):
            lzma.decompress(x)
        assert oracle.xz_decode(x, 10000)[0] == -3, name
        if oracle.have_ref():
            r = oracle.ref().stream_decode(95, x, 10000)
            assert r["rets"][-1] == -3, name


@needs_ref
def test_xz_parity_with_reference():
    """Valid streams: bytes and TOTAL_IN equal to mz_stream_lzma_read(method 95).  3000 corrupted / truncated
    streams: the same accept / reject decision, and on accept the same bytes and TOTAL_IN."""
    import random

    ref = oracle.ref()
    cases = synth.xz_cases()
    for name, d, x in cases:
        r = ref.stream_decode(95, x + b"tail", len(d) + 64)
        assert (r["out"], r["total_in"], r["error"], r["close"]) == (d, len(x), 0, 0), name
    rnd = random.Random(5)
    bases = [x for _, d, x in cases if 0 < len(d) <= 100000]
    for it in range(3000):
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
        st, used, out = oracle.xz_decode(x, 200000)
        if st == -109:
            continue            # filter chains outside the backend's scope
        r = ref.stream_decode(95, x, 200000)
        ok_ref = r["error"] in (0, 1) and r["rets"][-1] >= 0
        assert (st == 0) == ok_ref, (it, k, st, r["rets"][-2:], r["error"])
        if ok_ref:
            assert (out, used) == (r["out"], r["total_in"]), (it, k)
        else:
            assert r["rets"][-1] == -3 and st in (-3, -5), (it, k, st)
