"""Deterministic synthetic inputs shared by the tests (and mirrored by bench.py).

Corpus C = the English prose shipped with CPython (pydoc_data.topics, ~460 KB): present in
the image both here and on the GPU box, zlib level-6 ratio ~0.31 on 64 KiB slices -- the
stand-in for the "enwik-slice" entries BASELINE.json names (no enwik, no network)."""
import random
import zlib

import numpy as np


def corpus():
    import pydoc_data.topics as t

    return "".join(t.topics[k] for k in sorted(t.topics)).encode()


def bench_corpus():
    """(bytes, description): the SURVEY 8(d) corpus when oracle/_ref/corpus.bin travelled (built by
    oracle/make_corpus.py from the reference tree), else the CPython prose above."""
    import os

    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "corpus.bin")
    if os.path.exists(p):
        return open(p, "rb").read(), "appnote.txt || appnote.iz.txt || alice29.txt of the reference tree (SURVEY 8d)"
    return corpus(), "CPython's pydoc prose (oracle/_ref/corpus.bin absent)"


_CHAIN = {}


def markov_entries(n_unique, size, seed=3, text=None, jump=0.45):
    """Seeded order-2 word-Markov expansion of the corpus (mandatory for config 4: the corpus is only
    ~470 KB, so 1 MiB entries tiled from slices would repeat inside LZMA's dictionary).  `jump` = probability of leaving
    the chain at a word: 0.45 gives liblzma preset 6 a ratio of ~0.25 on the SURVEY corpus (0.15 without jumps)."""
    rnd = random.Random(seed)
    src = text if text is not None else corpus()
    ck = (len(src), zlib.crc32(src))
    if _CHAIN.get("key") != ck:  # the chain of one corpus is built once per process (bench.py's workers make one entry per call)
        words = src.split()
        nxt = {}
        for a, b, c in zip(words, words[1:], words[2:]):
            nxt.setdefault((a, b), []).append(c)
        _CHAIN.update(key=ck, nxt=nxt, keys=list(nxt))
    nxt, keys = _CHAIN["nxt"], _CHAIN["keys"]
    out = []
    for _ in range(n_unique):
        a, b = keys[rnd.randrange(len(keys))]
        parts, total = [a, b], len(a) + len(b) + 2
        while total < size + 64:                         # a few words more than needed: the slice below is exactly `size`
            cand = nxt.get((a, b))
            if not cand or rnd.random() < jump:          # dead end, or a jump: keeps phrases short enough for an LZMA ratio of ~0.25
                a, b = keys[rnd.randrange(len(keys))]
                parts += [a, b]
                total += len(a) + len(b) + 2
                continue
            c = cand[rnd.randrange(len(cand))]
            parts.append(c)
            total += len(c) + 1
            a, b = b, c
        out.append(b" ".join(parts)[:size])
    return out


def slices(n, size, seed=1234):
    c = corpus()
    rnd = random.Random(seed)
    return [c[o:o + size] for o in (rnd.randrange(len(c) - size) for _ in range(n))]


def deflate_raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    """Exactly the reference writer's parameters (mz_strm_zlib.c:87: raw, 32 KiB window, memLevel 8)."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(data) + c.flush()


def stored_blocks(data, block=65535):
    """A raw DEFLATE stream made of stored blocks only."""
    out = bytearray()
    if not data:
        return bytes([1, 0, 0, 0xFF, 0xFF])
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        last = 1 if i + block >= len(data) else 0
        out += bytes([last]) + len(chunk).to_bytes(2, "little") + (len(chunk) ^ 0xFFFF).to_bytes(2, "little") + chunk
    return bytes(out)


def edge_payloads():
    """(name, uncompressed bytes, raw deflate bytes) covering the shapes SURVEY 8(d) lists."""
    rnd = np.random.RandomState(7)
    c = corpus()
    cases = []

    def add(name, data, **kw):
        cases.append((name, data, deflate_raw(data, **kw)))

    add("empty", b"")
    add("one_byte", b"x")
    add("run_A_65535", b"A" * 65535)                      # dist=1 overlap, len 258 chains
    add("run_ab", b"ab" * 5000)                           # dist=2 < len
    add("text_64k_l6", c[1000:1000 + 65536])
    add("text_64k_l1", c[5000:5000 + 65536], level=1)
    add("text_64k_l9", c[9000:9000 + 65536], level=9)
    add("text_8k", c[20000:20000 + 8192])
    add("text_fixed", c[30000:30000 + 20000], strategy=zlib.Z_FIXED)
    add("text_huffman_only", c[40000:40000 + 30000], strategy=zlib.Z_HUFFMAN_ONLY)
    add("text_rle", c[50000:50000 + 30000], strategy=zlib.Z_RLE)
    add("random_incompressible", rnd.bytes(70000))        # zlib emits stored blocks
    add("multi_block_300k", c[:300000])                   # several dynamic blocks
    far = rnd.bytes(300)
    add("max_distance", far + rnd.bytes(32768 - 300) + far + c[:500] + far)   # back-refs at distance 32768
    add("binaryish", bytes((i * i >> 3) & 0xFF for i in range(50000)))
    cases.append(("stored_only", c[:70000], stored_blocks(c[:70000])))
    cases.append(("stored_empty", b"", stored_blocks(b"")))
    # a sync-flushed stream: empty stored blocks between dynamic blocks
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 8)
    z = co.compress(c[:10000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(c[10000:25000]) + co.flush(
        zlib.Z_FULL_FLUSH) + co.compress(c[25000:26000]) + co.flush()
    cases.append(("sync_flushed", c[:26000], z))
    return cases


def corruptions(z, seed=3):
    """(name, bytes) malformed variants of a valid raw-deflate stream."""
    rnd = random.Random(seed)
    out = [("truncated_half", z[:len(z) // 2]), ("truncated_1", z[:-1]), ("empty_input", b""),
           ("reserved_btype", bytes([z[0] | 0x06]) + z[1:])]
    for k in range(6):
        i = rnd.randrange(len(z))
        out.append(("flip_%d" % i, z[:i] + bytes([z[i] ^ (1 << rnd.randrange(8))]) + z[i + 1:]))
    i = len(z) // 3
    out.append(("xor55_third", z[:i] + bytes([z[i] ^ 0x55]) + z[i + 1:]))
    return out


# ---- .xz helpers (method 95) -------------------------------------------------------------------------------
def _vli(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _read_vli(b, p):
    v = s = 0
    while True:
        v |= (b[p] & 0x7F) << s
        s += 7
        p += 1
        if not b[p - 1] & 0x80:
            return v, p


def xz_blocks(x):
    """Split a single-stream .xz image into (check_id, [(block_bytes, unpadded_size, uncompressed_size)])."""
    check = x[7]
    csz = (0, 4, 4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64)[check]
    isize = (int.from_bytes(x[-8:-4], "little") + 1) * 4
    ipos = len(x) - 12 - isize
    n, p = _read_vli(x, ipos + 1)
    recs = []
    for _ in range(n):
        unp, p = _read_vli(x, p)
        usz, p = _read_vli(x, p)
        recs.append((unp, usz))
    pos, out = 12, []
    for unp, usz in recs:
        padded = (unp + 3) & ~3
        out.append((x[pos:pos + padded], unp, usz))
        pos += padded
    assert pos == ipos and csz >= 0
    return check, out


def xz_join(streams):
    """One .xz stream holding every block of the given single-stream images (same check id): a multi-block
    stream like `xz -T` / --block-size writes."""
    check, blocks = None, []
    for x in streams:
        c, b = xz_blocks(x)
        assert check in (None, c)
        check = c
        blocks += b
    flags = bytes([0, check])
    out = b"\xfd7zXZ\x00" + flags + zlib.crc32(flags).to_bytes(4, "little")
    for blk, _, _ in blocks:
        out += blk
    idx = b"\x00" + _vli(len(blocks)) + b"".join(_vli(u) + _vli(s) for _, u, s in blocks)
    idx += b"\x00" * (-len(idx) % 4)
    idx += zlib.crc32(idx).to_bytes(4, "little")
    tail = (len(idx) // 4 - 1).to_bytes(4, "little") + flags
    return out + idx + zlib.crc32(tail).to_bytes(4, "little") + tail + b"YZ"


def xz_cases():
    """(name, data, xz_stream) covering presets, every verified check id, lc/lp/pb variety, stored (uncompressed)
    LZMA2 chunks, multi-chunk and multi-block streams."""
    import lzma

    c = corpus()
    rnd = np.random.RandomState(11)
    noise = rnd.bytes(70000)
    cases = []
    for name, d in (("text100k", c[:100000]), ("empty", b""), ("one", b"a"), ("noise", noise), ("zeros", bytes(300000)),
                    ("text+noise", c[:50000] + noise + c[50000:90000])):
        for chk in (lzma.CHECK_NONE, lzma.CHECK_CRC32, lzma.CHECK_CRC64, lzma.CHECK_SHA256):
            cases.append(("%s/check%d" % (name, chk), d, lzma.compress(d, format=lzma.FORMAT_XZ, check=chk, preset=6)))
        for preset in (0, 9):
            cases.append(("%s/preset%d" % (name, preset), d, lzma.compress(d, format=lzma.FORMAT_XZ, preset=preset)))
        for lc, lp, pb, ds in ((0, 2, 0, 4096), (4, 0, 4, 1 << 16), (0, 4, 2, 1 << 20), (1, 3, 1, 1 << 12)):
            f = [{"id": lzma.FILTER_LZMA2, "lc": lc, "lp": lp, "pb": pb, "dict_size": ds}]
            cases.append(("%s/lc%dlp%dpb%d" % (name, lc, lp, pb), d, lzma.compress(d, format=lzma.FORMAT_XZ, filters=f)))
    big = c + c[:200000]          # > 2 MiB would need a bigger corpus; several 64 KiB-compressed chunks is what matters
    cases.append(("multi-chunk", big, lzma.compress(big, format=lzma.FORMAT_XZ, preset=1)))
    parts = [c[:70000], noise[:5000], b"", c[70000:200000]]
    cases.append(("multi-block", b"".join(parts), xz_join([lzma.compress(p, format=lzma.FORMAT_XZ) for p in parts])))
    cases.append(("multi-block-sha", b"".join(parts),
                  xz_join([lzma.compress(p, format=lzma.FORMAT_XZ, check=lzma.CHECK_SHA256) for p in parts])))
    cases += xz_filter_cases()
    return cases


def _branch_soup(rnd, n, kind):
    """n bytes of noise in which the branch patterns the BCJ filter `kind` converts are frequent"""
    b = bytearray(rnd.randrange(256) for _ in range(n))
    i = 0
    while i + 16 < n:
        r = rnd.random()
        if kind == "x86" and r < 0.3:
            b[i] = rnd.choice((0xE8, 0xE9))
            b[i + 4] = rnd.choice((0x00, 0xFF, 0x00, 0xFF, rnd.randrange(256)))
            if rnd.random() < 0.3:
                b[i + 1] = rnd.choice((0xE8, 0xE9))       # calls on top of each other: the prev_mask paths
            i += rnd.randrange(1, 9)
        elif kind == "arm" and r < 0.3:
            b[(i & ~3) + 3] = 0xEB
            i += 4
        elif kind == "armthumb" and r < 0.4:
            j = i & ~1
            b[j + 1] = 0xF0 | rnd.randrange(8)
            b[j + 3] = 0xF8 | rnd.randrange(8)
            i += rnd.choice((2, 2, 4, 6))                  # overlapping candidates: the skip-after-a-pair rule
        elif kind == "powerpc" and r < 0.3:
            j = i & ~3
            b[j] = 0x48 | rnd.randrange(4)
            b[j + 3] = (b[j + 3] & ~3) | 1
            i += 4
        elif kind == "sparc" and r < 0.3:
            j = i & ~3
            if rnd.random() < 0.5:
                b[j], b[j + 1] = 0x40, b[j + 1] & 0x3F
            else:
                b[j], b[j + 1] = 0x7F, b[j + 1] | 0xC0
            i += 4
        elif kind == "ia64" and r < 0.5:
            j = i & ~15
            b[j] = (b[j] & 0xE0) | rnd.choice((16, 17, 18, 19, 22, 23, 24, 25, 28, 29))
            for k in range(1, 16):
                if rnd.random() < 0.5:
                    b[j + k] = rnd.choice((0x00, 0x50, 0xA0, 0x0A, 0x28, 0x14, 0x05))
            i += 16
        else:
            i += rnd.randrange(1, 7)
    return bytes(b)


def xz_bad_chain_cases():
    """(name, xz_stream): block headers whose filter chain liblzma refuses (LZMA_OPTIONS_ERROR): a misaligned BCJ start
    offset, an unknown filter id, LZMA2 in front of another filter, Delta last.  Made from valid streams by editing the
    filter flags and re-sealing the header's CRC32."""
    import lzma

    lz2 = {"id": lzma.FILTER_LZMA2, "preset": 1}
    d = corpus()[:3000]

    def reseal(x, edit):
        x = bytearray(x)
        hsize = (x[12] + 1) * 4
        edit(x, 12)
        x[12 + hsize - 4:12 + hsize] = zlib.crc32(bytes(x[12:12 + hsize - 4])).to_bytes(4, "little")
        return bytes(x)

    out = []
    x = lzma.compress(d, format=lzma.FORMAT_XZ, filters=[{"id": lzma.FILTER_ARM, "start_offset": 4096}, lz2])
    out.append(("arm start offset 4098", reseal(x, lambda b, h: b.__setitem__(h + 4, 2))))      # id 07, size 04, offset bytes
    x = lzma.compress(d, format=lzma.FORMAT_XZ, filters=[{"id": lzma.FILTER_X86}, lz2])
    out.append(("unknown filter 0x0A", reseal(x, lambda b, h: b.__setitem__(h + 2, 0x0A))))
    out.append(("lzma2 in front", reseal(x, lambda b, h: b.__setitem__(slice(h + 2, h + 7), bytes([0x21, 0x01, b[h + 6], 0x04, 0x00])))))
    x = lzma.compress(d, format=lzma.FORMAT_XZ, filters=[{"id": lzma.FILTER_DELTA, "dist": 1}, lz2])
    out.append(("delta last", reseal(x, lambda b, h: b.__setitem__(slice(h + 5, h + 8), bytes([0x03, 0x01, 0x00])))))
    return out


def xz_filter_cases():
    """(name, data, xz_stream): .xz blocks whose filter chain has Delta / BCJ filters in front of LZMA2 -- every filter
    liblzma 5.2.5 knows, with and without a start offset, sizes around the filters' unit and look-ahead, more than one
    8 KiB round of the device's unfilter, and chains of two and three filters -- written by liblzma itself."""
    import lzma
    import random

    rnd = random.Random(77)
    bcj = (("x86", lzma.FILTER_X86, 1), ("powerpc", lzma.FILTER_POWERPC, 4), ("ia64", lzma.FILTER_IA64, 16),
           ("arm", lzma.FILTER_ARM, 4), ("armthumb", lzma.FILTER_ARMTHUMB, 2), ("sparc", lzma.FILTER_SPARC, 4))
    lz2 = {"id": lzma.FILTER_LZMA2, "preset": 1}
    cases = []
    for kind, fid, align in bcj:
        for k, n in enumerate((0, 3, 4, 5, 15, 16, 17, 5000, 70000)):
            f = {"id": fid}
            if k % 2:
                f["start_offset"] = align * rnd.randrange(1, 1 << 20)
            d = _branch_soup(rnd, n, kind)
            cases.append(("filter/%s/%d" % (kind, n), d, lzma.compress(d, format=lzma.FORMAT_XZ, filters=[f, lz2])))
        d = bytes(rnd.getrandbits(8) for _ in range(40000)) + _branch_soup(rnd, 60000, kind)
        cases.append(("filter/%s/100k-crc64" % kind, d, lzma.compress(d, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC64,
                                                                     filters=[{"id": fid, "start_offset": align * 777}, lz2])))
    for dist in (1, 2, 3, 4, 7, 63, 64, 65, 255, 256):
        n = rnd.choice((0, 1, dist, dist + 1, 3000, 50000))
        d = bytes((rnd.randrange(256) if rnd.random() < 0.1 else (i * 3) & 255) for i in range(n))
        cases.append(("filter/delta%d/%d" % (dist, n), d,
                      lzma.compress(d, format=lzma.FORMAT_XZ, filters=[{"id": lzma.FILTER_DELTA, "dist": dist}, lz2])))
    d = _branch_soup(rnd, 40000, "x86")
    cases.append(("filter/x86+delta4", d, lzma.compress(d, format=lzma.FORMAT_XZ, check=lzma.CHECK_SHA256, filters=[
        {"id": lzma.FILTER_X86}, {"id": lzma.FILTER_DELTA, "dist": 4}, lz2])))
    cases.append(("filter/delta1+delta2+arm", d, lzma.compress(d, format=lzma.FORMAT_XZ, filters=[
        {"id": lzma.FILTER_DELTA, "dist": 1}, {"id": lzma.FILTER_DELTA, "dist": 2}, {"id": lzma.FILTER_ARM}, lz2])))
    return cases


def long_code_payloads():
    """(name, data, raw_deflate): streams whose dynamic Huffman codes reach 13-15 bits and use many symbols, so the
    decoder's second-level literal/length tables are exercised up to their capacity (zlib Z_HUFFMAN_ONLY / level 9 on
    data with geometric byte statistics, all 256 byte values present)."""
    out = []
    for seed, ratio in ((1, 0.5), (2, 0.6), (3, 0.7), (4, 0.8)):
        rnd = np.random.RandomState(seed)
        p = ratio ** np.arange(256, dtype=np.float64)
        p /= p.sum()
        perm = rnd.permutation(256)
        d = perm[rnd.choice(256, size=200000, p=p)].astype(np.uint8).tobytes()
        d += bytes(range(256)) * 2                      # every byte value at least twice
        for strat, nm in ((zlib.Z_HUFFMAN_ONLY, "huff"), (zlib.Z_DEFAULT_STRATEGY, "l9")):
            out.append(("geom%.1f/%s" % (ratio, nm), d, deflate_raw(d, level=9, strategy=strat)))
    return out
