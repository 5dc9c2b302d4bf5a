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
