#!/usr/bin/env python3
"""oracle/make_corpus.py -- TEST / BENCH INFRASTRUCTURE.  Builds the synthetic-data corpus SURVEY 8(d) names,
    C = doc/zip/appnote.txt || doc/zip/appnote.iz.txt || alice29.txt
from the reference tree where it lies (alice29.txt is the xz-compressed entry of
test/fuzz/unzip_fuzzer_seed_corpus/xz.zip), into oracle/_ref/corpus.bin.  oracle/_ref/ is git-ignored (nothing of the
reference enters the history) but travels with the gpurun snapshot like the compiled reference beside it, so bench.py
and the probes find the same corpus on the GPU box, where /root/reference does not exist.
Usage: python oracle/make_corpus.py [reference root] [output]"""
import lzma
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def zip_entry_payload(raw, name):
    """(method, payload bytes) of one entry of a small, non-ZIP64 archive: central directory walk, appnote 4.3.12"""
    eocd = raw.rfind(b"PK\x05\x06")
    n, cd_size, cd_off = struct.unpack_from("<HII", raw, eocd + 10)
    p = cd_off
    for _ in range(n):
        assert raw[p:p + 4] == b"PK\x01\x02"
        method, = struct.unpack_from("<H", raw, p + 10)
        csize, usize, fn, ex, cm = struct.unpack_from("<IIHHH", raw, p + 20)
        loff, = struct.unpack_from("<I", raw, p + 42)
        if raw[p + 46:p + 46 + fn].decode() == name:
            lfn, lex = struct.unpack_from("<HH", raw, loff + 26)
            q = loff + 30 + lfn + lex
            return method, raw[q:q + csize], usize
        p += 46 + fn + ex + cm
    raise KeyError(name)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "_ref", "corpus.bin")
    parts = []
    for f in ("doc/zip/appnote.txt", "doc/zip/appnote.iz.txt"):
        parts.append(open(os.path.join(ref, f), "rb").read())
    raw = open(os.path.join(ref, "test/fuzz/unzip_fuzzer_seed_corpus/xz.zip"), "rb").read()
    method, payload, usize = zip_entry_payload(raw, "alice29.txt")
    assert method == 95, method  # MZ_COMPRESS_METHOD_XZ: the payload is a complete .xz stream
    alice = lzma.decompress(payload, format=lzma.FORMAT_XZ)
    assert len(alice) == usize == 152089, len(alice)
    parts.append(alice)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "wb") as f:
        f.write(b"".join(parts))
    print("%s: %d bytes (%s)" % (out, sum(len(p) for p in parts), " + ".join(str(len(p)) for p in parts)))


if __name__ == "__main__":
    main()
