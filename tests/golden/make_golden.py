#!/usr/bin/env python3
"""Regenerate tests/golden/fixtures.json from the reference's own fixture archives.

Run in the build container (needs /root/reference and oracle/_ref):

    python tests/golden/make_golden.py

For every entry of every archive under test/fuzz/unzip_fuzzer_seed_corpus/ whose method is
STORE(0), DEFLATE(8), LZMA(14) or XZ(95) the script records the raw entry payload exactly as it sits
in the archive, plus the (crc32, compressed size, uncompressed size, flag) the archive's
central directory pins for it -- these are the golden (payload, bytes, CRC) triples that
third-party tools wrote and that the reference verifies at mz_zip.c:2116-2128.  Each payload is
also pushed through the compiled reference (oracle/_ref, mz_stream_zlib / mz_stream_lzma over a
memory stream) and the observed read() sequence, TOTAL_IN/TOTAL_OUT and sha256 of the output are
stored beside it.  /root/reference is not available on the GPU box, hence this committed file.
"""
import base64
import glob
import hashlib
import json
import os
import struct
import sys
import zipfile
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

SEEDS = "/root/reference/test/fuzz/unzip_fuzzer_seed_corpus"
SKIP = {"gh_739.zip", "gh_740.zip", "encrypted_pkcrypt.zip", "encrypted_wzaes.zip"}  # malformed / encrypted


def payload_of(raw, info):
    off = info.header_offset
    sig, = struct.unpack_from("<I", raw, off)
    assert sig == 0x04034B50
    fn, ex = struct.unpack_from("<HH", raw, off + 26)
    start = off + 30 + fn + ex
    return raw[start:start + info.compress_size]


def main():
    ref = oracle.ref()
    out = []
    for path in sorted(glob.glob(os.path.join(SEEDS, "*.zip"))):
        name = os.path.basename(path)
        if name in SKIP:
            continue
        raw = open(path, "rb").read()
        try:
            zf = zipfile.ZipFile(path)
        except Exception as e:  # noqa: BLE001
            print("skip", name, e)
            continue
        for info in zf.infolist():
            if info.compress_type not in (0, 8, 14, 95) or info.flag_bits & 1:
                continue
            if info.file_size > (1 << 20):
                continue
            pl = payload_of(raw, info)
            rec = dict(archive=name, entry=info.filename, method=info.compress_type, flag=info.flag_bits,
                       crc=info.CRC, csize=info.compress_size, usize=info.file_size,
                       payload=base64.b64encode(pl).decode())
            if info.compress_type == 0:
                data = pl
            else:
                kw = {}
                if info.compress_type in (14, 95) and info.flag_bits & 2:
                    kw = dict(max_in=info.compress_size, max_out=info.file_size)  # mz_zip.c:1842-1847
                r = ref.stream_decode(info.compress_type, pl, info.file_size + 64, **kw)
                data = r["out"]
                rec["ref"] = dict(rets=r["rets"], total_in=r["total_in"], total_out=r["total_out"],
                                  close=r["close"], error=r["error"])
            assert len(data) == info.file_size, (name, info.filename, len(data), info.file_size)
            assert zlib.crc32(data) == info.CRC, (name, info.filename)
            assert ref.crc32(data) == info.CRC
            rec["sha256"] = hashlib.sha256(data).hexdigest()
            out.append(rec)
            print("%-34s %-28s m=%-2d %7d -> %7d crc %08x" % (name, info.filename[:28], info.compress_type,
                                                             info.compress_size, info.file_size, info.CRC))
    dst = os.path.join(ROOT, "tests", "golden", "fixtures.json")
    with open(dst, "w") as f:
        json.dump(dict(source="minizip-ng 4.0.10 test/fuzz/unzip_fuzzer_seed_corpus", entries=out), f, indent=0)
    print(len(out), "entries ->", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
