#!/usr/bin/env python3
"""Differential run of method-95 READ in window mode (not pytest): random .xz streams of 0.3 - 3 MB -- text, noise, runs,
mixtures; presets 0 - 9 and hand-set lc / lp / pb / dictionary sizes from 4 KiB up; CRC-64, CRC-32 and no check; one block
and several blocks per stream -- through the drop-in's mz_stream_lzma READ with a 192 KiB window and 48 KiB gulps and
through the all-reference build: whole (with and without the limits mz_zip sets), cut at a random byte, with a random bit
flipped anywhere (payload, block header, check field, index, footer).  Compared: every read() return value, the bytes,
close(), whether error() is set -- and TOTAL_IN / TOTAL_OUT / error() itself whenever the stream was not refused as corrupt
(at a data error liblzma's totals say how far its range decoder happened to run on into the garbage -- it checks a chunk's
compressed size only after the fact, lzma2_decoder.c -- which no caller can rely on: SURVEY appendix B, best effort).
    python tests/fuzz_xz_windows.py [streams] [seed] [library]"""
import ctypes as C
import lzma
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402

DROP = os.environ.get("MZ_FUZZ_LIB", os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so"))
KEYS = ("rets", "out", "close", "open")


def run(n_streams, seed, hip=None, ref=None, verbose=True, window=192 << 10, gulp=48 << 10):
    """-> (cases, mismatches, corrupted streams refused by both but not in the same read() call)"""
    rnd = random.Random(seed)
    hip = hip or oracle.MzDriver(DROP)
    ref = ref or oracle.ref()
    L = hip.L
    L.mzhip_set_stream_window.argtypes = [C.c_int64, C.c_int64]
    L.mzhip_set_stream_window.restype = None
    L.mzhip_set_stream_window(window, gulp)
    text = synth.bench_corpus()[0]
    try:
        return _run(n_streams, rnd, hip, ref, text, verbose)
    finally:
        L.mzhip_set_stream_window(0, 0)


def _run(n_streams, rnd, hip, ref, text, verbose):
    def piece():
        k = rnd.randrange(5)
        n = rnd.randrange(20000, 500000)
        if k == 0:
            o = rnd.randrange(len(text) - 1)
            return (text[o:] + text)[:n]
        if k == 1:
            return bytes(rnd.getrandbits(8) for _ in range(min(n, 150000)))
        if k == 2:
            return bytes([rnd.randrange(256)]) * n
        if k == 3:
            w = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40)))
            return (w * (n // len(w) + 1))[:n]
        return text[::-1][:n]

    mism = cases = soft = 0
    for it in range(n_streams):
        check = rnd.choice((lzma.CHECK_CRC64, lzma.CHECK_CRC64, lzma.CHECK_CRC32, lzma.CHECK_NONE))
        nblocks = rnd.choice((1, 1, 1, 2, 4))
        blocks, datas = [], []
        for _ in range(nblocks):
            d = b"".join(piece() for _ in range(rnd.randrange(1, 5)))
            if rnd.random() < 0.5:
                kw = dict(preset=rnd.randrange(0, 10))
            else:
                lc = rnd.randrange(0, 5)
                lp = rnd.randrange(0, 5 - lc)
                kw = dict(filters=[dict(id=lzma.FILTER_LZMA2, preset=rnd.randrange(0, 7), lc=lc, lp=lp, pb=rnd.randrange(0, 5),
                                        dict_size=1 << rnd.randrange(12, 22))])
            blocks.append(lzma.compress(d, format=lzma.FORMAT_XZ, check=check, **kw))
            datas.append(d)
        z = blocks[0] if nblocks == 1 else synth.xz_join(blocks)
        d = b"".join(datas)
        cap = len(d) + (1 << 16)
        variants = [("whole", z, dict()), ("limits", z, dict(max_in=len(z), max_out=len(d))),
                    ("cut", z[:rnd.randrange(len(z) // 4, len(z))], dict()),
                    ("cut tail", z[:len(z) - rnd.randrange(1, 60)], dict())]
        for _ in range(2):
            zz = bytearray(z)
            at = rnd.randrange(len(zz)) if rnd.random() < 0.6 else len(zz) - 1 - rnd.randrange(min(80, len(zz)))
            zz[at] ^= 1 << rnd.randrange(8)
            variants.append(("flip@%d" % (at - len(z)), bytes(zz), rnd.choice((dict(), dict(max_in=len(z), max_out=len(d))))))
        for name, data, kw in variants:
            # (read() calls that end exactly where a block or the stream ends: liblzma walks on through check, index and
            # footer with a full output buffer, as far as its staging buffer reaches)
            chunk = rnd.choice((65535, 65535, 1 << 20, 7777, max(len(d), 1), max(len(datas[0]), 1), max(len(d) // 2, 1)))
            a = hip.stream_decode(95, data, cap, chunk=chunk, **kw)
            b = ref.stream_decode(95, data, cap, chunk=chunk, **kw)
            cases += 1
            same = all(a[k] == b[k] for k in KEYS) and (a["error"] != 0) == (b["error"] != 0)
            if same and (b["error"] in (0, 10)) and not name.startswith("flip"):      # whole, or truncated: the accounting is exact (a CORRUPTED stream that
                # ends in LZMA_BUF_ERROR -- a flipped size field asks for bytes that are not there -- has liblzma's internal progress in its totals:
                # seed 203, stream 199: TOTAL_OUT 60 bytes apart, everything a caller sees equal)
                same = (a["total_in"], a["total_out"], a["error"]) == (b["total_in"], b["total_out"], b["error"])
            if not same and name.startswith("flip") and b["error"] == 9 and a["error"] == 9 and a["close"] == b["close"] and \
                    a["rets"][-1] == b["rets"][-1] and (a["out"].startswith(b["out"]) or b["out"].startswith(a["out"])):
                # a corrupted stream refused by both, the bytes in front of the refusal agree as far as both returned them, but not
                # the read() call that reports it: liblzma hands its LZMA decoder the input behind a chunk's compressed size and
                # checks the size afterwards (lzma2_decoder.c), so on garbage it can run on to the chunk's end where this decoder
                # stops at the chunk's last byte -- or the read() ends exactly where the stream does and the entry took the
                # one-buffer path, which does not model liblzma's walk through the trailer with a full output buffer
                soft += 1
                continue
            if not same:
                mism += 1
                if verbose:
                    print("MISMATCH stream %d %s blocks %d check %d chunk %d kw %s len %d:" % (it, name, nblocks, check, chunk, kw, len(data)),
                          {k: (a[k][-3:], b[k][-3:]) if k == "rets" else (a[k], b[k]) for k in KEYS + ("total_in", "total_out", "error") if k != "out" and a[k] != b[k]},
                          "bytes equal" if a["out"] == b["out"] else "BYTES DIFFER %d %d" % (len(a["out"]), len(b["out"])))
    return cases, mism, soft


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    lib = sys.argv[3] if len(sys.argv) > 3 else DROP
    cases, mism, soft = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1, hip=oracle.MzDriver(lib))
    print("xz window fuzz: %d streams, %d cases -- %d mismatches (%d corrupted streams refused by both, in different read() calls)" % (n, cases, mism, soft))
    sys.exit(1 if mism else 0)
