#!/usr/bin/env python3
"""Differential run of window mode on the device (not pytest): random streams of 1 - 6 MB -- text, noise, runs, mixtures;
levels 0 - 9; dynamic, fixed and stored blocks with flush points; raw, zlib and gzip framing -- through the drop-in's
mz_stream_zlib READ with a 3 MiB window and 512 KiB gulps (many-wave and serial windows both happen) and through the
all-reference build: whole, cut at a random byte, with a random bit flipped.  Everything the driver reports is compared
(after a data error the base position is not: window mode pulls ahead; TOTAL_IN at a data error is counted apart).
    python tests/fuzz_gpu_windows.py [streams] [seed]"""
import ctypes as C
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402

ONLY = set(int(x) for x in os.environ.get("MZ_FUZZ_ONLY", "").split(",") if x)
DROP = os.environ.get("MZ_FUZZ_LIB", os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so"))
KEYS = ("rets", "out", "total_in", "total_out", "close", "error", "open")
TOTAL_IN_SLACK = 2  # bytes: at a DATA error the reference's TOTAL_IN is where inflate()'s bit buffer stood (best effort)


def run(n_streams, seed, hip=None, ref=None, verbose=True):
    """-> (cases, hard mismatches, cases that agree in everything but TOTAL_IN at a data error, the largest such difference)"""
    rnd = random.Random(seed)
    hip = hip or oracle.MzDriver(DROP)
    ref = ref or oracle.ref()
    L = hip.L
    L.mzhip_set_stream_window.argtypes = [C.c_int64, C.c_int64]
    L.mzhip_set_stream_window.restype = None
    if os.environ.get("MZ_FUZZ_ONE_BUFFER"):  # (the same cases through the one-buffer path: entries below the default window of 64 MiB)
        L.mzhip_set_stream_window(0, 0)
    else:
        L.mzhip_set_stream_window(3 << 20, 512 << 10)
    text = synth.bench_corpus()[0]
    try:
        return _run(n_streams, rnd, hip, ref, text, verbose)
    finally:
        L.mzhip_set_stream_window(0, 0)


def _run(n_streams, rnd, hip, ref, text, verbose):


    def piece():
        k = rnd.randrange(5)
        n = rnd.randrange(20000, 900000)
        if k == 0:
            o = rnd.randrange(len(text) - 1)
            return (text[o:] + text)[:n]
        if k == 1:
            return bytes(rnd.getrandbits(8) for _ in range(min(n, 200000)))
        if k == 2:
            return bytes([rnd.randrange(256)]) * n
        if k == 3:
            w = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40)))
            return (w * (n // len(w) + 1))[:n]
        return text[::-1][:n]


    mism = cases = soft = 0
    worst = 0
    for it in range(n_streams):
        wb = rnd.choice((-15, -15, 15, 31))
        parts = [piece() for _ in range(rnd.randrange(2, 9))]
        z = b""
        co = zlib.compressobj(rnd.randrange(0, 10), zlib.DEFLATED, wb, 8, rnd.choice((0, 0, 0, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE)))
        for i, p in enumerate(parts):
            z += co.compress(p)
            if rnd.random() < 0.3:
                z += co.flush(rnd.choice((zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH)))
        z += co.flush()
        d = b"".join(parts)
        cap = len(d) + (1 << 20)
        variants = [("whole", z), ("cut", z[:rnd.randrange(len(z) // 4, len(z))])]
        zz = bytearray(z)
        zz[rnd.randrange(len(zz))] ^= 1 << rnd.randrange(8)
        variants.append(("flip", bytes(zz)))
        for name, data in variants:
            chunk = rnd.choice((65535, 65535, 1 << 20, 7777))
            if ONLY and it not in ONLY:
                continue  # (MZ_FUZZ_ONLY=i,j,...: only these streams are decoded; the others are still generated, so that the random sequence is the same)
            # TOTAL_IN_MAX as mz_zip sets it (the entry's compressed size) -- or, for a cut, the limit instead of the end of the file
            lim = rnd.choice((0, 0, len(data)))
            if name == "cut" and rnd.random() < 0.5:
                lim, data = len(data), z
            a = hip.stream_decode(8, data, 2 * cap, chunk=chunk, window_bits=wb, max_in=lim)
            b = ref.stream_decode(8, data, 2 * cap, chunk=chunk, window_bits=wb, max_in=lim)
            cases += 1
            if name == "flip" and b["error"] != 0 and a["total_in"] != b["total_in"] and all(a[k] == b[k] for k in KEYS if k != "total_in"):
                if verbose or abs(a["total_in"] - b["total_in"]) > TOTAL_IN_SLACK:
                    print("TOTAL_IN at the error: stream %d wbits %d chunk %d len %d: %d here, %d there (error %d, %d bytes out)" % (it, wb, chunk, len(data), a["total_in"], b["total_in"], b["error"], len(b["out"])))
                soft += 1   # TOTAL_IN at a data error is where inflate()'s bit buffer stood: best effort (as tests/test_gpu_dropin.py's bit flips)
                worst = max(worst, abs(a["total_in"] - b["total_in"]))
                continue
            if {k: a[k] for k in KEYS} != {k: b[k] for k in KEYS}:
                mism += 1
                if verbose:
                    print("MISMATCH stream %d %s wbits %d chunk %d len %d:" % (it, name, wb, chunk, len(data)),
                          {k: (a[k], b[k]) for k in KEYS if k != "out" and a[k] != b[k]}, "bytes equal" if a["out"] == b["out"] else "BYTES DIFFER")
    return cases, mism, soft, worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cases, mism, soft, worst = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("gpu window fuzz: %d streams, %d cases -- %d mismatches (%d corrupted streams agree in everything but TOTAL_IN at the error, by %d bytes at most)" % (n, cases, mism, soft, worst))
    sys.exit(1 if mism else 0)
