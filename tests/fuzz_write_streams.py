#!/usr/bin/env python3
"""Differential run of the WRITE direction (not pytest): random buffers of 1 B - 1.5 MB -- text, noise, runs, mixtures, empty --
through the drop-in's mz_stream_zlib / mz_stream_lzma WRITE (methods 8, 14, 95; every level; raw / zlib / gzip framing and small
windows for method 8; write() calls of several sizes; WRITE segments of 128 KiB so that entries leave in several launches) --
what comes out is decoded by the all-reference build and must be the input, byte for byte; TOTAL_IN, close(), error() and
is_open() are compared with the reference's own WRITE of the same calls.
    python tests/fuzz_write_streams.py [buffers] [seed] [library]"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402


def run(n, seed, lib):
    rnd = random.Random(seed)
    hip, ref = oracle.MzDriver(lib), oracle.ref()
    L = hip.L
    L.mzhip_set_write_segment.argtypes = [C.c_int64]
    L.mzhip_set_write_segment.restype = None
    text = synth.bench_corpus()[0]

    def piece():
        k, m = rnd.randrange(6), rnd.randrange(1, 300000)
        if k == 0:
            o = rnd.randrange(len(text) - 1)
            return (text[o:] + text)[:m]
        if k == 1:
            return bytes(rnd.getrandbits(8) for _ in range(min(m, 50000)))
        if k == 2:
            return bytes([rnd.randrange(256)]) * m
        if k == 3:
            w = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40)))
            return (w * (m // len(w) + 1))[:m]
        if k == 4:
            return text[::-1][:m]
        return b""

    cases = bad = 0
    for it in range(n):
        d = b"".join(piece() for _ in range(rnd.randrange(0, 6)))
        method = rnd.choice((8, 8, 14, 95))
        if method == 95 and "mock" in os.path.basename(lib):
            method = 14  # (the host emulation has no .xz container writer: that is host code next to the kernels)
        level = rnd.randrange(0, 10)
        wb = rnd.choice((0, -15, 15, 31, -9, 10, 28)) if method == 8 else 0
        chunk = rnd.choice((65535, 65535, 1 << 20, 7777, 100))
        if chunk == 100 and len(d) > 200000:
            chunk = 7777
        L.mzhip_set_write_segment(rnd.choice((0, 128 << 10)))
        try:
            z, ia = hip.stream_encode(method, d, level=level, chunk=chunk, window_bits=wb)
        except RuntimeError as e:
            cases += 1
            bad += 1
            print("MISMATCH buffer %d method %d level %d wbits %d chunk %d len %d: %s" % (it, method, level, wb, chunk, len(d), e))
            continue
        finally:
            L.mzhip_set_write_segment(0)
        zr, ib = ref.stream_encode(method, d, level=level, chunk=chunk, window_bits=wb)
        back = ref.stream_decode(method, z, len(d) + 70000, window_bits=wb)
        cases += 1
        why = []
        if back["out"] != d or back["error"] != 0 or (back["rets"] and back["rets"][-1] < 0):
            why.append("the reference does not read it back: rets %s error %d, %d of %d bytes" % (back["rets"][-2:], back["error"], len(back["out"]), len(d)))
        for k in ("total_in", "close", "error", "open"):
            if ia[k] != ib[k]:
                why.append("%s %d / %d" % (k, ia[k], ib[k]))
        if ia["total_out"] != len(z):
            why.append("TOTAL_OUT %d for %d bytes written" % (ia["total_out"], len(z)))
        if why:
            bad += 1
            print("MISMATCH buffer %d method %d level %d wbits %d chunk %d len %d -> %d (reference %d):" % (it, method, level, wb, chunk, len(d), len(z), len(zr)), "; ".join(why))
    return cases, bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    cases, bad = run(n, seed, lib)
    print("write fuzz: %d buffers -- %d mismatches" % (cases, bad))
