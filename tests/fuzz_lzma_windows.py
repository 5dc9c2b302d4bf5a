#!/usr/bin/env python3
"""Differential run of method-14 READ (not pytest): random ZIP-LZMA streams of 0.1 - 2 MB -- text, noise, runs, mixtures;
presets 0 - 9 and hand-set lc / lp / pb / dictionary sizes from 4 KiB up; with and without the entry's size as TOTAL_OUT_MAX (as
mz_zip sets it) -- through the drop-in's mz_stream_lzma READ, one buffer and in windows (192 KiB, 48 KiB
gulps), and through the all-reference build: whole, cut at a random byte, cut inside the end marker, with a random bit flipped.  Compared: every read()
return value, the bytes, close(), error(), TOTAL_OUT, the base position -- and TOTAL_IN whenever the stream was not refused as
corrupt (at a data error liblzma's total says how far its range decoder ran on: best effort).
    python tests/fuzz_lzma_windows.py [streams] [seed] [library]"""
import ctypes as C
import lzma
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402

KEYS = ("rets", "out", "total_out", "close", "error", "open")


def zip_lzma(data, filt, eos):
    raw = lzma.compress(data, format=lzma.FORMAT_ALONE, filters=[filt])
    assert raw[5:13] == b"\xff" * 8  # python writes size = -1 and the end marker
    return bytes([5, 2, 5, 0]) + raw[:5] + raw[13:]  # (python always writes the end marker; TOTAL_OUT_MAX is set or not below)


def run(n_streams, seed, lib):
    rnd = random.Random(seed)
    hip, ref = oracle.MzDriver(lib), oracle.ref()
    L = hip.L
    L.mzhip_set_stream_window.argtypes = [C.c_int64, C.c_int64]
    L.mzhip_set_stream_window.restype = None
    text = synth.bench_corpus()[0]

    def piece():
        k, n = rnd.randrange(5), rnd.choice((rnd.randrange(1, 20000), rnd.randrange(2000, 400000)))
        if k == 0:
            o = rnd.randrange(len(text) - 1)
            return (text[o:] + text)[:n]
        if k == 1:
            return bytes(rnd.getrandbits(8) for _ in range(min(n, 60000)))
        if k == 2:
            return bytes([rnd.randrange(256)]) * n
        if k == 3:
            w = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40)))
            return (w * (n // len(w) + 1))[:n]
        return text[::-1][:n]

    cases = mism = soft = 0
    for it in range(n_streams):
        d = b"".join(piece() for _ in range(rnd.randrange(1, 6)))
        if rnd.random() < 0.5:
            filt = dict(id=lzma.FILTER_LZMA1, preset=rnd.randrange(10))
        else:
            lc = rnd.randrange(5)
            filt = dict(id=lzma.FILTER_LZMA1, preset=rnd.randrange(7), lc=lc, lp=rnd.randrange(5 - lc), pb=rnd.randrange(5),
                        dict_size=1 << rnd.randrange(12, 24))
        eos = rnd.random() < 0.6
        z = zip_lzma(d, filt, eos)
        variants = [("whole", z), ("cut", z[:rnd.randrange(len(z) // 4, len(z))]), ("tail", z[:len(z) - rnd.randrange(1, 14)])]  # (tail: a cut inside the end marker)
        zz = bytearray(z)
        zz[rnd.randrange(len(zz))] ^= 1 << rnd.randrange(8)
        variants.append(("flip", bytes(zz)))
        for name, data in variants:
            chunk = rnd.choice((65535, 65535, 1 << 20, 7777, 7, 100) if len(d) < 60000 else (65535, 65535, 1 << 20, 7777))  # (small entries also in tiny read() calls: which call reports a refusal)
            max_out = len(d) if eos else -1
            b = ref.stream_decode(14, data, len(d) + 70000, chunk=chunk, max_out=max_out)
            if name == "flip" and max_out >= 0 and (b["total_out"] >= max_out or any(r < 0 and r != -3 for r in b["rets"])):
                continue  # (a corrupted stream that reaches TOTAL_OUT_MAX: behind the limit the reference decodes on in read()-sized steps and its
                          # byte counts go negative, e.g. -601 -- the case tests/test_gpu_dropin.py::test_lzma_window_mode leaves out)
            for win in (0, 1):
                L.mzhip_set_stream_window(192 << 10 if win else 0, 48 << 10 if win else 0)
                a = hip.stream_decode(14, data, len(d) + 70000, chunk=chunk, max_out=max_out)
                cases += 1
                # (a corrupted stream that runs past TOTAL_OUT_MAX: the reference's TOTAL_OUT goes down again by what liblzma wrote
                # behind the limit, once per lzma_code call that follows -- mz_strm_lzma.c:214-221 -- not compared, as in
                # tests/test_gpu_dropin.py::test_lzma_window_mode)
                same = all(a[k] == b[k] for k in KEYS if not (k == "total_out" and name == "flip" and b["error"] != 0))
                if same and a["total_in"] != b["total_in"]:
                    if b["error"] == -3 or (b["rets"] and b["rets"][-1] == -3):
                        soft += 1
                        continue
                    same = False
                if not same:
                    mism += 1
                    print("MISMATCH stream %d %s %s chunk %d max_out %d len %d filt %s eos %d:" % (it, name, "windows" if win else "one buffer", chunk, max_out, len(data), filt, eos),
                          {k: (a[k], b[k]) for k in KEYS + ("total_in", "base_pos") if k != "out" and a[k] != b[k]}, "bytes equal" if a["out"] == b["out"] else "BYTES DIFFER")
    L.mzhip_set_stream_window(0, 0)
    return cases, mism, soft


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "integration", "_build", "libmzhipdrop.so")
    cases, mism, soft = run(n, seed, lib)
    print("lzma window fuzz: %d streams, %d cases -- %d mismatches (%d corrupted streams agree in everything but TOTAL_IN at the data error)" % (n, cases, mism, soft))
