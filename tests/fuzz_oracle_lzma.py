#!/usr/bin/env python3
"""CPU-side probe (not a pytest; needs oracle/_ref, i.e. this container): the oracle's restatement of the method-14 decode
against the compiled reference (mz_stream_lzma READ over liblzma 5.2.5) on random streams -- text, noise, runs, mixtures;
presets 0 - 9 and hand-set lc / lp / pb / dictionary sizes -- whole, cut, and with a byte flipped or replaced anywhere
(properties and header included).  Compared as tests/test_oracle.py::test_lzma_parity_with_reference compares its six
variants per case: refused by the reference -> refused by the restatement; else the same bytes.
    python tests/fuzz_oracle_lzma.py [N=300] [seed=1]"""
import lzma
import multiprocessing as mp
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402


def zip_lzma(data, filt):
    raw = lzma.compress(data, format=lzma.FORMAT_ALONE, filters=[filt])
    return bytes([5, 2, 5, 0]) + raw[:5] + raw[13:]


def make(seed):
    rnd = random.Random(seed)
    c = synth.corpus()
    k, n = rnd.randrange(5), rnd.randrange(1, 60000)
    if k == 0:
        d = c[rnd.randrange(len(c) - n):][:n]
    elif k == 1:
        d = bytes(rnd.getrandbits(8) for _ in range(min(n, 6000)))
    elif k == 2:
        d = bytes([rnd.randrange(256)]) * n
    elif k == 3:
        w = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 40)))
        d = (w * (n // len(w) + 1))[:n]
    else:
        d = c[:n // 2] + bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 500))) + c[1000:1000 + n // 2]
    if rnd.random() < 0.5:
        filt = dict(id=lzma.FILTER_LZMA1, preset=rnd.randrange(0, 10))
    else:
        lc = rnd.randrange(0, 5)
        filt = dict(id=lzma.FILTER_LZMA1, preset=rnd.randrange(0, 7), lc=lc, lp=rnd.randrange(0, 5 - lc), pb=rnd.randrange(0, 5),
                    dict_size=1 << rnd.randrange(12, 22))
    z = zip_lzma(d, filt)
    variants = [("whole", z), ("cut", z[:rnd.randrange(1, len(z))]), ("cut tail", z[:max(1, len(z) - rnd.randrange(1, 8))])]
    for _ in range(4):
        b = bytearray(z)
        at = rnd.randrange(len(b)) if rnd.random() < 0.7 else rnd.randrange(min(len(b), 24))
        if rnd.random() < 0.5:
            b[at] ^= 1 << rnd.randrange(8)
        else:
            b[at] = rnd.randrange(256)
        variants.append(("byte %d" % at, bytes(b)))
    return d, variants


def check(seed):
    ref = oracle.ref()
    d, variants = make(seed)
    bad = []
    for name, z in variants:
        cap = len(d) + 70000
        r = ref.stream_decode(14, z, cap)
        last = r["rets"][-1] if r["rets"] else r["open"]
        st, used, out = oracle.lzma_zip_decode(z, cap, -1)
        ok = (st == -3) if last < 0 else (st == 0 and out == r["out"])
        if not ok:
            bad.append((seed, name, len(z), "oracle", st, used, len(out), "reference", r["rets"][-3:], r["open"], r["error"], len(r["out"])))
    return len(variants), bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    with mp.Pool() as pool:
        res = pool.map(check, [seed * 1000003 + i for i in range(n)], chunksize=4)
    cases = sum(r[0] for r in res)
    bad = [b for r in res for b in r[1]]
    for b in bad[:10]:
        print("MISMATCH", b)
    print("oracle vs reference, method 14: %d streams, %d cases -- %d mismatches" % (n, cases, len(bad)))
    sys.exit(1 if bad else 0)
