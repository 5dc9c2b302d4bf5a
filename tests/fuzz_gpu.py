"""GPU-box probe (not a pytest): one large differential run of K1 through the C ABI -- N random valid raw-DEFLATE streams
(levels 0-9, all strategies, window 9-15 bits, memLevel 1-9, mid-stream full flushes) and four corruptions of each, one
launch; every entry is then checked on the host cores against the oracle restatement (status class; bytes, consumed
input and CRC wherever the stream decodes).  Usage: python tests/fuzz_gpu.py [N=8000] [seed=1]"""
import multiprocessing as mp
import random
import sys
import zlib

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402
from tests import gpu_util, synth  # noqa: E402


def gen(rnd, c):
    k, n = rnd.randrange(6), rnd.randrange(1, 30000)
    if k == 0:
        d = c[rnd.randrange(len(c) - n):][:n]
    elif k == 1:
        d = bytes(rnd.randrange(256) for _ in range(min(n, 4000)))
    elif k == 2:
        d = bytes([rnd.randrange(4)]) * n
    elif k == 3:
        d = bytes(min(255, int(rnd.expovariate(1 / (8 + 200 * rnd.random())))) for _ in range(min(n, 8000)))
    elif k == 4:
        d = (c[rnd.randrange(1000):][:rnd.randrange(1, 300)]) * rnd.randrange(1, 60)
    else:
        d = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 300))) + c[:n]
    co = zlib.compressobj(rnd.randrange(0, 10), zlib.DEFLATED, -rnd.randrange(9, 16), rnd.randrange(1, 10),
                          rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    z = co.compress(d[:len(d) // 2]) + (co.flush(zlib.Z_FULL_FLUSH) if rnd.random() < 0.3 else b"") + \
        co.compress(d[len(d) // 2:]) + co.flush()
    return d, z


def corrupt(rnd, z):
    b = bytearray(z)
    kk = rnd.randrange(4)
    if kk == 0:
        b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    elif kk == 1:
        b[rnd.randrange(len(b))] = rnd.randrange(256)
    elif kk == 2:
        del b[rnd.randrange(1, len(b) + 1):]
    else:
        b[rnd.randrange(min(len(b), 60))] = rnd.randrange(256)
    return bytes(b)


def check(args):
    z, cap = args
    st, used, out = oracle.inflate_raw(z, cap)
    return st, used, len(out), zlib.crc32(out)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    c = synth.corpus()
    pays, caps = [], []
    for _ in range(n):
        d, z = gen(rnd, c)
        pays.append(z + b"xyz")
        caps.append(len(d) + 16)
        for _ in range(4):
            pays.append(corrupt(rnd, z))
            caps.append(len(d) + 70000)
    with mp.Pool() as pool:                      # forks before the HIP context exists
        want = pool.map(check, list(zip(pays, caps)), chunksize=64)
    batch = gpu_util.make_batch(pays, caps)
    out_len, in_used, crc, status = gpu_util.run_inflate(batch)
    bad = 0
    for i, (st, used, ol, k) in enumerate(want):
        ok = status[i] == st and (st != 0 or (in_used[i], out_len[i], int(crc[i])) == (used, ol, k))
        if not ok:
            bad += 1
            if bad < 10:
                print("MISMATCH", i, "gpu", status[i], in_used[i], out_len[i], hex(int(crc[i])), "oracle", st, used, ol, hex(k))
    n_ok = sum(1 for w in want if w[0] == 0)
    print("gpu fuzz: %d streams (%d decode, %d fail) -- %d mismatches" % (len(pays), n_ok, len(pays) - n_ok, bad))
    sys.exit(1 if bad else 0)
